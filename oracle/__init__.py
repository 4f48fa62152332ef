"""TEST INFRASTRUCTURE ONLY -- CPU / eager restatement of the MRLA hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` / eager-baseline legs
may import it, and there only as the checker or the thing a baseline is timed on.
``mrla_amd`` never imports this package.

Parity status: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, generated in the build container by ``oracle/make_goldens.py`` (which imports
``/root/reference``) and committed under ``tests/golden/``.
"""
