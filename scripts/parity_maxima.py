"""Fold gpurun_out/parity_maxima.jsonl (written by tests/cases.py::record during `pytest -m gpu` and the CPU golden tests)
into a markdown table: per assertion (file:line) the number of comparisons, the worst measured error relative to the
tensor's max-abs, the test / tag that produced it, and the asserted bound as written in the source line.

    python scripts/parity_maxima.py gpurun_out/parity_maxima.jsonl > profiles/r04_parity_maxima.md
"""
import collections
import json
import sys


def main(path):
    by = collections.OrderedDict()
    for ln in open(path):
        r = json.loads(ln)
        e = by.setdefault(r["where"], dict(n=0, worst=-1.0, test="", tag="", expr=r["expr"]))
        e["n"] += 1
        if r["rel"] > e["worst"]:
            e.update(worst=r["rel"], test=r["test"], tag=r.get("tag", ""))
    keyf = lambda kv: (kv[0].split(":")[0], int(kv[0].split(":")[1]))                    # noqa: E731
    print("| assertion | comparisons | worst measured (rel. to max-abs) | produced by | source line |")
    print("|---|---|---|---|---|")
    for where, e in sorted(by.items(), key=keyf):
        test = e["test"].split("::", 1)[-1]
        # (the tag is a loop variable of the test -- the parameter's name; shown where the source line uses one)
        tag = f" `{e['tag']}`" if e["tag"] and any(v in e["expr"] for v in ("ours", "theirs", "pn", "key]", "[k]")) else ""
        expr = e["expr"].replace("|", "\\|")
        print(f"| `{where}` | {e['n']} | {e['worst']:.2e} | `{test}`{tag} | `{expr}` |")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_maxima.jsonl")
