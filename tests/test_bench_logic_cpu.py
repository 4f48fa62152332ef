"""CPU: the decision logic of bench.py (benchkit/ranks.py, benchkit/report.py) that cannot wait for an 8-GPU box to be exercised
for the first time.

* `measure_exchange_schedules`: a failure while capturing the OPTIONAL second exchange schedule must leave the first
  schedule's finished graph-replayed measurement standing -- mocked captures, no GPU.
* `counters_current` / scripts/lib_identity.py: committed counter passes are quoted only for the library they were taken on.
"""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402  (re-exports the pieces below)
from benchkit import ranks as bench_ranks, report as bench_report  # noqa: E402


class _Clock:
    """time_region(run, n) -> n * the per-step time the mocked replay callable stands for."""

    def __init__(self):
        self.calls = []

    def __call__(self, run, n):
        self.calls.append((run.__name__, n))
        return n * run()


def _schedules(ms_a, ms_b, fail=None, verify_ok=(True, True)):
    log = []

    def prepare(name, first):
        log.append(("prepare", name, first))
        return {"name": name}

    def capture(h, name):
        log.append(("capture", name))
        if fail == name:
            raise bench.CaptureBroken("injected")

        def a():
            return ms_a * 1e-3

        def b():
            return ms_b * 1e-3
        return a if name == "after_backward" else b

    def verify(h, run, name):
        ok = verify_ok[0 if name == "after_backward" else 1]
        return {"ok": ok, "weights_rel_l2": 1e-4 if ok else 3.0}, ok
    clock = _Clock()
    after = []
    recs, chosen, failure = bench.measure_exchange_schedules(["after_backward", "bucketed_overlap"], prepare, capture, clock, 20, 6,
                                                             verify, lambda rec: after.append(rec["handle"]["name"]))
    return recs, chosen, failure, clock.calls, log, after


def test_a_failed_capture_of_the_second_schedule_keeps_the_first_schedules_graph_measurement():
    recs, chosen, failure, calls, log, after = _schedules(31.0, 30.0, fail="bucketed_overlap")
    assert failure is not None and failure[0] == "bucketed_overlap"
    assert chosen == "after_backward"                            # a finished, graph-replayed, FULL region
    assert recs["after_backward"]["dt"] == pytest.approx(20 * 31e-3) and recs["after_backward"]["check"]["ok"]
    assert "bucketed_overlap" not in recs
    assert calls == [("a", 20)]                                  # the full region of A ran BEFORE B was even prepared
    assert log.index(("capture", "bucketed_overlap")) > log.index(("capture", "after_backward"))
    assert after == ["after_backward"]


def test_a_failed_capture_of_the_first_schedule_reports_nothing_graphed():
    recs, chosen, failure, calls, _, _ = _schedules(31.0, 30.0, fail="after_backward")
    assert failure[0] == "after_backward" and chosen is None and recs == {} and calls == []


def test_the_faster_schedule_runs_its_own_full_region_and_wins():
    recs, chosen, failure, calls, _, after = _schedules(31.0, 30.0)
    assert failure is None and chosen == "bucketed_overlap"
    assert calls == [("a", 20), ("b", 6), ("b", 20)]
    assert recs["bucketed_overlap"]["ab_ms"] == pytest.approx(30.0) and recs["after_backward"]["ab_ms"] == pytest.approx(31.0)
    assert after == ["after_backward", "bucketed_overlap"]


def test_a_slower_second_schedule_gets_no_full_region():
    recs, chosen, failure, calls, _, _ = _schedules(30.0, 31.5)
    assert failure is None and chosen == "after_backward" and calls == [("a", 20), ("b", 6)]
    assert recs["bucketed_overlap"]["dt"] is None and recs["bucketed_overlap"]["ab_ms"] == pytest.approx(31.5)


def test_a_replay_that_does_not_reproduce_the_eager_step_is_never_the_timed_value():
    recs, chosen, _, calls, _, _ = _schedules(31.0, 30.0, verify_ok=(False, True))
    assert chosen == "bucketed_overlap" and recs["after_backward"]["dt"] is None and ("a", 20) not in calls
    recs, chosen, _, calls, _, _ = _schedules(31.0, 30.0, verify_ok=(True, False))
    assert chosen == "after_backward" and recs["bucketed_overlap"]["dt"] is None
    recs, chosen, _, _, _, _ = _schedules(31.0, 30.0, verify_ok=(False, False))
    assert chosen is None and all(v["dt"] is None for v in recs.values())


def test_counter_passes_are_tied_to_the_library(tmp_path, monkeypatch):
    me = bench.library_identity()
    assert len(me["lib_sha256"]) == 64 and len(me["src_sha256"]) == 64 and me["sources"] >= 20
    assert bench.counters_current({"lib_sha256": me["lib_sha256"], "src_sha256": "x"}) == "library sha256"
    assert bench.counters_current({"lib_sha256": "x", "src_sha256": me["src_sha256"]}) == "source sha256"
    assert bench.counters_current({"lib_sha256": "x", "src_sha256": "y"}) is None
    assert bench.counters_current(None) is None                   # a file without the record (an older round's) is stale

    class A:
        arch, batch = "resnet50_mrlal", 256
    assert bench.measure_exchange_schedules is bench_ranks.measure_exchange_schedules and bench.pmc_traffic is bench_report.pmc_traffic
    monkeypatch.setattr(bench_report, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench_report, "library_identity", lambda: me)
    os.makedirs(tmp_path / "profiles")
    path = tmp_path / "profiles" / f"{bench.ROUND}_pmc_traffic_resnet50_mrlal_b256.json"
    rec = {"light_apply_bwd": {"hbm_bytes_per_launch": 1.1e9}}
    path.write_text(json.dumps(dict(rec, _meta={"lib_sha256": me["lib_sha256"]})))
    assert bench.pmc_traffic(A, "mrla_light_apply_bwd")[0] == 1100000000
    path.write_text(json.dumps(dict(rec, _meta={"lib_sha256": "0" * 64, "src_sha256": "1" * 64})))
    assert bench.pmc_traffic(A, "mrla_light_apply_bwd") == (None, os.path.join("profiles", path.name), "stale")
    path.write_text(json.dumps(rec))
    assert bench.pmc_traffic(A, "mrla_light_apply_bwd")[2] == "stale"
    path.unlink()
    assert bench.pmc_traffic(A, "mrla_light_apply_bwd") == (None, None, None)
