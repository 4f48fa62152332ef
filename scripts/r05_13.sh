#!/bin/bash
set -u
OUT=gpurun_out/r05_13
mkdir -p $OUT
python3 -m pytest tests -q -m gpu > $OUT/pytest_all.txt 2>&1
tail -15 $OUT/pytest_all.txt
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -3 $OUT/bench_default.err
python3 - <<'PY'
import json
r = json.loads(open("gpurun_out/r05_13/bench_default.json").read().strip().splitlines()[-1])
c = r["config"]["replay_check"] or {}
print(r["value"], "img/s", r["ms_per_step"], "ms; eager", r["eager_launch_ms_per_step"], "/", r.get("eager_launch_with_kernel_events_ms_per_step"), "ms;", r["config"]["launch"][:60], "; check ok", c.get("ok"), "upd", c.get("update_rel_l2"), "noise", c.get("noise_update_rel_l2"), r["config"]["miopen"])
print("fwd", r.get("forward_only"))
print("eager_rocm", r.get("eager_rocm", {}).get("fwd_images_per_sec"), r.get("eager_rocm", {}).get("fwd_bwd_images_per_sec"), "cpu", r.get("cpu_baseline", {}).get("value"))
print({k: v for k, v in r["roofline"].items() if k in ("frac", "avg_launch_us", "frac_fused", "path_frac", "path_ms_per_step", "traffic", "traffic_stale")})
for k, v in (r.get("other_configs") or {}).items():
    print(k, {a: b for a, b in v.items() if a in ("value", "ms_per_step", "launch", "replay_matches_eager", "weights_finite", "eager_launch_ms_per_step", "miopen", "error")}, (v.get("replay_check") or {}).get("ok"), (v.get("roofline") or {}).get("frac"))
PY
