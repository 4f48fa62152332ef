#!/bin/bash
set -u
OUT=gpurun_out/r05_12
mkdir -p $OUT
python3 -m pytest tests/test_conv1x1_gpu.py -q -m gpu -x -k "strided" > $OUT/pytest_strided.txt 2>&1; tail -5 $OUT/pytest_strided.txt
python3 scripts/replay_grad_diag.py 256 0 0 2>&1 | grep replay > $OUT/diag.txt
python3 scripts/replay_grad_diag.py 256 1 1 2>&1 | grep replay >> $OUT/diag.txt
cat $OUT/diag.txt
python3 bench.py --no-baselines > $OUT/b.json 2> $OUT/b.err; tail -3 $OUT/b.err
python3 - <<'PY'
import json
r = json.loads(open("gpurun_out/r05_12/b.json").read().strip().splitlines()[-1])
c = r["config"]["replay_check"] or {}
print(r["value"], "img/s", r["ms_per_step"], "ms; eager", r["eager_launch_ms_per_step"], "ms;", r["config"]["launch"][:60], "; check ok", c.get("ok"), "upd", c.get("update_rel_l2"), "noise", c.get("noise_update_rel_l2"), r["config"]["miopen"])
print("fwd", r.get("forward_only"))
print({k: v for k, v in r["roofline"].items() if k in ("frac", "avg_launch_us", "frac_fused", "path_frac", "path_ms_per_step")})
PY
python3 -m pytest tests -q -m gpu > $OUT/pytest_all.txt 2>&1
tail -12 $OUT/pytest_all.txt
