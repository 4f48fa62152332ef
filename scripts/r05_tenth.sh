#!/bin/bash
set -u
OUT=gpurun_out/r05_tenth
mkdir -p $OUT
python3 scripts/replay_grad_diag.py 256 1 0 2>&1 | grep replay > $OUT/diag.txt
python3 scripts/replay_grad_diag.py 256 1 1 2>&1 | grep replay >> $OUT/diag.txt
python3 scripts/replay_grad_diag.py 256 0 0 2>&1 | grep replay >> $OUT/diag.txt
cat $OUT/diag.txt
for v in "--benchmark 1 --deterministic 1" "--benchmark 0 --graph 1 --deterministic 0"; do
  python3 bench.py $v --no-baselines --no-forward-only > $OUT/b.json 2> $OUT/b.err
  python3 - "$v" <<'PY'
import json, sys
r = json.loads(open("gpurun_out/r05_tenth/b.json").read().strip().splitlines()[-1])
c = r["config"]["replay_check"] or {}
print(sys.argv[1], "->", r["value"], "img/s", r["ms_per_step"], "ms; eager", r["eager_launch_ms_per_step"], "ms;", r["config"]["launch"][:40], "; check ok", c.get("ok"), "upd", c.get("update_rel_l2"), "noise", c.get("noise_update_rel_l2"))
PY
done
for i in 1 2 3 4; do python3 -m pytest tests/test_ddp_gpu.py -q -m gpu -k "graph_captures_the_rccl or keeps_the_first" 2>&1 | tail -1; done
