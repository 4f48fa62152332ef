#!/bin/bash
set -u
OUT=gpurun_out/r05_14
mkdir -p $OUT
python3 bench.py --arch deit_mrlal_tiny_patch16_224 --steps 10 --warmup 3 --no-baselines > $OUT/deit.json 2> $OUT/deit.err; tail -2 $OUT/deit.err
python3 - <<'PY'
import json
r = json.loads(open("gpurun_out/r05_14/deit.json").read().strip().splitlines()[-1])
c = r["config"]["replay_check"] or {}
print(r["value"], "img/s", r["ms_per_step"], "ms; eager", r["eager_launch_ms_per_step"], "ms;", r["config"]["launch"][:40], "; ok", c.get("ok"), "upd", c.get("update_rel_l2"), "noise", c.get("noise_update_rel_l2"))
PY
python3 -m pytest tests/test_graph_replay_gpu.py tests/test_tokens_gpu.py tests/test_token_base_gpu.py -q -m gpu 2>&1 | tail -3
