#!/bin/bash
set -u
OUT=gpurun_out/r05_second
mkdir -p $OUT
python3 -X faulthandler bench.py --batch 32 --steps 3 --warmup 2 --no-baselines --no-forward-only --benchmark 0 --graph 1 > $OUT/bench_small.json 2> $OUT/bench_small.err
echo "bench rc=$?"
tail -30 $OUT/bench_small.err
python3 -m pytest tests/test_sequences_gpu.py tests/test_graph_replay_gpu.py -q -m gpu -s > $OUT/pytest.txt 2>&1
tail -40 $OUT/pytest.txt
