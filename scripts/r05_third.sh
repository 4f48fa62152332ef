#!/bin/bash
set -u
OUT=gpurun_out/r05_third
mkdir -p $OUT
python3 -X faulthandler scripts/deit_replay_debug.py 32 0.1 > $OUT/deit_debug.txt 2>&1
python3 -X faulthandler scripts/deit_replay_debug.py 32 0.0 >> $OUT/deit_debug.txt 2>&1
grep -v Warning $OUT/deit_debug.txt | tail -30
python3 -X faulthandler bench.py --batch 32 --steps 3 --warmup 2 --no-baselines --no-forward-only --benchmark 0 --graph 1 > $OUT/bench_small.json 2> $OUT/bench_small.err
echo "bench rc=$?"; tail -5 $OUT/bench_small.err; cat $OUT/bench_small.json | head -c 3000
python3 -m pytest tests/test_sequences_gpu.py tests/test_graph_replay_gpu.py -q -m gpu -s > $OUT/pytest.txt 2>&1
tail -15 $OUT/pytest.txt
