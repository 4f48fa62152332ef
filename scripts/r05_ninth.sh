#!/bin/bash
set -u
OUT=gpurun_out/r05_ninth
mkdir -p $OUT
python3 -m pytest tests/test_graph_replay_gpu.py tests/test_ddp_gpu.py tests/test_tokens_gpu.py tests/test_token_base_gpu.py tests/test_fullsize_tokens_gpu.py -q -m gpu -s > $OUT/pytest.txt 2>&1
tail -25 $OUT/pytest.txt
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc=$?"; tail -3 $OUT/bench_default.err; head -c 2500 $OUT/bench_default.json
