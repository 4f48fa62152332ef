#!/bin/bash
# round 5, first GPU call: MIOpen probe over batches, bench with the replay check on the three configs
set -u
OUT=gpurun_out/r05_first
mkdir -p $OUT
python3 scripts/miopen_wrw_graph_probe.py 0 16,32,64,128 > $OUT/probe_imm.txt 2>&1
python3 scripts/miopen_wrw_graph_probe.py 1 16,32,128 > $OUT/probe_find.txt 2>&1
python3 bench.py --no-baselines --no-forward-only > $OUT/bench_r50.json 2> $OUT/bench_r50.err
python3 bench.py --arch resnet101_mrlab --batch 128 --steps 10 --warmup 3 --no-baselines --no-forward-only > $OUT/bench_r101.json 2> $OUT/bench_r101.err
python3 bench.py --arch deit_mrlal_tiny_patch16_224 --steps 10 --warmup 3 --no-baselines --no-forward-only > $OUT/bench_deit.json 2> $OUT/bench_deit.err
python3 -m pytest tests/test_graph_replay_gpu.py -x -q -m gpu -s > $OUT/pytest_replay.txt 2>&1
tail -5 $OUT/pytest_replay.txt
