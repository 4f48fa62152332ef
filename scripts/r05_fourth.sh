#!/bin/bash
set -u
OUT=gpurun_out/r05_fourth
mkdir -p $OUT
for a in "32 0.1 1 std" "32 0.1 0 std" "32 0.1 1 nomask"; do python3 scripts/deit_replay_debug.py $a 2>&1 | grep -v Warn | grep replay >> $OUT/deit_debug.txt; done
cat $OUT/deit_debug.txt
python3 -m pytest tests/test_sequences_gpu.py -q -m gpu -s -k resnet50 > $OUT/pytest.txt 2>&1
tail -8 $OUT/pytest.txt
