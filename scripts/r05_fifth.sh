#!/bin/bash
set -u
OUT=gpurun_out/r05_fifth
mkdir -p $OUT
TORCH_BLAS_PREFER_HIPBLASLT=0 python3 scripts/deit_replay_debug.py 32 0.1 1 std 2>&1 | grep replay | sed 's/^/hipblaslt-off: /' >> $OUT/deit_debug.txt
DISABLE_ADDMM_CUDA_LT=1 python3 scripts/deit_replay_debug.py 32 0.1 1 std 2>&1 | grep replay | sed 's/^/addmm-lt-off: /' >> $OUT/deit_debug.txt
python3 scripts/deit_replay_debug.py 256 0.1 1 std 2>&1 | grep replay | sed 's/^/b256: /' >> $OUT/deit_debug.txt
cat $OUT/deit_debug.txt
python3 -m pytest tests -x -q -m gpu > $OUT/pytest_all.txt 2>&1
tail -15 $OUT/pytest_all.txt
