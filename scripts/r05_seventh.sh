#!/bin/bash
set -u
OUT=gpurun_out/r05_seventh
mkdir -p $OUT
for v in cache_on cache_off block linear_mul; do python3 scripts/deit_replay_debug2.py $v 2>&1 | grep replay >> $OUT/deit_debug2.txt; done
cat $OUT/deit_debug2.txt
