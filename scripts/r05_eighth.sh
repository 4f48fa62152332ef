#!/bin/bash
set -u
OUT=gpurun_out/r05_eighth
mkdir -p $OUT
for v in fp32 nomrla mmbias nomlp noattn; do python3 scripts/deit_replay_debug3.py $v 2>&1 | grep -E "replay|Error|error" >> $OUT/deit_debug3.txt; done
cat $OUT/deit_debug3.txt
