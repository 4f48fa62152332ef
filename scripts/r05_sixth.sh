#!/bin/bash
set -u
OUT=gpurun_out/r05_sixth
mkdir -p $OUT
python3 scripts/reduce_graph_probe.py > $OUT/reduce_probe.txt 2>&1
cat $OUT/reduce_probe.txt | grep -v amdgpu.ids
python3 -m pytest tests -q -m gpu > $OUT/pytest_all.txt 2>&1
tail -60 $OUT/pytest_all.txt
