#!/bin/bash
set -u
OUT=gpurun_out/r05_11
mkdir -p $OUT
python3 scripts/miopen_bwd_graph_probe.py 0 256 0 2>&1 | grep conv > $OUT/probe_bwd.txt
python3 scripts/miopen_bwd_graph_probe.py 1 256 0 2>&1 | grep conv >> $OUT/probe_bwd.txt
python3 scripts/miopen_bwd_graph_probe.py 1 256 1 2>&1 | grep conv >> $OUT/probe_bwd.txt
cat $OUT/probe_bwd.txt
for i in 1 2 3 4 5 6; do
  python3 -m pytest tests/test_ddp_gpu.py -q -m gpu -x -k "graph_captures_the_rccl or keeps_the_first" > $OUT/wd_$i.txt 2>&1
  tail -1 $OUT/wd_$i.txt
  if grep -q failed $OUT/wd_$i.txt; then grep -n "what():\|terminate\|Error" $OUT/wd_$i.txt | head -8; fi
done
