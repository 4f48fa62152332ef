#!/bin/bash
# rocprofv3 kernel statistics of the detection backbone's step (2 x 3 x 800 x 1344; MIOpen in immediate mode so that no solver
# search is traced).  Usage on the GPU box: bash scripts/r06_det_trace.sh <outdir>
set -u
OUT=${1:-gpurun_out/r06_det_trace}
RAW=/tmp/det_trace_$$
mkdir -p $OUT $RAW
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -- python3 bench.py --arch det_resnet50_mrlal --shape 2x3x800x1344 --steps 10 --warmup 3 --no-baselines --benchmark 0 --graph 0 > $OUT/trace.log 2>&1
cp $RAW/*/*_kernel_stats.csv $OUT/kernel_stats.csv
grep '^{' $OUT/trace.log | tail -1 > $OUT/bench_line.json
head -40 $OUT/kernel_stats.csv | cut -c1-160
