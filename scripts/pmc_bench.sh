#!/bin/bash
# HBM-traffic PMC passes over the real bench.py step (separate passes for FETCH_SIZE / WRITE_SIZE, as the microarch
# guide prescribes; no trace domains besides --kernel-trace; MIOpen in immediate mode there -- its solver search under
# counter collection takes tens of minutes and does not touch the MRLA kernels' traffic).  Usage on the GPU box: [TRACE_ONLY=1] bash scripts/pmc_bench.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc_bench}
ARGS=${BENCH_ARGS:-}            # e.g. BENCH_ARGS="--arch resnet101_mrlab --batch 128"
RAW=/tmp/pmc_raw_$$           # raw traces are large: only the summaries go back through gpurun_out/
mkdir -p $OUT $RAW
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
if [ "${TRACE_ONLY:-0}" != "1" ]; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $RAW/fetch -- python3 bench.py $ARGS --steps 2 --warmup 2 --no-baselines --benchmark 0 --graph 0 > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $RAW/write -- python3 bench.py $ARGS --steps 2 --warmup 2 --no-baselines --benchmark 0 --graph 0 > $OUT/write.log 2>&1
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -- python3 bench.py $ARGS --steps 6 --warmup 3 --no-baselines > $OUT/trace.log 2>&1
PMC_COMMAND="rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py $ARGS --steps 2 --warmup 2 --no-baselines --benchmark 0 --graph 0" python3 scripts/summarize_profile.py $RAW $OUT
cp $RAW/trace/*/*_kernel_stats.csv $OUT/kernel_stats_full.csv 2>/dev/null
