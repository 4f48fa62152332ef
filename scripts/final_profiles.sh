#!/bin/bash
# Round-end measurement set on the GPU box: the full bench line (with baselines) and the trace / PMC passes per config.
# Usage: bash scripts/final_profiles.sh <outdir> <mode> config ...   (config = "arch:batch"; mode = full | trace | bench)
# full also takes the whole-step MFMA counter pass (scripts/pmc_mfma.sh) of every config.
set -u
OUT=$1; MODE=$2; shift 2
mkdir -p $OUT
for cfg in "$@"; do
  arch=${cfg%%:*}; batch=${cfg##*:}
  if [ $MODE != trace ]; then python3 bench.py --arch $arch --batch $batch --no-others > $OUT/bench_${arch}.json 2> $OUT/bench_${arch}.err; fi
  if [ $MODE = full ]; then BENCH_ARGS="--arch $arch --batch $batch" bash scripts/pmc_bench.sh $OUT/pmc_${arch}_b${batch} > /dev/null 2>&1; fi
  if [ $MODE = full ]; then BENCH_ARGS="--arch $arch --batch $batch" bash scripts/pmc_mfma.sh $OUT/mfma_${arch}_b${batch} > /dev/null 2>&1; fi
  if [ $MODE != full ]; then TRACE_ONLY=1 BENCH_ARGS="--arch $arch --batch $batch" bash scripts/pmc_bench.sh $OUT/pmc_${arch}_b${batch} > /dev/null 2>&1; fi
done
ls $OUT
