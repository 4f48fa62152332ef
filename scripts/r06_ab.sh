#!/bin/bash
# In-step A/B of two library builds: alternating runs of scripts/instep_kernels.py on the same box.
# Usage: bash scripts/r06_ab.sh <out-file> <runs> <name-substring> <libA|product> <libB|product> ...
OUT=$1; RUNS=$2; SUB=$3; shift 3
mkdir -p gpurun_out
: > "$OUT"
for i in $(seq 1 "$RUNS"); do
  for lib in "$@"; do
    if [ "$lib" = product ]; then unset KBENCH_LIB; else export KBENCH_LIB=$lib; fi
    echo "# run $i lib $lib" >> "$OUT"
    python scripts/instep_kernels.py 10 "$SUB" >> "$OUT" 2>> gpurun_out/ab_err.log
  done
done
