#!/bin/bash
# In-step A/B of the training tail with (MRLA_LEAN=0) and without (1) a stored x_t: alternating runs on the same box.
OUT=$1; RUNS=${2:-2}
: > "$OUT"
for i in $(seq 1 "$RUNS"); do
  for lean in 1 0; do
    echo "# run $i lean $lean" >> "$OUT"
    MRLA_LEAN=$lean python scripts/instep_kernels.py 10 mrla_light >> "$OUT" 2>> gpurun_out/ab_err.log
  done
done
