#!/bin/bash
# SQ counters + HBM traffic of the backward apply pass in its three forms, isolated launches at 256 x 56^2 and 512 x 28^2 (b = 256, bf16,
# NHWC; scripts/kbench.py through the C ABI; rocprofv3 --pmc in separate passes per counter group, scripts/pmc_kbench.sh):
#   plain  = the product kernel (light_apply_bwd_wide),
#   packed = experiments/light_apply_bwd_pk.h (scripts/variants/libmrla_hip_pk.so: -DMRLA_APPLY_BWD_PK=1 -DMRLA_APPLY_BWD_DEPTH=1),
#   lean   = light_apply_bwd_lean_wide (x_t re-formed; kbench "lean apply_bwd").
# Usage on the GPU box: bash scripts/r06_sq.sh <outdir>
set -u
OUT=${1:-gpurun_out/r06_sq}
mkdir -p $OUT
for stage in 0 1; do
  STAGE=$stage LAYOUT=nhwc bash scripts/pmc_kbench.sh $OUT/plain_s$stage "apply_bwd+bn3sums" > /dev/null 2>&1
  STAGE=$stage LAYOUT=nhwc KBENCH_LIB=scripts/variants/libmrla_hip_pk.so bash scripts/pmc_kbench.sh $OUT/packed_s$stage "apply_bwd+bn3sums" > /dev/null 2>&1
  STAGE=$stage LAYOUT=nhwc bash scripts/pmc_kbench.sh $OUT/lean_s$stage "lean" > /dev/null 2>&1
done
for d in $OUT/*; do echo "== $d"; python3 scripts/pmc_summarize.py $d | grep -A1 -E "light_apply_bwd|light_stats_bwd_lean|light_stats_fwd_fused|light_apply_fwd_pre"; done > $OUT/summary.txt
for stage in 0 1; do
  for v in product scripts/variants/libmrla_hip_pk.so; do
    echo "# stage $stage lib $v"
    if [ $v = product ]; then STAGE=$stage LAYOUT=nhwc python3 scripts/kbench.py 20 apply_bwd; else STAGE=$stage LAYOUT=nhwc KBENCH_LIB=$v python3 scripts/kbench.py 20 apply_bwd; fi
  done
  echo "# stage $stage lean"; STAGE=$stage LAYOUT=nhwc python3 scripts/kbench.py 20 lean
done > $OUT/timings.txt 2>&1
tail -30 $OUT/summary.txt
