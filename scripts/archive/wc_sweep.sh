#!/bin/bash
# Sweep of the workgroup shape of the NHWC row-pipeline passes (wc = neighbouring channel groups per workgroup).
# RECORD of what was run (profiles/r04_notes.md section 9): the build-time switch -DMRLA_WC_EXPERIMENT (MRLA_WC_<PASS>
# environment overrides inside wide_shape(), light_nhwc_wide.hip) existed at commit 6c5202f only, built with
#   bash scripts/build_variant.sh wcexp light_nhwc_wide.hip "-DMRLA_WC_EXPERIMENT -include cstdlib"
# and was removed once the shapes were chosen; the product has no such switch.
# Usage: [PASSES="APPLY_FWD:apply_fwd ..."] [WCS="1 2 4 8"] bash scripts/wc_sweep.sh <outdir>
set -u
OUT=${1:-gpurun_out/wc}; mkdir -p $OUT
export LAYOUT=nhwc KBENCH_LIB=scripts/variants/libmrla_hip_wcexp.so
for rep in 1 2; do
for wc in ${WCS:-1 2 4 8}; do
  for pk in ${PASSES:-STATS_FUSED:stats_fused STATS_FWD:stats_fwd APPLY_FWD:apply_fwd STATS_BWD:stats_bwd APPLY_BWD:apply_bwd+bn3sums}; do
    set -- ${pk%%:*} ${pk##*:}
    env MRLA_WC_$1=$wc python3 scripts/kbench.py 20 $2 2>/dev/null | grep "c=" | sed "s/^/wc=$wc /" >> $OUT/sweep.txt
  done
done
done
sort -k3,3n -k5,5 -k1,1 -s $OUT/sweep.txt
