#!/bin/bash
# After `gpurun -- bash scripts/r06_final.sh default profiles` has merged its outputs back: copy the summaries that are judged into
# profiles/ under this round's names (gpurun_out/ is scratch).  Usage (in the container): bash scripts/r06_collect.sh
set -u
SRC=gpurun_out/r06_final
for cfg in resnet50_mrlal:256 resnet101_mrlab:128 deit_mrlal_tiny_patch16_224:256 deit_mrlab_tiny_patch16_224:256; do
  arch=${cfg%%:*}; batch=${cfg##*:}
  d=$SRC/prof/pmc_${arch}_b${batch}
  [ -f $d/summary_kernels.csv ] && cp $d/summary_kernels.csv profiles/r06_kernels_steady_state_${arch}_b${batch}.csv
  [ -f $d/summary_traffic.json ] && cp $d/summary_traffic.json profiles/r06_pmc_traffic_${arch}_b${batch}.json
  [ -f $d/kernel_stats_full.csv ] && cp $d/kernel_stats_full.csv profiles/r06_kernel_stats_${arch}_b${batch}.csv
  m=$SRC/prof/mfma_${arch}_b${batch}/mfma_summary.json
  [ -f $m ] && cp $m profiles/r06_pmc_mfma_whole_step_${arch}_b${batch}.json
  b=$SRC/prof/bench_${arch}.json
  short=${arch/_patch16_224/}
  [ -s $b ] && cp $b profiles/r06_bench_${short}.json
done
[ -s $SRC/bench_default_run.json ] && cp $SRC/bench_default_run.json profiles/r06_bench_default_run.json
ls -la profiles | grep r06_
