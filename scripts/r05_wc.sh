#!/bin/bash
set -u
OUT=gpurun_out/r05_wc
mkdir -p $OUT
export LAYOUT=nhwc
for stage in 2 3; do
  export STAGE=$stage
  d=$OUT/pmc_s$stage
  bash scripts/pmc_kbench.sh $d "apply_bwd+bn3sums" > /dev/null 2>&1
  for sub in fetch write; do python3 scripts/pmc_summarize.py $d/$sub | grep -A1 "apply_bwd_wide" | tail -1 | sed "s/^/stage $stage $sub: /"; done
  rm -rf $d
done
