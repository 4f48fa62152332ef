#!/bin/bash
# isolated timings + SQ / traffic counters of the row-pipeline passes per stage shape (b = 256, bf16, NHWC) -- the table behind
# profiles/r05_notes.md section 9.  Usage on the GPU box: bash scripts/r05_sq.sh
set -u
OUT=gpurun_out/r05_sq
mkdir -p $OUT
export LAYOUT=nhwc
rm -f $OUT/counters.txt; python3 scripts/kbench.py 20 > $OUT/kbench.txt 2>&1
for stage in 0 1; do
export STAGE=$stage
for k in "apply_bwd+bn3sums" "stats_fused" "apply_fwd" "stats_bwd"; do
  d=$OUT/pmc_$(echo $k | tr -c 'a-z0-9_' '_')
  bash scripts/pmc_kbench.sh $d "$k" > /dev/null 2>&1
  for sub in sq lds fetch write; do echo "== stage $stage / $k / $sub"; python3 scripts/pmc_summarize.py $d/$sub | grep -A1 -E "light_(apply_bwd|stats_fwd_fused|apply_fwd|stats_bwd)_wide"; done >> $OUT/counters.txt
  rm -rf $d
done
done
unset STAGE
cat $OUT/kbench.txt | grep -v amdgpu
