#!/bin/bash
# Round-4 kernel experiments on the GPU box: the fused forward statistics pass against apply_fwd (same
# 3N bytes) with SQ counters side by side, its occupancy variants, and the token backward with two rows of DMA look-ahead.
# RECORD of what was run (profiles/r04_notes.md sections 5, 6).  The two variants of the fused statistics pass were built with
# build-time switches that existed at commit 1ac855d only (-DMRLA_FUSED_MAXWAVES=4, -DMRLA_FUSED_WAVES_PER_EU=4 through
# scripts/build_variant.sh; the switches were removed from the source once measured); the token look-ahead switch
# (-DMRLA_TOKEN_BWD_DEPTH=2) existed from commit 1ac855d until it was removed after two measurements (no gain).
# Usage: bash scripts/r04_experiments.sh <outdir>
set -u
OUT=${1:-gpurun_out/r04_exp}; mkdir -p $OUT
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
export LAYOUT=nhwc
for v in product fused4w fusedwpe4; do
  lib=""; [ $v != product ] && lib=scripts/variants/libmrla_hip_$v.so
  for k in stats_fused apply_fwd; do
    [ $v != product ] && [ $k = apply_fwd ] && continue
    KBENCH_LIB=$lib python3 scripts/kbench.py 30 $k 2>&1 | sed "s/^/$v /" >> $OUT/kbench_fwd.txt
  done
done
python3 scripts/tokbench.py 100 >> $OUT/tokbench.txt 2>&1
KBENCH_LIB=scripts/variants/libmrla_hip_tokdepth2.so python3 scripts/tokbench.py 100 --check >> $OUT/tokbench.txt 2>&1
# SQ counters, stats_fwd_fused vs apply_fwd (product build), two passes of 8 counters
RAW=/tmp/r04_sq_$$; mkdir -p $RAW
for k in stats_fused apply_fwd; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $RAW/a_$k -- python3 scripts/kbench.py 3 $k > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --output-format csv -d $RAW/b_$k -- python3 scripts/kbench.py 3 $k > /dev/null 2>&1
  for p in a b; do python3 scripts/pmc_summarize.py $RAW/${p}_$k | grep -A1 "light_stats_fwd_fused\|light_apply_fwd" >> $OUT/sq_counters_fwd.txt; done
done
cat $OUT/kbench_fwd.txt $OUT/tokbench.txt
