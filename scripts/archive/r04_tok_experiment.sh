# RECORD of the token-backward look-ahead experiment (profiles/r04_notes.md section 6): the tokdepth2 variant was built with
# -DMRLA_TOKEN_BWD_DEPTH=2, a build-time switch tokens_nhwc.hip carried from commit 1ac855d until it was removed (no gain).
set -u
OUT=gpurun_out/r04_exp; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
KBENCH_LIB=scripts/variants/libmrla_hip_tokdepth2.so python3 scripts/tokbench.py 100 --check > $OUT/tokbench2.txt 2>&1
RAW=/tmp/tok_sq_$$; mkdir -p $RAW
for v in product tokdepth2; do
  lib=""; [ $v != product ] && lib=scripts/variants/libmrla_hip_$v.so
  export KBENCH_LIB=$lib
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $RAW/a_$v -- python3 scripts/tokbench.py 3 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --output-format csv -d $RAW/b_$v -- python3 scripts/tokbench.py 3 > /dev/null 2>&1
  echo "== $v" >> $OUT/sq_counters_tok.txt
  for p in a b; do python3 scripts/pmc_summarize.py $RAW/${p}_$v | grep -A1 "token_apply_bwd" >> $OUT/sq_counters_tok.txt; done
done
cat $OUT/tokbench2.txt; cat $OUT/sq_counters_tok.txt
