#!/bin/bash
# SQ counters of the DeiT token kernels inside the real bench step.  Usage on the GPU box: bash scripts/pmc_tokens.sh
set -u
RAW=/tmp/pmc_tok_$$
mkdir -p $RAW
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $RAW/a -- python3 bench.py --arch deit_mrlal_tiny_patch16_224 --steps 2 --warmup 2 --no-baselines > $RAW/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM --output-format csv -d $RAW/b -- python3 bench.py --arch deit_mrlal_tiny_patch16_224 --steps 2 --warmup 2 --no-baselines > $RAW/b.log 2>&1
python3 scripts/pmc_summarize.py $RAW/a | grep -A1 "token_"
python3 scripts/pmc_summarize.py $RAW/b | grep -A1 "token_"
