#!/bin/bash
# PMC passes over the light kernels (stage-1 shape dominates).  Usage on the GPU box: bash scripts/pmc_kbench.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p $OUT
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/sq -- python3 scripts/kbench.py 3 "${2:-}" > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 scripts/kbench.py 3 "${2:-}" > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 scripts/kbench.py 3 "${2:-}" > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/lds -- python3 scripts/kbench.py 3 "${2:-}" > $OUT/lds.log 2>&1
find $OUT -name "*counter_collection.csv" | head
