#!/bin/bash
# Which of the two backward apply passes does the vector pipe bound?  Both forms with and without their dWv sums (63 vector
# instructions per row step; -DMRLA_EXP_SKIP_WG=1 variants built by scripts/build_variant.sh: wrong results, right timing),
# isolated launches through the C ABI, alternating, two rounds.  Usage on the GPU box: bash scripts/r06_skipwg.sh <out.txt>
set -u
OUT=${1:-gpurun_out/r06_skipwg.txt}
V=scripts/variants
for round in 1 2; do
  for stage in 0 1 2 3; do
    echo "# round $round stage $stage stored, product"; STAGE=$stage LAYOUT=nhwc python3 scripts/kbench.py 20 "apply_bwd+bn3sums" | grep -v "^#"
    echo "# round $round stage $stage stored, no dWv sums"; STAGE=$stage LAYOUT=nhwc KBENCH_LIB=$V/libmrla_hip_skipwg_wide.so python3 scripts/kbench.py 20 "apply_bwd+bn3sums" | grep -v "^#"
    echo "# round $round stage $stage lean, product"; STAGE=$stage LAYOUT=nhwc python3 scripts/kbench.py 20 "lean apply_bwd" | grep -v "^#"
    echo "# round $round stage $stage lean, no dWv sums"; STAGE=$stage LAYOUT=nhwc KBENCH_LIB=$V/libmrla_hip_skipwg_lean.so python3 scripts/kbench.py 20 "lean apply_bwd" | grep -v "^#"
  done
done > $OUT 2>&1
cat $OUT
