#!/bin/bash
# Rows of DMA look-ahead in the 3N passes (-DMRLA_ROW_DEPTH=2 / 3 variants through scripts/build_variant.sh): isolated kernel
# times per ResNet-50 stage, then the parity tests of the light path on the variant library.
# Usage: bash scripts/depth_experiment.sh <outdir> "<kernels>" <variant> ...
set -u
OUT=${1:-gpurun_out/depth}; KERNELS=$2; shift 2; mkdir -p $OUT
export LAYOUT=nhwc
for v in product "$@"; do
  lib=""; [ $v != product ] && lib=scripts/variants/libmrla_hip_$v.so
  for k in $KERNELS; do
    KBENCH_LIB=$lib python3 scripts/kbench.py 30 $k 2>&1 | grep -v "^per-kernel" | sed "s/^/$v /" >> $OUT/kbench.txt
  done
done
cat $OUT/kbench.txt
for v in "$@"; do
  cp scripts/variants/libmrla_hip_$v.so mrla_amd/libmrla_hip.so
  python3 -m pytest tests/test_light_gpu.py tests/test_random_shapes_gpu.py -x -q -m gpu -k "nhwc or cl" 2>&1 | tail -3 | sed "s/^/$v /" | tee -a $OUT/tests.txt
done
