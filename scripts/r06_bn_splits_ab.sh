#!/bin/bash
# BatchNorm moment pieces per image beyond 64 (bnact_nhwc.hip: MRLA_BN_SPLITS_MAX) on the detection backbone's step: product
# against a variant capped at 64 (scripts/build_variant.sh bn64 bnact_nhwc.hip "-DMRLA_BN_SPLITS_MAX=64"), every kernel's ms per step.
for round in 1 2; do
  for lib in product scripts/variants/libmrla_hip_bn64.so; do
    if [ $lib = product ]; then unset MRLA_HIP_LIB; else export MRLA_HIP_LIB=$PWD/$lib; fi
    python3 bench.py --arch det_resnet50_mrlal --shape 2x3x800x1344 --steps 10 --warmup 3 --no-baselines 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['mrla_kernels']
print('round $round', '$lib'.split('/')[-1], 'ms/step', d['ms_per_step'], 'eager', d.get('eager_launch_ms_per_step'), 'events', d.get('eager_launch_with_kernel_events_ms_per_step'),
      {n.replace('mrla_',''): k[n]['ms_per_step'] for n in k})"
  done
done
