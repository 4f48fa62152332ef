#!/bin/bash
# Row ranges of the detection batch (light_nhwc_wide.h: MRLA_CUT_MIN_ROWS rows per range at least, towards MRLA_CUT_TARGET_WGS
# workgroups per launch): the detection backbone's step with the product library and with variant builds
# (scripts/build_variant.sh cut_<rows>_<wgs> light_nhwc_wide.hip "-DMRLA_CUT_MIN_ROWS=.. -DMRLA_CUT_TARGET_WGS=.."), two rounds.
# Usage on the GPU box: bash scripts/r06_cut_sweep.sh <out.txt>
set -u
OUT=${1:-gpurun_out/r06_cut_sweep.txt}
: > $OUT
for round in 1 2; do
  for lib in product $(ls scripts/variants/libmrla_hip_cut_*.so 2>/dev/null); do
    if [ $lib = product ]; then unset MRLA_HIP_LIB; else export MRLA_HIP_LIB=$PWD/$lib; fi
    python3 bench.py --arch det_resnet50_mrlal --shape 2x3x800x1344 --steps 10 --warmup 3 --no-baselines 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['mrla_kernels']
print('round $round', '$lib'.split('/')[-1], 'ms/step', d['ms_per_step'], 'eager', d.get('eager_launch_ms_per_step'), 'replay', d['config'].get('replay_matches_eager'),
      {n.replace('mrla_light_',''): k[n]['ms_per_step'] for n in k if 'light_stats' in n or 'light_apply' in n})" >> $OUT
  done
done
cat $OUT
