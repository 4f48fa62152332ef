#!/bin/bash
# Row ranges at small classification batches (resnet50_mrlal, bf16): bench.py with the row ranges as shipped (mode 0) and off
# (mrla_tuning_row_ranges(1)), alternating.  Usage on the GPU box: bash scripts/r06_small_batch.sh <out.txt>
set -u
OUT=${1:-gpurun_out/r06_small_batch.txt}
: > $OUT
for b in 16 32 64; do
  for mode in 0 1 0 1; do
    python3 -c "
import sys, runpy
import torch
torch.cuda.init()
sys.path.insert(0, '.')
from mrla_amd import _lib
_lib.load().mrla_tuning_row_ranges($mode)
sys.argv = ['bench.py', '--batch', '$b', '--steps', '20', '--warmup', '5', '--no-baselines']
runpy.run_path('bench.py', run_name='__main__')
" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['mrla_kernels']
print('batch $b mode $mode', 'img/s', d['value'], 'ms/step', d['ms_per_step'], 'eager', d.get('eager_launch_ms_per_step'), 'replay', d['config'].get('replay_matches_eager'),
      {n.replace('mrla_light_',''): k[n]['ms_per_step'] for n in k if 'light_stats' in n or 'light_apply' in n})" >> $OUT
  done
done
cat $OUT
