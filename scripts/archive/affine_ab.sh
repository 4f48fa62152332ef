#!/bin/bash
# In-step A/B of the workgroup count of nhwc_affine_flat (BatchNorm(+ReLU) apply, forward and backward): the product (about
# 4096 workgroups walking `iters` 4 KB chunks each) against variants with more, shorter workgroups (working-tree switch
# -DMRLA_AFFINE_WGS=<n>, scripts/build_variant.sh).  RECORD of what was run (profiles/r04_notes.md section 9).
cp mrla_amd/libmrla_hip.so /tmp/product_libmrla_hip.so && trap 'cp /tmp/product_libmrla_hip.so mrla_amd/libmrla_hip.so' EXIT   # 
set -u
mkdir -p gpurun_out/aff
python3 scripts/bnbench.py 30 2>/dev/null | tail -14 | sed "s/^/product /" > gpurun_out/aff/bnbench.txt
for v in "$@"; do KBENCH_LIB=scripts/variants/libmrla_hip_$v.so python3 scripts/bnbench.py 30 2>/dev/null | tail -14 | sed "s/^/$v /" >> gpurun_out/aff/bnbench.txt; done
cp mrla_amd/libmrla_hip.so /tmp/product.so
run() { python3 bench.py --steps 30 --warmup 8 --no-baselines --no-others --no-forward-only 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', d['value'], d['ms_per_step'])" | tee -a gpurun_out/aff/ab.txt; }
run product_warm
for rep in 1 2; do
  for v in product "$@"; do
    if [ $v = product ]; then cp /tmp/product.so mrla_amd/libmrla_hip.so; else cp scripts/variants/libmrla_hip_$v.so mrla_amd/libmrla_hip.so; fi
    run $v
  done
done
cat gpurun_out/aff/bnbench.txt
