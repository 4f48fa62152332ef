cp mrla_amd/libmrla_hip.so /tmp/product_libmrla_hip.so && trap 'cp /tmp/product_libmrla_hip.so mrla_amd/libmrla_hip.so' EXIT   # 
set -u
mkdir -p gpurun_out/nts
cp mrla_amd/libmrla_hip.so /tmp/product.so
run() { python3 bench.py --steps 30 --warmup 8 --no-baselines --no-others --no-forward-only 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['avg_launch_us'], r['path_ms_per_step'])" | tee -a gpurun_out/nts/ab.txt; }
run product_warm
for rep in 1 2; do
  for v in product nts_fwd nts_bwd nts_all; do
    if [ $v = product ]; then cp /tmp/product.so mrla_amd/libmrla_hip.so; else cp scripts/variants/libmrla_hip_$v.so mrla_amd/libmrla_hip.so; fi
    run $v
  done
done
