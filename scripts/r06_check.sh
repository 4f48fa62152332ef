#!/bin/bash
# Round-6 verification batch on one GPU box: the new / changed test files, then bench.py in its new parts.
mkdir -p gpurun_out
python -m pytest tests/test_lean_gpu.py tests/test_graph_replay_gpu.py tests/test_conv1x1_gpu.py tests/test_tokens_gpu.py tests/test_models_gpu.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r06_check_tests.txt
python -m pytest tests/test_ddp_gpu.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r06_check_ddp.txt
python bench.py --steps 10 --warmup 3 --no-others > gpurun_out/r06_check_bench.json 2> gpurun_out/r06_check_bench.err
python bench.py --autocast none --batch 64 --steps 10 --warmup 3 --no-baselines > gpurun_out/r06_check_bench_fp32_b64.json 2> gpurun_out/r06_check_bench_fp32.err
python bench.py --arch det_resnet50_mrlal --shape 2x3x800x1344 --steps 5 --warmup 2 --no-baselines > gpurun_out/r06_check_bench_det.json 2> gpurun_out/r06_check_bench_det.err
for b in 32 64; do python scripts/miopen_bwd_graph_probe.py 1 $b 0 fp32; done > gpurun_out/r06_check_miopen_fp32.txt 2>&1
tail -3 gpurun_out/r06_check_tests.txt gpurun_out/r06_check_ddp.txt
tail -c 400 gpurun_out/r06_check_bench.err
