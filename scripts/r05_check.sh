#!/bin/bash
set -u
OUT=gpurun_out/r05_check
mkdir -p $OUT
python3 -m pytest tests/test_light_gpu.py tests/test_tokens_gpu.py tests/test_random_shapes_gpu.py tests/test_sequences_gpu.py tests/test_base_gpu.py tests/test_fullsize_properties_gpu.py tests/test_block_bf16_gpu.py -q -m gpu > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
LAYOUT=nhwc python3 scripts/kbench.py 20 gate_bwd 2>&1 | grep gate_bwd
