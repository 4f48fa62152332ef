#!/bin/bash
# Round-6 measurement set on the GPU box: the GPU suite (fresh parity log), the driver's own bench command, and the per-config
# bench / trace / counter passes.  Usage: bash scripts/r06_final.sh [suite] [default] [profiles]
set -u
OUT=gpurun_out/r06_final
mkdir -p $OUT
for what in "$@"; do
  case $what in
    suite)
      rm -f gpurun_out/parity_maxima.jsonl
      python3 -m pytest tests -q -m gpu > $OUT/pytest_all.txt 2>&1; tail -4 $OUT/pytest_all.txt ;;
    default)
      python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default_run.json 2> $OUT/bench_default_run.err
      echo "default rc=$?"; tail -2 $OUT/bench_default_run.err; head -c 400 $OUT/bench_default_run.json; echo ;;
    profiles)
      bash scripts/final_profiles.sh $OUT/prof full resnet50_mrlal:256 resnet101_mrlab:128 deit_mrlal_tiny_patch16_224:256 deit_mrlab_tiny_patch16_224:256 ;;
  esac
done
