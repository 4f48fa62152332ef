#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_flaky
for i in 1 2; do python3 -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r05_flaky/run_$i.txt 2>&1; tail -3 gpurun_out/r05_flaky/run_$i.txt; done
for i in 1 2 3; do python3 -m pytest tests/test_ddp_gpu.py tests/test_graph_replay_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/r05_flaky/ddp_$i.txt 2>&1; tail -2 gpurun_out/r05_flaky/ddp_$i.txt; done
python3 -c "import __graft_entry__ as g; g.smoke()"
