#!/bin/bash
# One-GPU A/B of the N > 1 code paths of bench.py through a one-rank RCCL group (`--ddp-probe`): the flat exchange inside
# the graph in both schedules (`--exchange after`: one all-reduce after backward; `--exchange overlap`: bucketed, sent from
# backward), the same launched eagerly, DistributedDataParallel launched eagerly, and no process group at all.  ONE rank has
# nobody to exchange with: this measures the OVERHEAD side of each schedule only (bench.py's `--exchange ab` default makes
# the same comparison on the real N ranks and reports it in the line).  Usage: bash scripts/dp_probe.sh [outdir]
OUT=${1:-gpurun_out/dp_probe}; mkdir -p $OUT
run() { name=$1; shift; python3 bench.py --no-baselines "$@" > $OUT/$name.out 2> $OUT/$name.err;
        python3 - $OUT/$name.out $name <<'PY'
import json, sys
rec = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
if not rec:
    print(sys.argv[2], "NO JSON LINE")
else:
    r = rec[0]
    print(f"{sys.argv[2]:14s} {r['value']:8.1f} img/s  {r['ms_per_step']:7.3f} ms/step  eager {r['eager_launch_ms_per_step']:7.3f}  | {r['config']['launch'][:70]} | {r['config'].get('gradient_exchange', '-')[:40]}")
PY
}
run flat_graph --ddp-probe --exchange after
run flat_overlap --ddp-probe --exchange overlap
run flat_eager --ddp-probe --graph 0 --dp flat
run ddp_eager --ddp-probe --dp ddp
run single
