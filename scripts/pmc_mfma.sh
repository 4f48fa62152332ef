#!/bin/bash
# Whole-step MFMA utilisation (MIOpen's convolutions + this build's 1x1 GEMMs): one PMC pass, kernel-trace only; warm-up
# launches (MIOpen's solver search) excluded by keeping the second half of the dispatches only.
# Usage: [BENCH_ARGS="--arch .. --batch .."] bash scripts/pmc_mfma.sh <outdir>     -> <outdir>/mfma_summary.json
set -u
OUT=${1:-gpurun_out/pmc_mfma}
RAW=/tmp/pmc_mfma_$$
mkdir -p $OUT $RAW
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
ARGS=${BENCH_ARGS:-}
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $RAW/m -- python3 bench.py $ARGS --steps 2 --warmup 2 --no-baselines --no-forward-only --benchmark 0 --graph 0 > $OUT/mfma.log 2>&1
python3 - "$RAW" "$OUT" $ARGS <<'PY'
import csv, glob, sys, collections, json, os
raw, out = sys.argv[1], sys.argv[2]
sys.path.insert(0, "scripts")
import lib_identity
f = glob.glob(raw + "/m/*/*_counter_collection.csv")
agg = collections.defaultdict(float); per = collections.defaultdict(lambda: collections.defaultdict(float))
rows = list(csv.DictReader(open(f[0])))
# steady state: the timed region is the LAST 2 of the 4 steps the command runs (2 warm-up + 2 timed, launched kernel by
# kernel with --graph 0 so that every kernel is a dispatch the counters see); keep the second half of the dispatches
ids = sorted({int(r["Dispatch_Id"]) for r in rows})
cut = ids[len(ids) // 2]
for r in rows:
    if int(r["Dispatch_Id"]) < cut:
        continue
    agg[r["Counter_Name"]] += float(r["Counter_Value"])
    per[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
res = {"command": "bench.py " + " ".join(sys.argv[3:]) + " --steps 2 --warmup 2 --no-baselines --benchmark 0 --graph 0",
       "dispatches_counted": len([i for i in ids if i >= cut]), "totals": dict(agg),
       "_meta": lib_identity.identity(os.getcwd())}
if agg.get("SQ_BUSY_CU_CYCLES"):
    res["mfma_busy_over_cu_busy"] = agg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / agg["SQ_BUSY_CU_CYCLES"]
if agg.get("GRBM_GUI_ACTIVE"):
    # SQ_VALU_MFMA_BUSY_CYCLES sums over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles
    res["mfma_busy_over_gpu_active_all_simds"] = agg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (agg["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
top = sorted(per.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))[:8]
res["top_mfma_kernels"] = {k: dict(v) for k, v in top}
json.dump(res, open(out + "/mfma_summary.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "top_mfma_kernels"}, indent=1))
PY
