#!/bin/bash
# Whole-step MFMA utilisation (MIOpen's convolutions): one PMC pass, kernel-trace only.  Usage: bash scripts/pmc_mfma.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc_mfma}
RAW=/tmp/pmc_mfma_$$
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $RAW/m -- python3 bench.py --steps 2 --warmup 2 --no-baselines > $OUT/mfma.log 2>&1
python3 - "$RAW" "$OUT" <<'PY'
import csv, glob, sys, collections, json
raw, out = sys.argv[1], sys.argv[2]
f = glob.glob(raw + "/m/*/*_counter_collection.csv")
agg = collections.defaultdict(float); per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f[0])):
    agg[r["Counter_Name"]] += float(r["Counter_Value"])
    per[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
res = {"totals": dict(agg)}
if agg.get("SQ_BUSY_CU_CYCLES"):
    res["mfma_busy_over_cu_busy"] = agg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / agg["SQ_BUSY_CU_CYCLES"]
top = sorted(per.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))[:8]
res["top_mfma_kernels"] = {k: dict(v) for k, v in top}
json.dump(res, open(out + "/mfma_summary.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "top_mfma_kernels"}, indent=1))
PY
