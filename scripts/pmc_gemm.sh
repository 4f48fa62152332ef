#!/bin/bash
# MFMA-pipe utilisation of the own GEMM kernels on the isolated shapes of scripts/ksbench.py / wgradbench.py (one PMC pass
# each, kernel-trace only).  SQ_VALU_MFMA_BUSY_CYCLES counts 32 cycles per 32x32x16 bf16 MFMA, summed over the chip's 1024
# SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles: utilisation = BUSY / (GUI_ACTIVE / 8 * 1024).
# Usage: bash scripts/pmc_gemm.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc_gemm}
RAW=/tmp/pmc_gemm_$$
mkdir -p $OUT $RAW
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
NOSTOCK=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY --output-format csv -d $RAW/ks -- python3 scripts/ksbench.py 3 > $OUT/ks.log 2>&1
python3 - "$RAW" "$OUT" <<'PY'
import csv, glob, sys, collections, json
raw, out = sys.argv[1], sys.argv[2]
f = glob.glob(raw + "/ks/*/*_counter_collection.csv")
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if "kstream" not in r["Kernel_Name"] and "wgrad_kernel" not in r["Kernel_Name"]:
        continue
    key = ("kstream256" if "kstream256" in r["Kernel_Name"] else "kstream" if "kstream" in r["Kernel_Name"] else "wgrad") + "/grid" + r.get("Grid_Size", "")
    per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, v in sorted(per.items()):
    m = {c: sum(x) / len(x) for c, x in v.items()}
    gui = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    res[k] = {"launches": len(v.get("GRBM_GUI_ACTIVE", [])), "gui_cycles_per_xcd": gui,
              "mfma_util": m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024.0) if gui else None,
              "wait_inst_lds_over_wave_cycles": m.get("SQ_WAIT_INST_LDS", 0.0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else None,
              "wait_any_over_wave_cycles": m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else None}
json.dump(res, open(out + "/gemm_mfma.json", "w"), indent=1)
for k, v in res.items():
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
PY
