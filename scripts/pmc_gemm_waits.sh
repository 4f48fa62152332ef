RAW=/tmp/pmc_gemm2_$$; mkdir -p $RAW
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NOSTOCK=1 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $RAW/ks -- python3 scripts/ksbench.py 3 > /dev/null 2>&1
python3 - $RAW <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/ks/*/*_counter_collection.csv")
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if "kstream" not in r["Kernel_Name"]: continue
    key = ("kstream256" if "kstream256" in r["Kernel_Name"] else "kstream") + "/grid" + r.get("Grid_Size", "")
    per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(per.items()):
    m = {c: sum(x) / len(x) for c, x in v.items()}
    w = m["SQ_WAVE_CYCLES"]
    print(k, {c: round(x / w, 3) for c, x in m.items() if c != "SQ_WAVE_CYCLES"})
PY
