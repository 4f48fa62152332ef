"""Section 8 of profiles/r06_notes.md: one row per committed bench line of the final library (profiles/r06_bench_*.json).
    python scripts/r06_table.py > /tmp/table.md"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def line(name):
    return json.loads(open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1])


def traffic_of(arch_batch, kernel):
    p = os.path.join(ROOT, "profiles", f"r06_pmc_traffic_{arch_batch}.json")
    if not os.path.exists(p):
        return "-"
    d = json.load(open(p))
    for k, v in d.items():
        if isinstance(v, dict) and k in kernel.replace("mrla_", "") and "hbm_bytes_per_launch" in v:
            return str(int(v["hbm_bytes_per_launch"]))
    return "-"


rows = []


def row(name, r, note="", pmc=None):
    rf = r.get("roofline") or {}
    cfg = r.get("config", {})
    tr = rf.get("traffic")
    if tr is None and pmc and rf.get("kernel"):
        tr = traffic_of(pmc, rf["kernel"].split("<")[0])
    rows.append(f"| {name} | {r['value']} | {r['ms_per_step']} | {r.get('eager_launch_ms_per_step', '-')} | "
                f"{cfg.get('replay_matches_eager', r.get('replay_matches_eager'))} | {rf.get('kernel', '-')} | {rf.get('avg_launch_us', '-')} | "
                f"{rf.get('frac', '-')} | {tr if tr is not None else '-'} | {note} |")


d = line("r06_bench_default_run.json")
row("resnet50_mrlal b256 bf16 (default run, `r06_bench_default_run.json`)", d, "the headline line")
row("resnet50_mrlal b256 bf16 (its own run, another box)", line("r06_bench_resnet50_mrlal.json"), pmc="resnet50_mrlal_b256")
row("resnet101_mrlab b128 bf16", line("r06_bench_resnet101_mrlab.json"), pmc="resnet101_mrlab_b128")
row("deit_mrlal_tiny b256 bf16", line("r06_bench_deit_mrlal_tiny.json"), pmc="deit_mrlal_tiny_patch16_224_b256")
row("deit_mrlab_tiny b256 bf16", line("r06_bench_deit_mrlab_tiny.json"), pmc="deit_mrlab_tiny_patch16_224_b256")
row("resnet50_mrlal b64 fp32 (no autocast)", line("r06_bench_resnet50_mrlal_fp32_b64.json"), "no counter pass at this size")
t = line("r06_bench_det_resnet50_mrlal_2x3x800x1344.json")
row("det_resnet50_mrlal 2x3x800x1344 bf16 (its own run, 20 steps)", t, f"eager restatement {t['eager_rocm']['fwd_bwd_ms']} ms")
det = d["detection_backbone"]
rows.append(f"| det_resnet50_mrlal 2x3x800x1344 bf16 (child of the default run) | {det['value']} | {det['ms_per_step']} | "
            f"{det['eager_launch_ms_per_step']} | {det['replay_matches_eager']} | - | - | - | - | eager restatement {det['eager_rocm']['fwd_bwd_ms']} ms |")
dp = d["config"]["drop_path_0"]
rows.append(f"| resnet50_mrlal b256 bf16, drop_path 0 (child) | {dp['value']} | {dp['ms_per_step']} | - | {dp['replay_matches_eager']} | - | - | - | - | 10 steps |")
for k, v in d["other_configs"].items():
    rf = v.get("roofline") or {}
    rows.append(f"| {k} (child of the default run) | {v['value']} | {v['ms_per_step']} | {v.get('eager_launch_ms_per_step', '-')} | "
                f"{v.get('replay_matches_eager')} | {rf.get('kernel', '-')} | {rf.get('avg_launch_us', '-')} | {rf.get('frac', '-')} | {rf.get('traffic', '-')} | 10 steps |")
print("| line | img/s | ms/step (graph) | ms/step (eager launches) | replay vs eager | roofline kernel | us/launch | frac | PMC bytes/launch | note |")
print("|---|---|---|---|---|---|---|---|---|---|")
print("\n".join(rows))
