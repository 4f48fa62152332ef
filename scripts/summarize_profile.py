"""Condense rocprofv3 outputs of scripts/pmc_bench.sh into small committed summaries:
  <out>/summary_kernels.csv   steady-state per-kernel time per step (from the kernel trace, warm-up/find excluded)
  <out>/summary_traffic.json  HBM bytes per launch of the MRLA kernels: 2*FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE
                              reports half of a wide coalesced read, MI355X_MICROARCH.md section HBM), units of KB -> bytes
"""
import collections
import csv
import glob
import json
import os
import sys

raw = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else raw


def short(name):
    if "token_fwd_rows" in name:            # <T, VALUE>: VALUE = true is the MRLA-base token module's value forward
        return "token_value_fwd" if (", true>" in name or "Lb1EE" in name) else "token_apply_fwd"
    if "base_value_bwd_wide" in name:
        return "base_value_bwd"
    if "token_stats_vec_kernel" in name:
        return "token_stats_kernel"
    if "token_apply_bwd_rows" in name:      # <T, RAGGED, BASE>: the MRLA-base token module's value backward is BASE = true
        return "token_base_value_bwd" if (", true>" in name or "Lb1EEv" in name) else "token_apply_bwd"
    for key in ("light_stats_fwd_fused", "conv1x1_kstream", "conv1x1_wide", "conv1x1_fwd", "conv1x1_wgrad_reduce", "conv1x1_wgrad", "weight_bank",
                "plain_bn_fwd_rec", "reduce_rows2", "light_stats_fwd", "light_apply_fwd_pre", "light_apply_fwd", "light_stats_bwd", "light_apply_bwd", "plane_moments_small",
                "plane_moments", "affine_act", "nhwc_moments_flat", "nhwc_affine_flat", "nhwc_moments", "nhwc_affine",
                "base_combine_nhwcIDF16bDF16bLi0", "base_combine_nhwcIDF16bDF16bLi1", "base_combine_nhwcIffLi0",
                "base_combine_nhwcIffLi1", "base_combine", "base_attend_fwd", "token_apply_fwd", "token_apply_bwd", "token_value_fwd", "token_stats_kernel",
                "token_pool_kernel", "token_cls_fwd", "token_ln_bwd", "token_norm_pool", "token_ln",
                "base_attend_bwd", "base_value_bwd", "base_tail", "base_pmom", "base_gate_fwd", "base_gate_bwd",
                "gate_fwd_kernel", "gate_bwd_kernel", "bn_fwd_kernel", "bn_bwd_kernel", "bn_relu_pool_fwd",
                "bn_relu_pool_dmoments", "bn_relu_pool_bwd",
                "plain_bn_fwd", "plain_bn_bwd", "reduce_rows"):
        if key in name:
            return key
    return name[:80]


def pmc(kind, counter):
    f = glob.glob(os.path.join(raw, kind, "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(list)
    if not f:
        return agg
    for r in csv.DictReader(open(f[0])):
        if "mrla" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
            if "conv1x1" in r["Kernel_Name"]:       # per shape as well (grid size identifies it)
                agg[short(r["Kernel_Name"]) + "/grid" + r.get("Grid_Size", "")].append(float(r["Counter_Value"]))
    return agg


fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
traffic = {}
for k in sorted(set(fetch) | set(write)):
    fl, wl = fetch.get(k, []), write.get(k, [])
    if not fl or not wl:
        continue
    traffic[k] = {"launches_profiled": len(fl), "fetch_kb_per_launch_raw": sum(fl) / len(fl),
                  "write_kb_per_launch": sum(wl) / len(wl),
                  "hbm_bytes_per_launch": (2.0 * sum(fl) / len(fl) + sum(wl) / len(wl)) * 1024.0}
# the library these counters were measured on (bench.py quotes them only for the same build: roofline.traffic_stale)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lib_identity  # noqa: E402
traffic["_meta"] = dict(lib_identity.identity(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")),
                        command=os.environ.get("PMC_COMMAND", "scripts/pmc_bench.sh"))
json.dump(traffic, open(os.path.join(out, "summary_traffic.json"), "w"), indent=1)

f = glob.glob(os.path.join(raw, "trace", "*", "*_kernel_trace.csv"))
if f:
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
    # one launch per training step: the loss kernel (any architecture), else ResNet's max-pool backward
    marks = [i for i, r in enumerate(rows) if "nll_loss_forward" in r["Kernel_Name"]]
    if len(marks) < 5:
        marks = [i for i, r in enumerate(rows) if "max_pool_backward" in r["Kernel_Name"]]
    if len(marks) >= 5:
        sel, steps = rows[marks[-5]:marks[-1]], 4
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in sel:
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        wall = (int(rows[marks[-1]]["Start_Timestamp"]) - int(rows[marks[-5]]["Start_Timestamp"])) / 1e6 / steps
        with open(os.path.join(out, "summary_kernels.csv"), "w") as g:
            g.write(f"# steady state over {steps} steps; GPU kernel time {sum(v[1] for v in agg.values()) / 1e3 / steps:.2f} "
                    f"ms/step, wall {wall:.2f} ms/step (profiler attached)\nkernel,calls_per_step,ms_per_step,avg_us\n")
            for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
                g.write(f"\"{k}\",{c / steps:.1f},{us / 1e3 / steps:.3f},{us / c:.2f}\n")
print(open(os.path.join(out, "summary_traffic.json")).read()[:1500])
