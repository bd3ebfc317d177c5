"""Reduce rocprofv3 CSV output to small per-kernel summaries (run on the GPU box).

usage: python tools/summarize_prof.py <prof_dir> <out_dir>
Keeps: kernel_stats (top 25 rows), per-kernel average duration of our kernels from the kernel
trace, and per-kernel mean of every PMC counter for kernels whose name contains nsgp_/repre_.
"""
import csv
import glob
import os
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
OURS = ("nsgp_", "repre_", "nsgp::")


def short(name):
    for key in ("nsgp_project_v2_kernel", "nsgp_update_lr_kernel", "nsgp_lr_apply_kernel", "nsgp_lr_reduce_kernel", "nsgp_projector_head_kernel", "nsgp_project_kernel", "nsgp_update_kernel", "nsgp_project_single_kernel", "nsgp_projector_kernel",
                "rh_skinny_kernel", "rh_reduce_kernel", "rh_scores_kernel", "rh_dz_kernel", "rh_tn_kernel", "repre_replay_ce_fwd_kernel", "repre_replay_ce_bwd_kernel",
                "nsgp_cov_syrk_v2_kernel", "nsgp_cov_reduce_v2_kernel", "nsgp_cov_im2col_split_kernel", "nsgp_cov_syrk_kernel", "nsgp_cov_reduce_kernel", "nsgp_batch_mean_pad_kernel", "repre_sim_mask_kernel",
                "repre_row_norm_kernel", "repre_masked_sum_kernel"):
        if key in name:
            tail = name[name.find(key):]
            return tail[:80]
    return name[:80]


def suffix(f):      # the trace of the default bench command (training step) is kept beside the hot-path-only one
    return "_training_step" if "trace_e2e" in f else ""


for f in glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.reader(open(f)))
    with open(os.path.join(dst, f"kernel_stats_top{suffix(f)}.csv"), "w", newline="") as o:
        w = csv.writer(o)
        for r in rows[:26]:
            w.writerow([c[:120] for c in r])

for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    dur = defaultdict(list)
    rd = csv.DictReader(open(f))
    for r in rd:
        n = r.get("Kernel_Name", "")
        if any(k in n for k in OURS):
            dur[short(n)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    with open(os.path.join(dst, f"our_kernels_duration_us{suffix(f)}.csv"), "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(["kernel", "calls", "avg_us", "min_us", "max_us"])
        for k, v in sorted(dur.items()):
            w.writerow([k, len(v), sum(v) / len(v), min(v), max(v)])

agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    rd = csv.DictReader(open(f))
    for r in rd:
        n = r.get("Kernel_Name", "")
        if any(k in n for k in OURS):
            agg[short(n)][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(dst, "our_kernels_pmc_mean.csv"), "w", newline="") as o:
    w = csv.writer(o)
    w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
    for k in sorted(agg):
        for c in sorted(agg[k]):
            v = agg[k][c]
            w.writerow([k, c, len(v), sum(v) / len(v)])
print("summaries written to", dst)

# HBM traffic per launch of the dominant kernel, corrected as MI355X_MICROARCH.md (HBM section)
# prescribes: FETCH_SIZE is in KiB and reads exactly 1/2 of a wide (16 B/lane) coalesced stream on
# gfx950 -> x2; WRITE_SIZE (KiB) is exact for 16-B-per-lane / dword stores.
import json


def mean(v):
    return sum(v) / len(v)


traffic = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (kernels matching nsgp|repre only) of `python3 bench.py --steps 10 --warmup 2 "
                     "--hot-path-only` (tools/profile.sh); FETCH_SIZE x2 per the gfx950 correction"}
for k in agg:
    for tag, key in (("nsgp_project_v2_kernel<0>", "nsgp_project_kernel"), ("nsgp_update_lr_kernel<0, 1>", "nsgp_update_lr_kernel"),
                     ("nsgp_lr_apply_kernel<0, 128>", "nsgp_lr_apply_kernel"), ("nsgp_update_kernel<0>", "nsgp_update_kernel")):      # <SGD, common rank classes>
        if tag not in k:
            continue
        if "FETCH_SIZE" in agg[k] and "WRITE_SIZE" in agg[k]:
            f, w = mean(agg[k]["FETCH_SIZE"]), mean(agg[k]["WRITE_SIZE"])
            traffic[key + "_hbm_bytes_per_launch"] = f * 1024 * 2 + w * 1024
            traffic[key + "_fetch_size_kib_raw"], traffic[key + "_write_size_kib_raw"] = f, w
        if "SQ_VALU_MFMA_BUSY_CYCLES" in agg[k] and "GRBM_GUI_ACTIVE" in agg[k]:
            # 1024 SIMDs x cycles per XCD (GRBM_GUI_ACTIVE sums 8 XCDs)
            busy = mean(agg[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) / (1024 * mean(agg[k]["GRBM_GUI_ACTIVE"]) / 8)
            traffic["mfma_busy_fraction" if key == "nsgp_project_kernel" else key + "_mfma_busy_fraction"] = busy
            traffic["mfma_busy_note"] = "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)"
# every kernel of this library that both byte counters saw: HBM bytes per launch (same correction) and, where available, MFMA-busy
per_kernel = {}
for k in sorted(agg):
    rec = {}
    if "FETCH_SIZE" in agg[k] and "WRITE_SIZE" in agg[k]:
        rec["hbm_bytes_per_launch"] = mean(agg[k]["FETCH_SIZE"]) * 1024 * 2 + mean(agg[k]["WRITE_SIZE"]) * 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES" in agg[k] and "GRBM_GUI_ACTIVE" in agg[k] and mean(agg[k]["GRBM_GUI_ACTIVE"]) > 0:
        rec["mfma_busy_fraction"] = mean(agg[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) / (1024 * mean(agg[k]["GRBM_GUI_ACTIVE"]) / 8)
    if rec:
        per_kernel[k] = rec
if per_kernel:
    traffic["per_kernel"] = per_kernel
if len(traffic) > 1:
    traffic["kernel"] = "nsgp_project_v2_kernel<0> (dense path), nsgp_update_lr_kernel<0> / nsgp_lr_apply_kernel<0> / nsgp_update_kernel<0> (default path)"
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
