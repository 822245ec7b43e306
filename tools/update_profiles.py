#!/usr/bin/env python3
"""Copies the judged evidence of one tools/profile_pmc.sh run into profiles/:
the kernel-trace statistics, the per-kernel counter summary, and the HBM-side
traffic of the dominant kernel that bench.py reports as roofline.traffic.
Usage: update_profiles.py <gpurun_out/dir> <tag, e.g. r01> [spp_per_step]"""
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_wavefront_dragon1080p_kernel_stats.csv"))
summary = json.load(open(os.path.join(src, "summary.json")))
json.dump(summary, open(os.path.join(dst, f"{tag}_wavefront_dragon1080p_pmc_summary.json"), "w"), indent=1,
          sort_keys=True)
kernels = summary["kernels"]
name = max(("k_wf_drain", "k_wf_primary"), key=lambda k: kernels[k]["total_ms"])
k = kernels[name]
fetch_kb, write_kb = k["FETCH_SIZE_per_dispatch"], k["WRITE_SIZE_per_dispatch"]
out = {
    "kernel": name.replace("k_wf_", ""),
    "kernel_symbol": name,
    "scene": "dragon",
    "spp_per_step": spp,
    "mean_launch_us": k["mean_us"],
    "FETCH_SIZE_KB_per_launch": fetch_kb,
    "WRITE_SIZE_KB_per_launch": write_kb,
    # FETCH_SIZE tallies the 128-byte fabric requests of 16-byte-per-lane loads at 64 bytes on gfx950
    "hbm_bytes_per_launch": int(2 * fetch_kb * 1024 + write_kb * 1024),
    "l2_hit_rate": k["TCC_HIT_sum_per_dispatch"] / k["TCC_REQ_sum_per_dispatch"],
    "method": "rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum and --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum in separate "
              "passes (tools/profile_pmc.sh); FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B; "
              "uncalibrated for scattered loads); the counters sit on the L2's memory side and include "
              "Infinity-Cache hits, so this is traffic beyond L2, an upper bound on HBM traffic",
    "source": f"profiles/{tag}_wavefront_dragon1080p_pmc_summary.json",
}
json.dump(out, open(os.path.join(dst, f"{tag}_dominant_kernel_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
