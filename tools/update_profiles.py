#!/usr/bin/env python3
"""Copies the judged evidence of one tools/profile_pmc.sh run into profiles/:
the kernel-trace statistics, the per-kernel counter summary, and one entry per
render kernel in profiles/<tag>_pmc_index.json, which bench.py reads for
roofline.traffic and the counter-derived fields beside it.

Usage: update_profiles.py <gpurun_out/dir> <tag, e.g. r02> <label, e.g. dragon1080p>
           [--scene dragon --width 1920 --height 1080 --spp 256 --max-depth 32]
"""
import argparse
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402  (the build a profile belongs to)

ap = argparse.ArgumentParser()
ap.add_argument("src")
ap.add_argument("tag")
ap.add_argument("label")
ap.add_argument("--scene", default="dragon")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=256)
ap.add_argument("--max-depth", type=int, default=32)
ap.add_argument("--cus", type=int, default=256)
ap.add_argument("--two-level", action="store_true")
ap.add_argument("--regroup", action="store_true")
a = ap.parse_args()

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
stats = max(glob.glob(os.path.join(a.src, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
shutil.copy(stats, os.path.join(dst, f"{a.tag}_{a.label}_kernel_stats.csv"))
summary = json.load(open(os.path.join(a.src, "summary.json")))
summary_name = f"{a.tag}_{a.label}_pmc_summary.json"
json.dump(summary, open(os.path.join(dst, summary_name), "w"), indent=1, sort_keys=True)

index_path = os.path.join(dst, f"{a.tag}_pmc_index.json")
index = json.load(open(index_path)) if os.path.exists(index_path) else []
for symbol, short in (("k_wf_drain", "drain"), ("k_wf_primary", "primary"), ("k_render_mega", "megakernel")):
    k = summary["kernels"].get(symbol)
    if not k or "FETCH_SIZE_per_dispatch" not in k:
        continue
    fetch_kb, write_kb = k["FETCH_SIZE_per_dispatch"], k["WRITE_SIZE_per_dispatch"]
    # rocprofv3 reports GRBM_GUI_ACTIVE summed over the chip's 8 XCDs
    cycles = (k.get("GRBM_GUI_ACTIVE_per_dispatch") or 0.0) / 8.0
    entry = {
        "scene": a.scene, "width": a.width, "height": a.height, "spp_per_step": a.spp, "max_depth": a.max_depth,
        "two_level": a.two_level, "regroup": a.regroup, "kernel_source_hash": kernel_source_hash(),
        "kernel": short, "kernel_symbol": symbol, "mean_launch_us": k["mean_us"],
        "clock_ghz": round(cycles / (k["mean_us"] * 1e3), 3) if cycles else None,
        "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
        # FETCH_SIZE tallies the 128-byte fabric requests of 16-byte-per-lane loads at 64 bytes on gfx950
        "hbm_bytes_per_launch": int(2 * fetch_kb * 1024 + write_kb * 1024),
        "l2_hit_rate": round(k["TCC_HIT_sum_per_dispatch"] / k["TCC_REQ_sum_per_dispatch"], 4),
        # share of the vector L1's line accesses that it served itself
        "l1_hit_rate": round(1.0 - k["TCP_TCC_READ_REQ_sum_per_dispatch"] / k["TCP_TOTAL_CACHE_ACCESSES_sum_per_dispatch"], 4),
        # busy cycles summed over the CUs' texture-data / texture-address units against CUs x kernel cycles
        "td_busy": round(k["TD_TD_BUSY_sum_per_dispatch"] / (a.cus * cycles), 4) if cycles else None,
        "ta_busy": round(k["TA_TA_BUSY_sum_per_dispatch"] / (a.cus * cycles), 4) if cycles else None,
        # active lanes per VALU instruction issued, of 64
        "lane_util": round(k["SQ_THREAD_CYCLES_VALU_per_dispatch"] / (64.0 * k["SQ_ACTIVE_INST_VALU_per_dispatch"]), 4),
        # VALU wave-instructions issued against what the CUs' 4 SIMDs can issue (a wave64 instruction takes a
        # SIMD two cycles; quarter-rate instructions, which take longer, are counted as one like the rest)
        "valu_busy": round(k["SQ_INSTS_VALU_per_dispatch"] * 2.0 / (4.0 * a.cus * cycles), 4) if cycles else None,
        "vmem_wave_insts_per_launch": k.get("SQ_INSTS_VMEM_RD_per_dispatch"),
        "method": "rocprofv3 --pmc, one pass per counter group (tools/profile_pmc.sh); FETCH_SIZE doubled (gfx950 "
                  "counts 128-B requests as 64 B; uncalibrated for scattered loads); FETCH/WRITE_SIZE sit on the "
                  "L2's memory side and include Infinity-Cache hits: traffic beyond L2, an upper bound on HBM traffic",
        "source": f"profiles/{summary_name}",
    }
    index = [e for e in index if (e["scene"], e["width"], e["height"], e["spp_per_step"], e["max_depth"], e["kernel"],
                                  bool(e.get("two_level")), bool(e.get("regroup"))) !=
             (a.scene, a.width, a.height, a.spp, a.max_depth, short, a.two_level, a.regroup)]
    index.append(entry)
    print(json.dumps(entry, indent=1))
json.dump(index, open(index_path, "w"), indent=1)
