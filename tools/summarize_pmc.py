#!/usr/bin/env python3
"""Summarises the CSVs written by tools/profile_pmc.sh: per kernel name, the
mean duration from the kernel trace and the sum of every counter over the
dispatches of the timed region. Usage: summarize_pmc.py <dir> [out.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    # the counting variants of the render kernels (instrumented passes of bench.py, slower by design) are kept
    # apart from the plain ones: k_wf_primary<mapping, two-level, counting>, k_render_mega<counting, two-level>
    for k, arg in (("k_wf_primary<", 2), ("k_render_mega<", 0)):
        if k in name:
            args = name.split(k, 1)[1].split(">", 1)[0].split(",")
            if len(args) > arg and args[arg].strip() == "true":
                return k[:-1] + "_counting"
    for k in ("k_wf_primary", "k_wf_drain", "k_wf_trace", "k_wf_shade", "k_wf_resolve", "k_render_mega",
              "k_trace", "k_nearest", "k_wide_emit", "k_flatten"):
        if k in name:
            return k
    return None


def main():
    d = sys.argv[1]
    out = {"kernels": defaultdict(dict)}
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
        dur = defaultdict(list)
        meta = {}
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                meta[k] = {x: r[x] for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                             "Scratch_Size", "Workgroup_Size_X", "Grid_Size_X")}
        for k, v in dur.items():
            out["kernels"][k].update(calls=len(v), mean_us=sum(v) / len(v) / 1e3, total_ms=sum(v) / 1e6, **meta[k])
    for f in glob.glob(os.path.join(d, "*", "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        n = defaultdict(lambda: defaultdict(int))
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                n[k][r["Counter_Name"]] += 1
        for k in acc:
            for c, v in acc[k].items():
                out["kernels"][k][c + "_per_dispatch"] = v / n[k][c]
    out["kernels"] = {k: v for k, v in out["kernels"].items()}
    text = json.dumps(out, indent=1, sort_keys=True)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
