#!/usr/bin/env python3
"""Registers, scratch, LDS and occupancy of every kernel in a HIP source, from the
compiler's own remarks (-Rpass-analysis=kernel-resource-usage).
Usage: tools/kernel_resources.py path_tracer_amd/csrc/ptc_wavefront.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
       "-fno-slp-vectorize", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
text = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in text.splitlines():
    m = re.search(r"remark: (?:\[[^\]]*\])?\s*(.*)", line)
    if not m:
        continue
    body = m.group(1).strip()
    if body.startswith("Function Name:"):
        cur = body.split(":", 1)[1].split("[-R")[0].strip()
        rows[cur] = {}
    elif cur and ":" in body:
        k, v = body.split(":", 1)
        rows[cur][k.strip()] = v.split("[-R")[0].strip()
for name, r in rows.items():
    demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in demangled:
        continue
    short = re.sub(r"\(ptc::DScene.*", "", demangled).replace("ptc::(anonymous namespace)::", "").replace("void ", "")
    print(f"{short:55s} vgpr {r.get('VGPRs'):>4s} agpr {r.get('AGPRs'):>3s} sgpr {r.get('TotalSGPRs'):>4s} "
          f"scratch {r.get('ScratchSize [bytes/lane]'):>5s} occupancy {r.get('Occupancy [waves/SIMD]'):>2s} "
          f"lds {r.get('LDS Size [bytes/block]')}")
