#!/bin/bash
# Per-bounce cost of the headline configuration: runs one step with every bounce
# as its own traversal / shading launch pair under rocprofv3 --kernel-trace and
# prints the duration of each launch in order (the populations shrink from
# bounce to bounce; what is left of the time does not).
# Usage (on the GPU box): bash tools/bounce_probe.sh [bench args...]
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/bounce_probe"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$GRAFT_REPO_ROOT/bench.py" \
  --steps 1 --warmup 0 --no-cpu-baseline --wavefront-bounces 31 "$@" > "$OUT/bench.log" 2>&1 || { tail -5 "$OUT/bench.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "k_wf_trace" in r["Kernel_Name"] or "k_wf_shade" in r["Kernel_Name"] or "k_wf_drain" in r["Kernel_Name"] or "k_wf_primary<" in r["Kernel_Name"]]
# the last step's launches: everything after the last non-counting primary launch
last = max(i for i, r in enumerate(sel) if "k_wf_primary" in r["Kernel_Name"])
for r in sel[last:]:
    name = r["Kernel_Name"].split("::")[-1].split("(")[0]
    print(f"{name:28s} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:9.3f} ms")
PY
