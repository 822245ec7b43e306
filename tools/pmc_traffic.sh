#!/bin/bash
# Traffic beyond L2 and L2 hit rate of bench.py's kernels for each build in
# path_tracer_amd/_ab/libptcore_<v>.so, v in $AB_VARIANTS: two counter passes per build.
# Usage (on the GPU box): AB_VARIANTS="a b" bash tools/pmc_traffic.sh <out-dir-under-gpurun_out> [bench args...]
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
ARGS="--steps 2 --warmup 1 --no-cpu-baseline $*"
cd /tmp && export TMPDIR=/tmp
for v in ${AB_VARIANTS:-a b}; do
  export PTC_LIBRARY=$GRAFT_REPO_ROOT/path_tracer_amd/_ab/libptcore_$v.so
  mkdir -p "$OUT/$v"
  timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$v/trace" -- python3 "$GRAFT_REPO_ROOT/bench.py" $ARGS > "$OUT/$v/trace.log" 2>&1 || exit 1
  timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d "$OUT/$v/tcc1" -- python3 "$GRAFT_REPO_ROOT/bench.py" $ARGS > "$OUT/$v/tcc1.log" 2>&1 || exit 1
  timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/$v/tcc2" -- python3 "$GRAFT_REPO_ROOT/bench.py" $ARGS > "$OUT/$v/tcc2.log" 2>&1 || exit 1
  python3 "$GRAFT_REPO_ROOT/tools/summarize_pmc.py" "$OUT/$v" "$OUT/$v/summary.json" > /dev/null || exit 1
  python3 - "$OUT/$v/summary.json" "$v" <<'PY'
import json, sys
s = json.load(open(sys.argv[1]))
for name in ("k_wf_drain", "k_wf_primary"):
    k = s["kernels"].get(name)
    if k and "FETCH_SIZE_per_dispatch" in k:
        gb = (2 * k["FETCH_SIZE_per_dispatch"] + k["WRITE_SIZE_per_dispatch"]) * 1024 / 1e9
        hit = k["TCC_HIT_sum_per_dispatch"] / max(k["TCC_REQ_sum_per_dispatch"], 1)
        print(sys.argv[2], name, f"{k['mean_us'] / 1e3:.2f} ms, {gb:.1f} GB beyond L2 per launch, L2 hit rate {hit:.3f}")
PY
done
