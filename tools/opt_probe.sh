#!/bin/bash
# Reinsertion rounds against tree quality and render rate: bench.py per value of
# PTC_OPT_ROUNDS (the measurement override of the builder's default).
# Usage (on the GPU box): bash tools/opt_probe.sh "-1 2 4 8"  (-1: no rounds; 0 would mean the default) [bench args...]
set -u
rounds=$1; shift
for r in $rounds; do
  PTC_OPT_ROUNDS=$r timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null |
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; print('rounds $r', 'Mrays/s', round(d['value'],1), 'build_ms', c['bvh_build_ms'], 'nodes', c['bvh_nodes'], 'SAH', c['bvh_sah_cost'], 'nodes/ray', r['nodes_per_ray'], 'tris/ray', r['tris_per_ray'], r['phase_ms_per_step'])" || exit 1
done
