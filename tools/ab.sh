#!/bin/bash
# A/B of builds of the library on one box: alternates bench.py between the
# builds path_tracer_amd/_ab/libptcore_<v>.so for v in $AB_VARIANTS (default
# "a b"), three rounds. Usage (on the GPU box): bash tools/ab.sh [bench args...]
set -u
for round in 1 2 3; do
  for v in ${AB_VARIANTS:-a b}; do
    PTC_LIBRARY=$GRAFT_REPO_ROOT/path_tracer_amd/_ab/libptcore_$v.so timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), d['roofline']['phase_ms_per_step'])" || exit 1
  done
done
