#!/bin/bash
# Counter profiles of the round's configurations: the headline, the 10 M-triangle
# scene, and the four whose step the persistent launch dominates (the reference's
# four-instance table flattened and two-level, the mesh inside the Cornell box at
# 4K, the Cornell box alone). tools/collect_r03.sh then files the summaries under
# profiles/ (here, on the CPU side), and tools/bench_r03.sh, in a second GPU call,
# writes the default bench line and the side table with their counter fields.
# Usage (on the GPU box): bash tools/profile_r03.sh [tag [a|b]]
set -u
TAG=${1:-r03}
one() { # label, bench args...
  local label=$1; shift
  bash tools/profile_pmc.sh ${TAG}_$label "$@" &&
  python3 tools/summarize_pmc.py gpurun_out/${TAG}_$label gpurun_out/${TAG}_$label/summary.json > /dev/null &&
  find gpurun_out/${TAG}_$label -name '*.csv' ! -name '*kernel_stats.csv' -delete
}
# (the ten take about 25 minutes of box time: "a" and "b" run half of them each, for calls with a time limit)
PART=${2:-all}
if [ "$PART" != "b" ]; then
one dragon1080p &&
one torus10m --scene torus10m --max-depth 8 --spp-per-step 128 &&
one ref4 --scene ref4 --spp-per-step 128 &&
one ref4_two_level --scene ref4 --spp-per-step 128 --two-level &&
one c5_4k --scene c5 --width 3840 --height 2160 --spp-per-step 64 || exit 1
fi
if [ "$PART" != "a" ]; then
one cornell1024 --scene cornell --width 1024 --height 1024 --max-depth 8 &&
one cornell512 --scene cornell --width 512 --height 512 --max-depth 8 --spp-per-step 16 &&
one dragon4k --width 3840 --height 2160 --spp-per-step 64 &&
one mega_dragon4k --width 3840 --height 2160 --spp-per-step 16 --integrator megakernel &&
one mega_c5_4k --scene c5 --width 3840 --height 2160 --spp-per-step 16 --integrator megakernel || exit 1
fi
