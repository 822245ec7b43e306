#!/bin/bash
# The configurations beside the headline one: 10 M triangles at depth 8 (working
# set beyond the Infinity Cache), 4K with the two integrators (the dragon
# stand-in alone, and inside the Cornell box under a sky read from a .hdr file),
# the Cornell box, the reference's four-instance table flattened and two-level.
# Usage (on the GPU box): bash tools/side_configs.sh > gpurun_out/side_configs.jsonl
set -u
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 10 400 $B --scene torus10m --max-depth 8 --spp-per-step 128 &&
timeout -k 10 300 $B --width 3840 --height 2160 --spp-per-step 64 &&
timeout -k 10 300 $B --width 3840 --height 2160 --spp-per-step 16 --integrator megakernel &&
timeout -k 10 300 $B --scene c5 --width 3840 --height 2160 --spp-per-step 64 &&
timeout -k 10 300 $B --scene c5 --width 3840 --height 2160 --spp-per-step 16 --integrator megakernel &&
timeout -k 10 200 $B --scene cornell --width 1024 --height 1024 --max-depth 8 --spp-per-step 256 &&
timeout -k 10 200 $B --scene cornell --width 512 --height 512 --max-depth 8 --spp-per-step 16 &&
timeout -k 10 300 $B --scene ref4 --spp-per-step 128 &&
timeout -k 10 300 $B --scene ref4 --spp-per-step 128 --two-level
