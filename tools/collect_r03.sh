#!/bin/bash
# Copies what tools/profile_r03.sh left under gpurun_out/ into profiles/ (run here, after the GPU call);
# with a second argument "bench", also the lines tools/bench_r03.sh wrote.
set -eu
TAG=${1:-r03}
U="python3 tools/update_profiles.py"
$U gpurun_out/${TAG}_dragon1080p $TAG wavefront_dragon1080p > /dev/null
$U gpurun_out/${TAG}_torus10m $TAG wavefront_torus10m --scene torus10m --spp 128 --max-depth 8 > /dev/null
$U gpurun_out/${TAG}_ref4 $TAG wavefront_ref4 --scene ref4 --spp 128 > /dev/null
$U gpurun_out/${TAG}_ref4_two_level $TAG wavefront_ref4_two_level --scene ref4 --spp 128 --two-level > /dev/null
$U gpurun_out/${TAG}_c5_4k $TAG wavefront_c5_4k --scene c5 --width 3840 --height 2160 --spp 64 > /dev/null
$U gpurun_out/${TAG}_cornell1024 $TAG wavefront_cornell1024 --scene cornell --width 1024 --height 1024 --spp 256 --max-depth 8 > /dev/null
$U gpurun_out/${TAG}_cornell512 $TAG wavefront_cornell512 --scene cornell --width 512 --height 512 --spp 16 --max-depth 8 > /dev/null
$U gpurun_out/${TAG}_dragon4k $TAG wavefront_dragon4k --width 3840 --height 2160 --spp 64 > /dev/null
$U gpurun_out/${TAG}_mega_dragon4k $TAG megakernel_dragon4k --width 3840 --height 2160 --spp 16 > /dev/null
$U gpurun_out/${TAG}_mega_c5_4k $TAG megakernel_c5_4k --scene c5 --width 3840 --height 2160 --spp 16 > /dev/null
if [ "${2:-}" = "bench" ]; then cp gpurun_out/${TAG}_side_configs.jsonl gpurun_out/${TAG}_bench_default.json profiles/; fi
echo "profiles/${TAG}_* updated"
