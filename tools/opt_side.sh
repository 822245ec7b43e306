set -u
echo "== torus10m"; bash tools/opt_probe.sh "-1 8 16" --scene torus10m --max-depth 8 --spp-per-step 128 || exit 1
echo "== ref4"; bash tools/opt_probe.sh "-1 8 16" --scene ref4 --spp-per-step 128 || exit 1
echo "== ref4 two-level"; bash tools/opt_probe.sh "-1 8 16" --scene ref4 --spp-per-step 128 --two-level || exit 1
echo "== c5"; bash tools/opt_probe.sh "-1 8 16" --scene c5 --width 3840 --height 2160 --spp-per-step 64 || exit 1
echo "== cornell"; bash tools/opt_probe.sh "-1 8" --scene cornell --width 1024 --height 1024 --max-depth 8 --spp-per-step 256 || exit 1
