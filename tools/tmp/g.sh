AB_VARIANTS="a f" bash tools/ab.sh --scene ref4 --spp-per-step 128 --regroup | head -4
AB_VARIANTS="a f" bash tools/ab.sh --regroup | head -4
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
run() { timeout -k 10 300 $B "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), d['roofline']['phase_ms_per_step'])" || exit 1; }
for wb in 1 2; do for rg in "" "--regroup"; do
run --scene ref4 --spp-per-step 128 --wavefront-bounces $wb $rg
run --scene c5 --width 3840 --height 2160 --spp-per-step 64 --wavefront-bounces $wb $rg
done; done
