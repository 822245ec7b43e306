#!/bin/bash
# The default bench line and the side configurations, after tools/collect_r03.sh has
# filed the round's counter profiles (their figures then appear in the lines).
# Usage (on the GPU box): bash tools/bench_r03.sh [tag]
set -u
TAG=${1:-r03}
bash tools/side_configs.sh > gpurun_out/${TAG}_side_configs.jsonl 2> gpurun_out/${TAG}_side_configs.err &&
timeout -k 10 300 python3 bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
