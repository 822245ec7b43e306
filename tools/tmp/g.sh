set -o pipefail
AB_VARIANTS="m t4 t16" bash tools/ab.sh | head -6
