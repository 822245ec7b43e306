#!/bin/bash
# Collects rocprofv3 evidence for bench.py on the GPU box: one kernel-trace run
# and one run per counter group (each group fits the per-block slot limits).
# Usage: tools/profile_pmc.sh <out-dir-under-gpurun_out> [bench args...]
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
ARGS="--steps 2 --warmup 1 --no-cpu-baseline $*"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { # name, rocprof options...
  local name=$1; shift
  timeout -s KILL 180 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 "$GRAFT_REPO_ROOT/bench.py" $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
run trace --kernel-trace --stats
run sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
run sq2 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE GRBM_COUNT
run lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run tcc1 --pmc FETCH_SIZE TCC_HIT_sum
run tcc2 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
run tcp1 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum
echo "profile passes done: $(ls $OUT)"
