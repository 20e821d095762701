#!/bin/bash
# Vector-memory pipeline counters of one tools/pt_bench run on the GPU box (texture addresser,
# L1 cache, texture data), two passes; sums per kernel substring.
#   tools/pmc_tcp.sh <kernel substring> <pt_bench arguments...>
set -u
SUB=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() {
    local name=$1; shift
    rm -rf "$REPO/gpurun_out/$name"
    timeout -s KILL 120 rocprofv3 --pmc $PMC --output-format csv -d "$REPO/gpurun_out/$name" -o pmc -- "$REPO/tools/pt_bench" "$@" > "$REPO/gpurun_out/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$REPO/gpurun_out/$name.log"; exit 1; }
    python3 "$REPO/tools/pmc_sum.py" "$REPO/gpurun_out/$name" "$SUB"
}
PMC="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN2_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE"
run pmc_tcp1 "$@" &&
PMC="TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TD_LOAD_WAVEFRONT_sum TD_COALESCABLE_WAVEFRONT_sum GRBM_GUI_ACTIVE" &&
run pmc_tcp2 "$@" &&
PMC="TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TAGRAM0_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" &&
run pmc_tcp3 "$@"
