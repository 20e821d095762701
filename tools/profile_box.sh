#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): kernel-trace statistics and PMC passes of one bench.py
# command.  Each counter group gets a run of its own; results land under gpurun_out/prof/<tag>/.
#   tools/profile_box.sh <tag> <bench.py arguments...>
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --no-cpu-baseline --no-other-workloads $*"

timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- $BENCH > "$OUT/kt.log" 2>&1 || { echo "kernel-trace failed"; tail -5 "$OUT/kt.log"; exit 1; }
# The default render overlaps the launches of consecutive batches on two streams, so the durations of
# that trace are not additive.  A second trace with one buffer set on one stream (every launch after
# the one before) gives per-kernel durations that sum to the step.
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_serial" -o kt -- $BENCH --serial-launches > "$OUT/kt_serial.log" 2>&1 || { echo "serial kernel-trace failed"; tail -5 "$OUT/kt_serial.log"; exit 1; }

pass() {
    local name=$1; shift
    timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- $BENCH > "$OUT/$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS &&
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 &&
pass sq3 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_LEVEL_WAVES GRBM_GUI_ACTIVE GRBM_COUNT &&
pass tcc_rd FETCH_SIZE &&
pass tcc_wr WRITE_SIZE &&
find "$OUT" -name '*.csv' | sort
