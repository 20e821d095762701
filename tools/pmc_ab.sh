#!/bin/bash
# A/B of instruction counters on the GPU box: one rocprofv3 --pmc pass of tools/pt_bench with the
# baseline build (tools/ab/libptmi.so) and one with the tree's build; sums per kernel substring.
#   tools/pmc_ab.sh <kernel substring> <pt_bench arguments...>
set -u
SUB=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
rm -rf "$REPO/gpurun_out/pmc_ab_a" "$REPO/gpurun_out/pmc_ab_b"
export LD_LIBRARY_PATH=$REPO/tools/ab
timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d "$REPO/gpurun_out/pmc_ab_a" -o pmc -- "$REPO/tools/pt_bench" "$@" > "$REPO/gpurun_out/pmc_ab_a.log" 2>&1 || { echo "pass A failed"; exit 1; }
unset LD_LIBRARY_PATH
timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d "$REPO/gpurun_out/pmc_ab_b" -o pmc -- "$REPO/tools/pt_bench" "$@" > "$REPO/gpurun_out/pmc_ab_b.log" 2>&1 || { echo "pass B failed"; exit 1; }
echo "A (baseline):"; python3 "$REPO/tools/pmc_sum.py" "$REPO/gpurun_out/pmc_ab_a" "$SUB"
echo "B (tree):";     python3 "$REPO/tools/pmc_sum.py" "$REPO/gpurun_out/pmc_ab_b" "$SUB"
