#!/bin/bash
# Runs ON THE GPU BOX: the round's measurements in one call -- profiles (kernel traces + counter passes) of
# the three default launch structures, bench lines of every workload, the multi-rank rehearsals that one
# GPU allows, and the launch-granularity table.  Results under gpurun_out/.
#   tools/r3_measure.sh [all|profiles|bench]
# The bench lines quote the counters of profiles/pmc_traffic_*.json only when those were taken from the very
# build of the library that runs (its build id): after a rebuild run "profiles", condense the result in
# the development container (tools/r3_collect.sh), then run "bench".
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out
WHAT=${1:-all}
if [ "$WHAT" != bench ]; then
bash tools/profile_box.sh r3_demo1080_pool --steps 5 --warmup 1 > $O/r3_prof_demo.log 2>&1 || { tail -3 $O/r3_prof_demo.log; exit 1; }
bash tools/profile_box.sh r3_synth10k_bvh --workload synth10k --steps 3 --warmup 1 > $O/r3_prof_synth.log 2>&1 || { tail -3 $O/r3_prof_synth.log; exit 1; }
bash tools/profile_box.sh r3_mixed16_pool --workload mixed16 --steps 3 --warmup 1 > $O/r3_prof_mixed.log 2>&1 || { tail -3 $O/r3_prof_mixed.log; exit 1; }
fi
[ "$WHAT" = profiles ] && { echo "r3 profiles done"; exit 0; }
B="timeout -k 10 300 python bench.py --no-other-workloads"
$B > $O/r3_bench_demo1080.json 2> $O/r3_bench.err || exit 1
$B --workload mixed16 > $O/r3_bench_mixed16.json 2>> $O/r3_bench.err || exit 1
$B --workload synth10k > $O/r3_bench_synth10k.json 2>> $O/r3_bench.err || exit 1
$B --workload synth10k --accel brute --steps 3 > $O/r3_bench_synth10k_brute.json 2>> $O/r3_bench.err || exit 1
$B --workload demo4k --no-cpu-baseline > $O/r3_bench_demo4k.json 2>> $O/r3_bench.err || exit 1
$B --workload triroom1080 > $O/r3_bench_triroom1080.json 2>> $O/r3_bench.err || exit 1
$B --variant 500 --no-cpu-baseline > $O/r3_bench_demo1080_stealing.json 2>> $O/r3_bench.err || exit 1
$B --integrator nee --no-cpu-baseline > $O/r3_bench_demo1080_nee.json 2>> $O/r3_bench.err || exit 1
$B --integrator nee --workload mixed16 --no-cpu-baseline > $O/r3_bench_mixed16_nee.json 2>> $O/r3_bench.err || exit 1
$B --integrator nee --workload synth10k --no-cpu-baseline > $O/r3_bench_synth10k_nee.json 2>> $O/r3_bench.err || exit 1
# N > 1 rehearsals on one GPU: gloo ranks sharing the device (the reduce staged through host memory), and the
# RCCL calls with a communicator of one
PT_BENCH_BACKEND=gloo $B --gpus 2 --no-cpu-baseline --steps 2 > $O/r3_bench_2rank_gloo_rehearsal_sample.json 2>> $O/r3_bench.err || exit 1
PT_BENCH_BACKEND=gloo $B --gpus 2 --split tile --no-cpu-baseline --steps 2 > $O/r3_bench_2rank_gloo_rehearsal_tile.json 2>> $O/r3_bench.err || exit 1
PT_BENCH_SINGLE_RANK_PG=1 $B --no-cpu-baseline --steps 3 > $O/r3_bench_single_rank_rccl_rehearsal.json 2>> $O/r3_bench.err || exit 1
timeout -k 10 200 python tools/run_granularity.py > $O/r3_launch_granularity.txt 2>> $O/r3_bench.err || exit 1
echo "r3 measurements done"
