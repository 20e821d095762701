#!/bin/bash
# A/B on the GPU box: tools/ab/libptmi.so (baseline build) against cuda-pathtracer_amd/libptmi.so
# (working tree), interleaved runs of tools/pt_bench in one call so both see the same box.
#   tools/ab.sh <rounds> <pt_bench arguments...>
set -u
R=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in $(seq 1 "$R"); do
    echo "A (baseline):"; LD_LIBRARY_PATH=tools/ab timeout -k 10 120 tools/pt_bench "$@" | grep kernel || exit 1
    echo "B (tree):";     timeout -k 10 120 tools/pt_bench "$@" | grep kernel || exit 1
done
