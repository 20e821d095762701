#!/bin/bash
# Interleaved runs of tools/pt_bench against several builds of the library on one GPU box:
#   tools/abn.sh <rounds> "<dir> <dir> ..." <pt_bench arguments...>
# every <dir> holds a libptmi.so ("tree" = the working tree's cuda-pathtracer_amd/).
set -u
R=$1; DIRS=$2; shift 2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in $(seq 1 "$R"); do
    for d in $DIRS; do
        echo "$d:"
        if [ "$d" = tree ]; then timeout -k 10 120 tools/pt_bench "$@" | grep kernel || exit 1
        else LD_LIBRARY_PATH=$d timeout -k 10 120 tools/pt_bench "$@" | grep kernel || exit 1; fi
    done
done
