#!/bin/bash
# Builds the library into tools/<dir>/libptmi.so with extra compiler flags, for tools/abn.sh:
#   tools/build_variant.sh <dir> [flags...]
set -eu
D=$1; shift
cd "$(dirname "$0")/.."
mkdir -p "tools/$D"
C=cuda-pathtracer_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
    -fhip-fp32-correctly-rounded-divide-sqrt -Iinclude -I$C -Wall -Wno-unused-function "$@" -shared -o "tools/$D/libptmi.so" \
    $C/pt_kernels.hip $C/pt_wavefront.hip $C/pt_nee.hip $C/pt_api.cpp $C/pt_scene.cpp $C/pt_bvh.cpp $C/pt_checkpoint.cpp
