#!/bin/bash
# Measurement build of the library with the queue-sorting experiment (pt_sort_experiment.hip) -> tools/ab_sort/libptmi.so
set -eu
cd "$(dirname "$0")/.."
mkdir -p tools/ab_sort
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
    -fhip-fp32-correctly-rounded-divide-sqrt -Iinclude -Icuda-pathtracer_amd/csrc -Wall -Wno-unused-function -DPT_SORT_EXPERIMENT \
    -shared -o tools/ab_sort/libptmi.so cuda-pathtracer_amd/csrc/pt_kernels.hip cuda-pathtracer_amd/csrc/pt_wavefront.hip \
    cuda-pathtracer_amd/csrc/pt_nee.hip cuda-pathtracer_amd/csrc/pt_api.cpp cuda-pathtracer_amd/csrc/pt_scene.cpp \
    cuda-pathtracer_amd/csrc/pt_bvh.cpp cuda-pathtracer_amd/csrc/pt_checkpoint.cpp tools/experiments/pt_sort_experiment.hip
