#!/bin/bash
# Runs HERE after tools/r3_measure.sh: condenses gpurun_out/ into profiles/ (tracked).
set -eu
cd "$(dirname "$0")/.."
python tools/kernel_resources.py > profiles/r3_kernel_resources.md
python tools/summarize_prof.py r3_demo1080_pool profiles/pmc_traffic_demo1080_pool.json > profiles/r3_demo1080_pool.md
python tools/summarize_prof.py r3_synth10k_bvh profiles/pmc_traffic_synth10k_bvh_v100whole.json > profiles/r3_synth10k_bvh_wavefront.md
python tools/summarize_prof.py r3_mixed16_pool profiles/pmc_traffic_mixed16_pool.json > profiles/r3_mixed16_pool.md
for f in gpurun_out/r3_bench_*.json; do cp "$f" profiles/; done
cp gpurun_out/r3_launch_granularity.txt profiles/
