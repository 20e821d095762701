#!/usr/bin/env python3
"""Registers, scratch, static LDS and the occupancy they allow for every kernel of the library,
from the compiler's own resource remarks (the flags of `make lib`).  No GPU needed.

    python tools/kernel_resources.py > profiles/<round>_kernel_resources.md
"""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize "
         "-fhip-fp32-correctly-rounded-divide-sqrt").split() + [f"-I{ROOT}/include", f"-I{ROOT}/cuda-pathtracer_amd/csrc"]


def demangle(name):
    out = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    out = out.replace("ptmi::(anonymous namespace)::", "").replace("ptmi::", "")
    return re.sub(r"^void ", "", re.sub(r"\((Wavefront|Render)Params\)", "", out))


def main():
    print("# kernel resources (hipcc -Rpass-analysis=kernel-resource-usage, gfx950)\n")
    print("Dynamic LDS is set by the launch and not part of these figures: the path pool takes the scene "
          "(32 B per sphere + materials + lights) + 4 rings of 192 camera rays of 20 B (15.8 KB on the 9-sphere scene), "
          "the BVH kernels 2 KiB per traversal-stack entry (24 KiB at the default 12 entries), the megakernels the scene "
          "or a 16 KiB sphere tile (+ 24 KiB camera-ray ring on scenes of up to 24 spheres).\n")
    print("| kernel | VGPRs | AGPRs | SGPRs | scratch bytes per lane | static LDS bytes per workgroup | waves per SIMD (compiler's estimate) |")
    print("|" + "---|" * 7)
    for f in ("pt_wavefront", "pt_kernels", "pt_nee"):
        err = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c",
                              f"{ROOT}/cuda-pathtracer_amd/csrc/{f}.hip", "-o", "/dev/null"], capture_output=True, text=True).stderr
        cur = {}
        for line in err.splitlines():
            m = re.search(r"remark: +([A-Za-z \[\]/]+): (\S+)", line)
            if not m:
                continue
            key, val = m.group(1).strip(), m.group(2)
            if key == "Function Name":
                cur = {"name": val}
            cur[key] = val
            if key.startswith("LDS Size") and "name" in cur:
                print(f"| `{demangle(cur['name'])}` | {cur.get('VGPRs')} | {cur.get('AGPRs')} | {cur.get('TotalSGPRs')} | "
                      f"{cur.get('ScratchSize [bytes/lane]')} | {val} | {cur.get('Occupancy [waves/SIMD]')} |")


if __name__ == "__main__":
    main()
