#!/usr/bin/env python3
"""Randomised comparison of the shipped kernels with the CPU oracle (the restatement of the
reference's trace loop): random small frames, depths, iteration counts and scenes (built-in ones
and random sphere clouds of all three materials, through the linear scan or the BVH; random triangle soups
with and without spheres, resident or streamed through LDS tiles), both integrators,
one call or a series of asynchronous calls; frame
buffers and Intersect-call counts must agree bit for bit.

    python tools/fuzz_oracle.py [n_cases] [first_seed]      (on the GPU box)"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("cuda-pathtracer_amd")
import oracle_py as O  # noqa: E402
from fuzz_steal import random_cloud  # noqa: E402
from triangle_scenes import triangle_soup  # noqa: E402


def one_case(seed):
    rng = np.random.default_rng(seed)
    kind = int(rng.integers(0, 5))
    tris = None
    eye, look, up, fov = [0, 45, 79.5], [0, 35, 0], [0, 1, 0], 55.0
    if kind == 0:
        mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO, 1)
    elif kind == 1:
        mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_MIXED, 1)
    elif kind == 4:
        mats, sph, tris, lights, (eye, look, up, fov) = triangle_soup(
            pt, rng, int(rng.choice([1, 9, 40, 256, 257, 700])), int(rng.choice([0, 0, 5, 30, 300])))
    else:
        mats, sph, lights = random_cloud(rng, int(rng.choice([3, 30, 150, 260, 600])))
        mats["coef"][2] = rng.choice([1.1, 1.5, 2.4])
    w, h = int(rng.integers(1, 56)), int(rng.integers(1, 40))
    depth = int(rng.choice([0, 1, 4, 8, 16]))
    n = int(rng.choice([1, 2, 3, 6]))
    variant = int(rng.choice([pt.VARIANT_AUTO, pt.VARIANT_AUTO, pt.VARIANT_STEALING, pt.VARIANT_WAVEFRONT]))
    accel = pt.ACCEL_BRUTE_FORCE if rng.random() < 0.3 else pt.ACCEL_AUTO
    # the direct-light integrator (the wavefront variant's per-bounce queues do not carry it), and the
    # render as a series of pt_run_async calls
    nee = variant != pt.VARIANT_WAVEFRONT and rng.random() < 0.4
    if tris is not None:        # scenes with triangles: the path pool only
        variant, accel, nee = pt.VARIANT_AUTO, pt.ACCEL_AUTO, False
    use_async = rng.random() < 0.3
    cam = pt.Camera(eye, look, up, w, h, 0, 0, fov)
    ref, _, _, ref_rays = O.oracle_render(cam.pod, mats, sph, lights, seed & 0xffff, depth, n_iterations=n, n_threads=16,
                                          integrator=O.INTEGRATOR_NEE_MIS if nee else O.INTEGRATOR_PATH, triangles=tris)
    tr = pt.Pathtracer(cam, depth, seed & 0xffff)
    tr.Init(mats, sph, lights, tris)
    tr.SetAccel(accel)
    tr.SetKernelVariant(variant)
    tr.SetIntegrator(pt.INTEGRATOR_NEE_MIS if nee else pt.INTEGRATOR_PATH)
    tr.ResetCounters()
    buf = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
    if use_async:
        for _ in range(n):
            tr.RunAsync(buf)
        tr.Wait()
    else:
        tr.RunN(buf, n)
    torch.cuda.synchronize()
    ok = buf.cpu().numpy().tobytes() == np.ascontiguousarray(ref, np.float32).tobytes() and tr.RayCount() == ref_rays
    tr.Release()
    return ok, (kind, len(sph), w, h, depth, n, variant, accel, nee, use_async)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    for k in range(n_cases):
        ok, what = one_case(first + k)
        if not ok:
            print(f"MISMATCH seed {first + k}: {what}", flush=True)
            sys.exit(1)
        if (k + 1) % 50 == 0:
            print(f"{k + 1} cases ok", flush=True)
    print(f"all {n_cases} cases agree with the oracle")


if __name__ == "__main__":
    main()
