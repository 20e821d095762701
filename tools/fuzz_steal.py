#!/usr/bin/env python3
"""Randomised comparison of the work-stealing megakernel (with and without iteration ranges, both
integrators) and of PT_VARIANT_AUTO (the path pool and the BVH wavefront launches where they apply,
both integrators, with random limits on the paths in flight, in one call and as a series of
pt_run_async calls) with the plain regeneration megakernel: random frame sizes, iteration counts, depths,
scenes (built-in ones and random sphere clouds scanned brute force or through the BVH), continued
renders, sum buffers and row bands.  Frame buffers and Intersect-call counts must agree bit for bit.

    python tools/fuzz_steal.py [n_cases] [first_seed]      (on the GPU box)"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("cuda-pathtracer_amd")


def random_cloud(rng, n):
    mats = np.zeros(5, pt.MATERIAL_DTYPE)
    mats["color"] = rng.uniform(0.2, 0.95, (5, 3))
    mats["type"] = [0, 1, 2, 0, 0]
    mats["coef"] = [0.0, 0.9, 1.4, 0.0, 0.0]
    sph = np.zeros(n, pt.SPHERE_DTYPE)
    sph["pos"] = rng.uniform(-30, 30, (n, 3)) + np.array([0, 35, -20])
    sph["radius"] = rng.uniform(1.0, 7.0, n) * (40.0 / n) ** (1 / 3)
    sph["material_id"] = rng.integers(0, 5, n)
    sph["light_id"] = -1
    sph["light_id"][0] = 0
    lights = np.zeros(1, pt.LIGHT_DTYPE)
    lights["type"] = 1
    lights["u_bits"][0, 0] = 0
    lights["intensity"] = [[9, 8, 7]]
    return mats, sph, lights


def one_case(seed):
    rng = np.random.default_rng(seed)
    kind = int(rng.integers(0, 4))
    eye, look, up, fov = [0, 45, 79.5], [0, 35, 0], [0, 1, 0], 55.0
    accel = pt.ACCEL_AUTO
    if kind == 0:
        mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO, 1)
    elif kind == 1:
        mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_MIXED, 1)
    else:
        mats, sph, lights = random_cloud(rng, int(rng.choice([3, 30, 150, 400, 700, 1500])))
        if kind == 3:
            accel = pt.ACCEL_BRUTE_FORCE
    w, h = int(rng.integers(1, 140)), int(rng.integers(1, 100))
    depth = int(rng.choice([0, 2, 8, 16]))
    n = int(rng.choice([1, 2, 3, 4, 5, 16, 33, 70]))
    first = int(rng.choice([1, 1, 3]))
    mode = int(rng.choice([pt.ACCUM_RUNNING_MEAN, pt.ACCUM_SUM]))
    bands = (16, int(rng.integers(0, 3)), 3) if rng.random() < 0.3 else None
    integrator = pt.INTEGRATOR_NEE_MIS if rng.random() < 0.45 else pt.INTEGRATOR_PATH
    cam = pt.Camera(eye, look, up, w, h, 0, 0, fov)
    out = {}
    # + 2 / + 3: single paths / whole pixel-iterations change hands; + 10 + k: k + 1 iteration ranges
    pool_limit = int(rng.choice([0, 0, 12 * w * h * 4 * int(rng.integers(1, 4)) + 2**20]))
    # "async": PT_VARIANT_AUTO rendered as a series of pt_run_async calls of random sizes + one pt_wait
    chunks = []
    while sum(chunks) < n:
        chunks.append(min(n - sum(chunks), int(rng.choice([1, 1, 2, 3, 7, 40]))))
    for variant in (pt.VARIANT_REGENERATION, pt.VARIANT_STEALING, pt.VARIANT_STEALING + 2, pt.VARIANT_STEALING + 3,
                    pt.VARIANT_STEALING + 10 + int(rng.integers(1, 6)), pt.VARIANT_AUTO, "async"):
        tr = pt.Pathtracer(cam, depth, seed)
        tr.Init(mats, sph, lights)
        tr.SetAccel(accel)
        tr.SetIntegrator(integrator)
        tr.SetKernelVariant(pt.VARIANT_AUTO if variant == "async" else variant)
        if variant in (pt.VARIANT_AUTO, "async"):
            tr.SetMemoryLimit(pool_limit)
        tr.SetAccumMode(mode)
        if bands:
            tr.SetRowBands(*bands)
        buf = torch.full((w * h * 3,), 0.5, dtype=torch.float32, device="cuda")
        tr.Reset()
        if first > 1:
            tr.RunN(buf, first - 1)
        tr.ResetCounters()
        if variant == "async":
            for k in chunks:
                tr.RunAsync(buf, k)
            tr.Wait()
        else:
            tr.RunN(buf, n)
        torch.cuda.synchronize()
        out[variant] = (buf.cpu().numpy().tobytes(), tr.RayCount())
        tr.Release()
    ref = out[pt.VARIANT_REGENERATION]
    return all(v == ref for v in out.values()), (kind, len(sph), w, h, depth, n, first, mode, bands, integrator)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first_seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    for k in range(n_cases):
        ok, what = one_case(first_seed + k)
        if not ok:
            print(f"MISMATCH at seed {first_seed + k}: {what}", flush=True)
            sys.exit(1)
        if (k + 1) % 50 == 0:
            print(f"{k + 1} cases ok", flush=True)
    print(f"all {n_cases} cases agree")


if __name__ == "__main__":
    main()
