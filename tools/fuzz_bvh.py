#!/usr/bin/env python3
"""Randomised search for a scene on which the sphere BVH and the linear scan disagree.

    python tools/fuzz_bvh.py [n_scenes] [first_seed]      (on the GPU box)

Every scene draws its own size, radius distribution, material mix (mirror- and glass-heavy mixes
stress the box widening for over-long directions), coordinate offset, camera and depth; the two
searches must agree bit for bit on hit-index sequences, per-path radiance and images, in the
megakernel and in the wavefront variant.  Exit status 1 and the scene's seed on a mismatch."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("cuda-pathtracer_amd")


def random_scene(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([200, 260, 400, 700, 1500, 4000, 12000]))
    centre = rng.choice([0.0, 0.0, 300.0, 5000.0]) * rng.normal(size=3)
    half = float(rng.choice([5.0, 60.0, 400.0]))
    mats = np.zeros(5, pt.MATERIAL_DTYPE)
    mats["color"] = rng.uniform(0.3, 0.98, (5, 3))
    mats["type"] = [0, 1, 2, 0, 1]
    mats["coef"] = [0.0, rng.uniform(0.5, 1.0), rng.uniform(1.1, 2.4), 0.0, 1.0]
    big = 1e5 * half / 60.0
    walls = np.zeros(7, pt.SPHERE_DTYPE)
    offs = [(-(big + half), 0, 0), (big + half, 0, 0), (0, -(big + half), 0), (0, big + half, 0),
            (0, 0, -(big + half)), (0, 0, big + half)]
    n_walls = int(rng.choice([6, 6, 5, 0]))
    for i, o in enumerate(offs):
        walls["pos"][i] = centre + np.array(o)
        walls["radius"][i] = big
    walls["material_id"][:6] = rng.integers(0, 5, 6)
    walls["light_id"] = -1
    walls = walls[:n_walls + 1]
    walls["pos"][-1] = centre + np.array([0, half * 0.7, 0])
    walls["radius"][-1] = half * rng.uniform(0.05, 0.3)
    walls["material_id"][-1] = 0
    walls["light_id"][-1] = 0
    lights = np.zeros(1, pt.LIGHT_DTYPE)
    lights["type"] = 1
    lights["u_bits"][0, 0] = len(walls) - 1
    lights["intensity"] = [[12, 12, 12]]
    balls = np.zeros(n, pt.SPHERE_DTYPE)
    kind = rng.integers(0, 5)
    if kind == 0:      # uniform cloud
        balls["pos"] = centre + rng.uniform(-0.85 * half, 0.85 * half, (n, 3))
    elif kind == 1:    # clumps
        seeds = rng.uniform(-0.7 * half, 0.7 * half, (int(rng.integers(2, 20)), 3))
        balls["pos"] = centre + seeds[rng.integers(0, len(seeds), n)] + rng.normal(0, 0.04 * half, (n, 3))
    elif kind == 2:    # a sheet
        balls["pos"] = centre + rng.uniform(-0.85 * half, 0.85 * half, (n, 3)) * np.array([1, 0.02, 1])
    elif kind == 4:    # a flat, elongated cloud: the boxes' slack exceeds the short axes' extent
        balls["pos"] = centre + rng.uniform(-0.85 * half, 0.85 * half, (n, 3)) * np.array([25, 1e-6, 3e-3])
    else:              # a regular grid (many exact ties in the split planes)
        k = int(np.ceil(n ** (1 / 3)))
        g = np.stack(np.meshgrid(*[np.arange(k)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n]
        balls["pos"] = centre + (g / max(k - 1, 1) - 0.5) * 1.6 * half
    spread = float(rng.choice([1.0, 3.0, 30.0]))
    base = half / 60.0 * rng.choice([0.02, 0.3, 1.0])
    balls["radius"] = base * np.exp(rng.uniform(0, np.log(spread), n))
    mix = rng.choice(3)
    p = [[0.7, 0.1, 0.1, 0.05, 0.05], [0.1, 0.4, 0.1, 0.0, 0.4], [0.05, 0.05, 0.9, 0.0, 0.0]][mix]
    balls["material_id"] = rng.choice(5, n, p=p)
    balls["light_id"] = -1
    sph = np.concatenate([walls, balls])
    eye = centre + half * np.array([0.0, 0.1, 0.9]) * rng.choice([1.0, 1.0, 25.0])
    cam = pt.Camera(eye.tolist(), centre.tolist(), [0, 1, 0], 96, 64, 0, 0, float(rng.choice([8.0, 60.0])))
    return mats, sph, lights, cam, int(rng.choice([4, 8, 16]))


def render(tr, buf, accel, variant):
    tr.SetAccel(accel)
    tr.SetKernelVariant(variant)
    tr.Reset()
    tr.ResetCounters()
    tr.RunN(buf, 2)
    return buf.cpu().numpy().tobytes(), tr.RayCount()


def main():
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    buf = torch.zeros(96 * 64 * 3, device="cuda")
    n_bvh = 0
    for seed in range(first, first + n_scenes):
        mats, sph, lights, cam, depth = random_scene(seed)
        tr = pt.Pathtracer(cam, depth, seed)
        tr.Init(mats, sph, lights)
        ref = render(tr, buf, pt.ACCEL_BRUTE_FORCE, 0)
        ref_trace = tr.DebugTrace(0, 0, 96, 64, 1)
        if tr.AccelNodes() > 0:
            n_bvh += 1
            # the BVH megakernel, the persistent waves (default launch structure), static chunks,
            # PT_VARIANT_AUTO, and the other two mappings of bounces to launches
            for variant, mode in ((0, -1), (100, -1), (101, -1), (-1, -1), (-1, 0), (-1, 1)):
                tr.SetTuning(pt.TUNE_WAVEFRONT_WHOLE_PATHS, mode)
                got = render(tr, buf, pt.ACCEL_AUTO, variant)
                if got != ref:
                    print(f"MISMATCH seed {seed} variant {variant} launches {mode}: {len(sph)} spheres, rays {got[1]} vs {ref[1]}")
                    return 1
            tr.SetTuning(pt.TUNE_WAVEFRONT_WHOLE_PATHS, -1)
            tr.SetKernelVariant(0)
            hits, rad, _, _ = tr.DebugTrace(0, 0, 96, 64, 1)
            if not np.array_equal(hits, ref_trace[0]) or rad.tobytes() != ref_trace[1].tobytes():
                print(f"MISMATCH (trace) seed {seed}")
                return 1
        tr.Release()
        if (seed - first) % 20 == 19:
            print(f"{seed - first + 1} scenes ok ({n_bvh} through a BVH)", flush=True)
    print(f"all {n_scenes} scenes agree ({n_bvh} through a BVH)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
