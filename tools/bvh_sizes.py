#!/usr/bin/env python3
"""Render time of clouds of n small spheres in the demo room through the sphere BVH (default
variant), optionally with another build of the library: python tools/bvh_sizes.py [libptmi.so]"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("cuda-pathtracer_amd")
if len(sys.argv) > 1:
    pt.LIB_PATH = os.path.abspath(sys.argv[1])
mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO)
cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
buf = torch.zeros(1920 * 1080 * 3, device="cuda")
rng = np.random.default_rng(1)
for n in (1000, 3000, 10000, 30000, 100000, 300000):
    extra = np.zeros(n, pt.SPHERE_DTYPE)
    extra["pos"] = np.column_stack([rng.uniform(-45, 45, n), rng.uniform(5, 75, n), rng.uniform(-75, 20, n)])
    extra["radius"] = rng.uniform(0.3, 1.2, n) * (10000 / n) ** (1 / 3)
    extra["material_id"] = rng.integers(1, 6, n); extra["light_id"] = -1
    tr = pt.Pathtracer(cam, 8, 1); tr.Init(mats, np.concatenate([sph, extra]), lights)
    ts = []
    for r in range(3):
        tr.Reset(); tr.RunN(buf, 4); ts.append(tr.LastKernelMs())
    print(f"{n + len(sph)} spheres, {tr.AccelNodes()} nodes: {min(ts):.2f} ms per 4 iterations", flush=True)
    tr.Release()
