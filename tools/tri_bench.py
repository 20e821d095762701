#!/usr/bin/env python3
"""Throughput of scenes with triangles (this build's extension) on the path pool: the box room of the
tests (14 triangles + 3 spheres, resident in LDS) and triangle soups that stream through LDS tiles.

    python tools/tri_bench.py [width height iterations]      (on the GPU box)"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("cuda-pathtracer_amd")
from triangle_scenes import box_room, triangle_soup  # noqa: E402


def measure(name, scene, w, h, n, depth=8):
    mats, sph, tris, lights, (eye, look, up, fov) = scene
    tr = pt.Pathtracer(pt.Camera(eye, look, up, w, h, 0, 0, fov), depth, 1)
    tr.Init(mats, sph, lights, tris)
    buf = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
    tr.RunN(buf, n)
    torch.cuda.synchronize()
    tr.Reset(); tr.ResetCounters()
    t0 = time.perf_counter()
    tr.RunN(buf, n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rays = tr.RayCount()
    print(f"{name:28s} {len(sph):4d} spheres {len(tris):5d} triangles  {w}x{h}x{n}: {dt * 1e3:8.2f} ms  "
          f"{rays / dt / 1e9:7.2f} G bounces/s  ({rays / (w * h * n):.2f} bounces/path)", flush=True)
    tr.Release()


def main():
    w, h, n = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (1920, 1080, 64)
    measure("box room", box_room(pt), w, h, n)
    rng = np.random.default_rng(3)
    measure("soup, resident", triangle_soup(pt, rng, 200, 20), w, h, max(n // 8, 1))
    measure("soup, streamed tiles", triangle_soup(pt, rng, 2000, 20), w // 2, h // 2, max(n // 16, 1))


if __name__ == "__main__":
    main()
