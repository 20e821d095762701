#!/usr/bin/env python3
"""Render time of the 10 007-sphere scene (1080p, 16 iterations, default variant) against the BVH
visit budget after which a ray is finished by the wave-wide scan: python tools/bvh_budget.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("cuda-pathtracer_amd")
mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_SYNTH10K, 1)
cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
buf = torch.zeros(1920 * 1080 * 3, device="cuda")
tr = pt.Pathtracer(cam, 8, 1)
tr.Init(mats, sph, lights)
for budget in (0, 48, 64, 90, 125, 180, 250, 400):
    tr.SetAccelBudget(budget)
    ts = []
    for r in range(3):
        tr.Reset(); tr.RunN(buf, 16); ts.append(tr.LastKernelMs())
    print(f"budget {budget if budget else 'default'}: {min(ts):.2f} ms", flush=True)
