#!/usr/bin/env python3
"""Per-bounce survival of paths on the GPU (runs on the GPU box): fraction of paths that reach
their k-th Intersect call, measured with pt_debug_trace on a window of the frame.  In the classic
megakernel this is the active-lane fraction at bounce k; with regeneration or compaction the
dead lanes are refilled, which is what the variants' counters show as a whole.

    python tools/lane_occupancy.py [demo|mixed|synth10k] [depth] > profiles/<name>.md
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pt = importlib.import_module("cuda-pathtracer_amd")


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "mixed"
    depth = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    sid = {"demo": pt.SCENE_DEMO, "mixed": pt.SCENE_MIXED, "synth10k": pt.SCENE_SYNTH10K}[scene]
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(sid, 1)
    w, h = 1920, 1080
    cam = pt.Camera(eye, look, up, w, h, 0, 0, fov)
    tr = pt.Pathtracer(cam, depth, 1)
    tr.Init(mats, sph, lights)
    win = (0, 412, 1920, 668) if scene != "synth10k" else (704, 476, 1216, 604)
    alive = np.zeros(depth + 1, np.int64)
    n_paths = 0
    for it in (1, 2):
        hits, _, _, _ = tr.DebugTrace(*win, it)
        alive += (hits != pt.HIT_NOT_TRACED).reshape(-1, depth + 1).sum(axis=0)
        n_paths += hits.size // (depth + 1)
    tr.Release()
    print(f"# paths alive per bounce: {scene} scene, 1080p window {win}, depth {depth}, {n_paths} paths\n")
    print("| Intersect call k | paths that reach it | fraction (= active lanes of a classic megakernel wave at bounce k) |")
    print("|---|---|---|")
    for k, a in enumerate(alive):
        print(f"| {k} | {a} | {a / n_paths:.4f} |")
    print(f"\nmean Intersect calls per path: {alive.sum() / n_paths:.3f}; "
          f"mean active-lane fraction of a wave that waits for its longest path "
          f"(all {depth + 1} bounces): {alive.sum() / n_paths / (depth + 1):.3f}")


if __name__ == "__main__":
    main()
