"""1080p demo render, 64 iterations per pt_run_n through the path pool: time against the number of
batches the render is cut into (PT_TUNE_WAVEFRONT_BATCH).  Runs on the GPU box."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pt = importlib.import_module("cuda-pathtracer_amd")
mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO, 1)
cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
buf = torch.zeros(1920 * 1080 * 3, device="cuda")
tr = pt.Pathtracer(cam, 8, 1); tr.Init(mats, sph, lights)
per_it = 1920 * 1080 * 4 / 2**20
for batch_it in (0, 32, 22, 16, 11, 8, 4, 2):
    tr.SetTuning(pt.TUNE_WAVEFRONT_BATCH, int(2 * batch_it * per_it + 1) if batch_it else 0)
    best = 1e9
    for rep in range(6):
        tr.Reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
        tr.RunN(buf, 64); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"iterations per batch {batch_it if batch_it else 'default'}: {best*1e3:.2f} ms", flush=True)
