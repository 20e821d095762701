"""64 x pt_run of the demo scene at 1080p (runs on the GPU box, also under rocprofv3 --kernel-trace)."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pt = importlib.import_module("cuda-pathtracer_amd")
mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO, 1)
cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
buf = torch.zeros(1920 * 1080 * 3, device="cuda")
tr = pt.Pathtracer(cam, 8, 1); tr.Init(mats, sph, lights)
for rep in range(3):
    tr.Reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(64):
        tr.Run(buf)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"64 x pt_run: {dt*1e3:.2f} ms", flush=True)
