import importlib, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pt = importlib.import_module("cuda-pathtracer_amd")
mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO, 1)
cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
buf = torch.zeros(1920*1080*3, device="cuda")
tr = pt.Pathtracer(cam, 8, 1); tr.Init(mats, sph, lights)
for n_per in (1, 2, 4, 8, 64):
    tr.Reset(); tr.ResetCounters()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(64 // n_per):
        if n_per == 1: tr.Run(buf)
        else: tr.RunN(buf, n_per)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{64//n_per} launches x {n_per} iterations: {dt*1e3:.1f} ms, {tr.RayCount()/dt/1e9:.1f} G bounces/s", flush=True)
