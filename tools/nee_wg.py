import importlib, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pt = importlib.import_module("cuda-pathtracer_amd")
mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO, 1)
cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
buf = torch.zeros(1920 * 1080 * 3, device="cuda")
tr = pt.Pathtracer(cam, 8, 1); tr.Init(mats, sph, lights)
tr.SetIntegrator(pt.INTEGRATOR_NEE_MIS)
for wg in (0, 2, 3, 4, 5):
    tr.SetTuning(pt.TUNE_POOL_WORKGROUPS, wg)
    best = 1e9
    for rep in range(3):
        tr.Reset(); tr.ResetCounters()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tr.RunN(buf, 64)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"NEE demo1080 x 64, pool workgroups per CU {wg}: {best*1e3:.2f} ms, {tr.RayCount()/best/1e9:.1f} G bounces/s", flush=True)
