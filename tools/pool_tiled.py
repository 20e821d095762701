"""The 10 007-sphere scene rendered by the linear scan (no BVH) at 1080p: path pool against the
work-stealing megakernel, images compared (runs on the GPU box)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pt = importlib.import_module("cuda-pathtracer_amd")
mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_SYNTH10K, 1)
cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
imgs = {}
for name, variant in (("path pool", pt.VARIANT_AUTO), ("stealing", pt.VARIANT_STEALING)):
    tr = pt.Pathtracer(cam, 8, 1); tr.Init(mats, sph, lights)
    tr.SetAccel(pt.ACCEL_BRUTE_FORCE); tr.SetKernelVariant(variant)
    buf = torch.zeros(1920 * 1080 * 3, device="cuda")
    ts = []
    for rep in range(2):
        tr.Reset(); tr.ResetCounters(); tr.RunN(buf, 4); ts.append(tr.LastKernelMs())
    torch.cuda.synchronize()
    imgs[name] = buf.cpu().numpy().tobytes()
    print(f"{name}: {min(ts):.1f} ms per 4 iterations, {tr.RayCount()/min(ts)/1e6:.3f} G bounces/s, launch {tr.LastLaunch()}", flush=True)
    tr.Release()
print("images identical:", imgs["path pool"] == imgs["stealing"])
