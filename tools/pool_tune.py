"""Path pool (resident scenes under PT_VARIANT_AUTO) at 1080p x 64 iterations in one pt_run_n (runs
on the GPU box): paths in flight (units of 2^20) for the demo scene and the 71-sphere scene, against
the work-stealing megakernel."""
import hashlib, importlib, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pt = importlib.import_module("cuda-pathtracer_amd")
for scene, depth in ((pt.SCENE_DEMO, 8), (pt.SCENE_MIXED, 16)):
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(scene, 1)
    cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
    buf = torch.zeros(1920 * 1080 * 3, device="cuda")
    tr = pt.Pathtracer(cam, depth, 1); tr.Init(mats, sph, lights)
    def run(label):
        best = 1e9
        for rep in range(3):
            tr.Reset(); tr.ResetCounters(); torch.cuda.synchronize(); t0 = time.perf_counter()
            tr.RunN(buf, 64); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f"scene {scene} {label}: {best*1e3:.2f} ms  {tr.RayCount()/best/1e9:.1f} G bounces/s  "
              f"{hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
    tr.SetTuning(pt.TUNE_POOL_ITERATIONS, 0); run("megakernel")
    tr.SetTuning(pt.TUNE_POOL_ITERATIONS, 1 << 30)
    for mpaths in [int(a) for a in sys.argv[1:]] or [0, 17, 34, 67, 133, 266, 540]:
        tr.SetTuning(pt.TUNE_WAVEFRONT_BATCH, mpaths); run(f"pool, {mpaths} Mi paths in flight")
    tr.Release()
