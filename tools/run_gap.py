"""Where the time of per-iteration pt_run calls goes (runs on the GPU box): wall time of 64 x pt_run
at 1080p against the sum of the kernels' own durations (HIP events around each launch), for the
default variant and the plain megakernel, and the same for pt_run_n(1) without the per-call sync."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pt = importlib.import_module("cuda-pathtracer_amd")
mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO, 1)
cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
buf = torch.zeros(1920 * 1080 * 3, device="cuda")
for variant in (pt.VARIANT_AUTO, pt.VARIANT_REGENERATION):
    tr = pt.Pathtracer(cam, 8, 1); tr.Init(mats, sph, lights); tr.SetKernelVariant(variant)
    for mode in ("run (sync per call)", "run_n(1, sync=False)"):
        for rep in range(2):
            tr.Reset(); tr.ResetCounters()
            torch.cuda.synchronize(); t0 = time.perf_counter(); kernel_ms = 0.0
            for k in range(64):
                if mode.startswith("run ("):
                    tr.Run(buf); kernel_ms += tr.LastKernelMs()
                else:
                    tr.RunN(buf, 1, sync=False)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"variant {variant} {mode}: wall {dt*1e3:.1f} ms, kernels {kernel_ms:.1f} ms, {tr.RayCount()/dt/1e9:.1f} G bounces/s", flush=True)
    tr.Release()
