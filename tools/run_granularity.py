"""64 iterations of the demo scene at 1080p as launches of 1 .. 64 iterations (runs on the GPU box):
the path pool (short launches under PT_VARIANT_AUTO) against the work-stealing megakernel, and the
same for the 71-sphere scene at depth 16."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pt = importlib.import_module("cuda-pathtracer_amd")
for scene, depth in ((pt.SCENE_DEMO, 8), (pt.SCENE_MIXED, 16)):
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(scene, 1)
    cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
    buf = torch.zeros(1920 * 1080 * 3, device="cuda")
    tr = pt.Pathtracer(cam, depth, 1); tr.Init(mats, sph, lights)
    for n_per in (1, 2, 4, 8, 16, 64):
        line = f"scene {scene} {64 // n_per} launches x {n_per} iterations:"
        for pool in (1024, 0):
            tr.SetTuning(pt.TUNE_POOL_ITERATIONS, pool)
            best = 1e9
            for rep in range(2):
                tr.Reset(); tr.ResetCounters()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for k in range(64 // n_per):
                    if n_per == 1: tr.Run(buf)
                    else: tr.RunN(buf, n_per)
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            line += f"  {'path pool' if pool else 'megakernel'} {best*1e3:.1f} ms ({tr.RayCount()/best/1e9:.1f} G bounces/s)"
        # the same launches without the wait between them: consecutive calls alternate between the two
        # buffer sets and streams (pt_run_async), one pt_wait at the end
        tr.SetTuning(pt.TUNE_POOL_ITERATIONS, 1024)
        best = 1e9
        for rep in range(3):
            tr.Reset(); tr.ResetCounters()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(64 // n_per):
                tr.RunAsync(buf, n_per)
            tr.Wait(); best = min(best, time.perf_counter() - t0)
        line += f"  path pool, asynchronous calls + one wait {best*1e3:.1f} ms ({tr.RayCount()/best/1e9:.1f} G bounces/s)"
        print(line, flush=True)
    tr.Release()
