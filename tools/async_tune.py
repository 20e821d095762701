"""64 x pt_run_async of one iteration + one pt_wait at 1080p against the workgroups per CU of the
path pool's launches (PT_TUNE_POOL_WORKGROUPS); runs on the GPU box."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pt = importlib.import_module("cuda-pathtracer_amd")
for scene, depth in ((pt.SCENE_DEMO, 8), (pt.SCENE_MIXED, 16)):
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(scene, 1)
    cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
    buf = torch.zeros(1920 * 1080 * 3, device="cuda")
    tr = pt.Pathtracer(cam, depth, 1); tr.Init(mats, sph, lights)
    for n_per in (1, 2, 4):
        line = f"scene {scene}, {64 // n_per} asynchronous calls x {n_per} iterations, workgroups per CU:"
        for wg in (2, 3, 4, 5, 6, 7, 8):
            tr.SetTuning(pt.TUNE_POOL_WORKGROUPS, wg)
            best = 1e9
            for rep in range(3):
                tr.Reset()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for k in range(64 // n_per):
                    tr.RunAsync(buf, n_per)
                tr.Wait(); best = min(best, time.perf_counter() - t0)
            line += f"  {wg}: {best*1e3:.1f} ms"
        print(line, flush=True)
    tr.Release()
