"""One pt_run_n of 64 iterations at 1080p (the headline workload) against the path pool's workgroups
per CU (PT_TUNE_POOL_WORKGROUPS) and paths in flight (PT_TUNE_WAVEFRONT_BATCH): the batches of the
call alternate between two streams, so fewer workgroups per launch let two batches be resident side by
side.  Runs on the GPU box."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pt = importlib.import_module("cuda-pathtracer_amd")
for scene, depth, its in ((pt.SCENE_DEMO, 8, 64), (pt.SCENE_MIXED, 16, 32)):
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(scene, 1)
    cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
    buf = torch.zeros(1920 * 1080 * 3, device="cuda")
    tr = pt.Pathtracer(cam, depth, 1); tr.Init(mats, sph, lights)
    for mpaths in (0, 133, 66, 33):
        tr.SetTuning(pt.TUNE_WAVEFRONT_BATCH, mpaths)
        line = f"scene {scene}, {its} iterations, {mpaths or 266} Mi paths in flight, workgroups per CU:"
        for wg in (0, 3, 4, 5, 6):
            tr.SetTuning(pt.TUNE_POOL_WORKGROUPS, wg)
            best = 1e9
            for rep in range(4):
                tr.Reset()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                tr.RunN(buf, its)
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            line += f"  {wg}: {best*1e3:.2f} ms"
        print(line, flush=True)
    tr.Release()
