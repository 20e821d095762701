"""Measurement (GPU box, with tools/ab_sort/libptmi.so on LD_LIBRARY_PATH... loaded through PT_LIB): the
10 007-sphere scene at 1080p, one launch per bounce over the queues, one stream, with the survivors
of every bounce sorted inside their sub-queues (PT_EXPERIMENT_SORT: 0 off, 1 octant, 2 octant then
origin cell, 3 origin cell then octant, 4 coarse cell then octant).  Prints image hash and time; the
per-kernel times come from rocprofv3 --kernel-trace --stats around this script."""
import hashlib, importlib, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pt = importlib.import_module("cuda-pathtracer_amd")
mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_SYNTH10K, 1)
cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
buf = torch.zeros(1920 * 1080 * 3, device="cuda")
tr = pt.Pathtracer(cam, 8, 1); tr.Init(mats, sph, lights)
tr.SetTuning(pt.TUNE_WAVEFRONT_WHOLE_PATHS, int(os.environ.get("PT_WF_MODE", "0")))
tr.SetTuning(pt.TUNE_BUFFER_SETS, 1)
its = int(os.environ.get("PT_ITS", "4"))
for rep in range(2):
    tr.Reset(); tr.ResetCounters()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.RunN(buf, its)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"sort {os.environ.get('PT_EXPERIMENT_SORT', '0')}: {dt*1e3:.1f} ms for {its} iterations incl. sorting, rays {tr.RayCount()}, "
      f"image {hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
