"""One render of the 10 007-sphere scene (1080p, 16 iterations) with a given mapping of bounces to
launches (argument: PT_TUNE_WAVEFRONT_WHOLE_PATHS value), for counter passes under rocprofv3."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pt = importlib.import_module("cuda-pathtracer_amd")
mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_SYNTH10K, 1)
cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
buf = torch.zeros(1920 * 1080 * 3, device="cuda")
tr = pt.Pathtracer(cam, 8, 1); tr.Init(mats, sph, lights)
tr.SetTuning(pt.TUNE_WAVEFRONT_WHOLE_PATHS, int(sys.argv[1]))
torch.cuda.synchronize(); t0 = time.perf_counter()
tr.RunN(buf, 16); torch.cuda.synchronize()
print(f"mode {sys.argv[1]}: {(time.perf_counter() - t0) * 1e3:.1f} ms (first render, allocations included)")
