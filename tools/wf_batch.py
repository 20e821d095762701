"""Paths in flight and launch structure of the wavefront variant on the 10 007-sphere scene at
1080p, 16 iterations (runs on the GPU box): ms per render and an image hash for several batch sizes
(units of 2^20 paths; arguments) and the three mappings of bounces to launches."""
import hashlib, importlib, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pt = importlib.import_module("cuda-pathtracer_amd")
mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_SYNTH10K, 1)
cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
buf = torch.zeros(1920 * 1080 * 3, device="cuda")
tr = pt.Pathtracer(cam, 8, 1); tr.Init(mats, sph, lights)
for mode, mpaths in [(m, b) for m in (0, 2, 1) for b in ([int(a) for a in sys.argv[1:]] or [0, 8, 16, 64, 128, 136])]:
    tr.SetTuning(pt.TUNE_WAVEFRONT_WHOLE_PATHS, mode)
    tr.SetTuning(pt.TUNE_WAVEFRONT_BATCH, mpaths)
    best = 1e9
    for rep in range(3):
        tr.Reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
        tr.RunN(buf, 16); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"launches {mode} batch {mpaths:4d} Mi paths: {best*1e3:.1f} ms  footprint {tr.MemoryFootprint(16)}  {hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
