"""BVH traversal knobs on the 10 007-sphere scene at 1080p (runs on the GPU box): node steps per
leaf step, stack entries, refill threshold; prints ms per 8 iterations and an image hash."""
import hashlib, importlib, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pt = importlib.import_module("cuda-pathtracer_amd")
if os.environ.get("PT_LIB"):      # another build of the library (A/B)
    pt.LIB_PATH = os.path.abspath(os.environ["PT_LIB"])
mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_SYNTH10K, 1)
cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
buf = torch.zeros(1920 * 1080 * 3, device="cuda")
tr = pt.Pathtracer(cam, 8, 1); tr.Init(mats, sph, lights)
def run(label):
    best = 1e9
    for rep in range(3):
        tr.Reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
        tr.RunN(buf, 8); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"{label}: {best*1e3:.1f} ms  {hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
run("default")
quick = os.environ.get("PT_QUICK") == "1"
for steps in ((2,) if quick else (1, 2, 3, 4, 6)):
    tr.SetTuning(pt.TUNE_BVH_NODE_STEPS, steps); run(f"node_steps {steps}")
tr.SetTuning(pt.TUNE_BVH_NODE_STEPS, int(sys.argv[1]) if len(sys.argv) > 1 else 3)
for stack in ((8, 10, 12) if quick else (6, 8, 10, 12, 16, 20)):
    tr.SetTuning(pt.TUNE_BVH_STACK, stack); run(f"stack {stack}")
tr.SetTuning(pt.TUNE_BVH_STACK, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
for refill in (() if quick else (16, 24, 32, 40, 48)):
    tr.SetTuning(pt.TUNE_WAVEFRONT_REFILL, refill); run(f"refill {refill}")
