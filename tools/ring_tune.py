"""Render time of the demo room with n extra small spheres (linear scan), to tune the sphere count up
to which the camera-ray ring pays: python tools/ring_tune.py [path/to/another/libptmi.so]"""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pt = importlib.import_module("cuda-pathtracer_amd")
if len(sys.argv) > 1:   # another build of the library (measurement only)
    pt.LIB_PATH = os.path.abspath(sys.argv[1])
mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO)
cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
buf = torch.zeros(1920*1080*3, device="cuda")
rng = np.random.default_rng(1)
for n in (8, 16, 24, 40, 64, 100):
    extra = np.zeros(n, pt.SPHERE_DTYPE)
    extra["pos"] = np.column_stack([rng.uniform(-45,45,n), rng.uniform(5,75,n), rng.uniform(-75,20,n)])
    extra["radius"] = rng.uniform(0.3,1.2,n)*(10000/n)**(1/3)
    extra["material_id"] = rng.integers(1,6,n); extra["light_id"] = -1
    scene = np.concatenate([sph, extra])
    tr = pt.Pathtracer(cam, 8, 1); tr.Init(mats, scene, lights)
    ts=[]
    for r in range(3):
        tr.Reset(); tr.RunN(buf, 32); ts.append(tr.LastKernelMs())
    print(n+len(sph), "spheres: %.2f ms" % min(ts), flush=True)
    tr.Release()
