"""PT_INTEGRATOR_NEE_MIS on the GPU against the CPU oracle's restatement of the same integrator.

Bar: BIT-EXACT images, hit-index sequences, per-path radiance and ray counts (shadow rays
included).  The integrator has no counterpart in the reference (its Trace never samples lights),
so parity against the reference is unpinned for the mode as a whole; its building blocks are
pinned to the reference on the CPU (test_nee_components.py) and its expectation to the
reference's estimator (statistically, here and there).
"""
import numpy as np
import pytest

from conftest import load_golden
from test_gpu_parity import render
from test_gpu_bvh import cloud, materials, room

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    torch.cuda.set_device(0)
    return torch


def check_against_oracle(pt, O, torch, cam, mats, sph, lights, depth, seed, iterations=2, threads=8):
    ref, ref_hits, ref_rad, ref_rays = O.oracle_render(cam.pod, mats, sph, lights, seed, depth, n_iterations=iterations,
                                                       want_paths=True, n_threads=threads,
                                                       integrator=O.INTEGRATOR_NEE_MIS)
    tr = pt.Pathtracer(cam, depth, seed)
    tr.Init(mats, sph, lights)
    tr.SetIntegrator(pt.INTEGRATOR_NEE_MIS)
    tr.ResetCounters()
    img = render(torch, tr, 1, iterations).cpu().numpy()
    assert tr.RayCount() == ref_rays
    w, h = cam.Width(), cam.Height()
    for k in range(iterations):
        hits, rad, rgb, _ = tr.DebugTrace(0, 0, w, h, k + 1)
        assert np.array_equal(hits, ref_hits[k]), k
        bad = np.argwhere(rad.view(np.uint32) != ref_rad[k].view(np.uint32))
        assert bad.size == 0, (k, bad[:3], rad[tuple(bad[0][:3])], ref_rad[k][tuple(bad[0][:3])])
    assert img.tobytes() == ref.tobytes()
    # per-iteration launches accumulate to the same image
    img2 = render(torch, tr, 1, iterations, batched=False).cpu().numpy()
    assert img2.tobytes() == ref.tobytes()
    tr.Release()
    return ref


@pytest.mark.parametrize("scene_id,size,depth,seed", [("SCENE_DEMO", (64, 48), 8, 7), ("SCENE_DEMO", (37, 21), 0, 3),
                                                      ("SCENE_MIXED", (64, 64), 16, 11)])
def test_builtin_scenes_match_the_oracle(pt, O, torch_cuda, scene_id, size, depth, seed):
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(getattr(pt, scene_id), 1)
    cam = pt.Camera(eye, look, up, size[0], size[1], 0, 0, fov)
    check_against_oracle(pt, O, torch_cuda, cam, mats, sph, lights, depth, seed)


def test_point_light_and_several_lights(pt, O, torch_cuda):
    """Two area lights (one small, one the demo's) and a point light: the light choice, the
    segment test of point lights and the 1/nLights weighting all take part."""
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO)
    extra = np.zeros(1, pt.SPHERE_DTYPE)
    extra["pos"], extra["radius"], extra["material_id"], extra["light_id"] = [[-20, 55, -30]], 3.0, 3, 1
    sph = np.concatenate([sph, extra])
    lights = np.concatenate([lights, np.zeros(2, pt.LIGHT_DTYPE)])
    lights["type"][1], lights["u_bits"][1, 0], lights["intensity"][1] = 1, len(sph) - 1, [30, 20, 10]
    lights["type"][2] = 0
    lights["u_bits"][2] = np.array([25.0, 60.0, 10.0], np.float32).view(np.uint32)
    lights["intensity"][2] = [5000.0, 5000.0, 6000.0]
    cam = pt.Camera(eye, look, up, 48, 40, 0, 0, fov)
    check_against_oracle(pt, O, torch_cuda, cam, mats, sph, lights, 6, 21)
    # the point light is only ever seen through direct-light sampling
    means = []
    for integrator in (pt.INTEGRATOR_PATH, pt.INTEGRATOR_NEE_MIS):
        tr = pt.Pathtracer(cam, 6, 21)
        tr.Init(mats, sph, lights)
        tr.SetIntegrator(integrator)
        means.append(float(render(torch_cuda, tr, 1, 64).mean()))
        tr.Release()
    assert means[1] > 1.2 * means[0]


def test_light_of_unknown_type_samples_to_black(pt, O, torch_cuda):
    """SampleLight returns black for a type that is neither POINT_LIGHT nor AREA_LIGHT (scene.h:169);
    such a light still emits when a path hits the sphere that carries it (Light::L does not look at
    the type, light.h:43-45).  Its union bits are arbitrary and must never be read as a sphere index."""
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO)
    lights = np.concatenate([lights, np.zeros(2, pt.LIGHT_DTYPE)])
    lights["type"][1], lights["u_bits"][1], lights["intensity"][1] = 2, [0xdeadbeef, 0x7fffffff, 5], [40, 40, 40]
    lights["type"][2], lights["u_bits"][2], lights["intensity"][2] = -7, [0xffffffff, 0, 0], [9, 9, 9]
    sph = sph.copy()
    sph["light_id"][6] = 1          # the mirror ball carries the type-2 light
    cam = pt.Camera(eye, look, up, 48, 40, 0, 0, fov)
    check_against_oracle(pt, O, torch_cuda, cam, mats, sph, lights, 6, 4)
    # the plain integrator never samples lights: same scene, bit-exact too
    ref, _, _, rays = O.oracle_render(cam.pod, mats, sph, lights, 4, 6, n_iterations=2)
    tr = pt.Pathtracer(cam, 6, 4)
    tr.Init(mats, sph, lights)
    tr.ResetCounters()
    assert render(torch_cuda, tr, 1, 2).cpu().numpy().tobytes() == ref.tobytes()
    assert tr.RayCount() == rays
    tr.Release()


def test_large_scene_through_the_bvh_and_the_tiled_scan(pt, O, torch_cuda):
    rng = np.random.default_rng(8)
    mats = materials(pt, [(0, 0.0), (1, 0.9), (2, 1.33), (0, 0.0)])
    walls, lights = room(pt, (0, 40, -20), 60)
    sph = np.concatenate([walls, cloud(pt, rng, 1500, (0, 40, -20), 50, 0.5, 3.0, 4)])
    cam = pt.Camera([0, 45, 35], [0, 38, -20], [0, 1, 0], 64, 40, 0, 0, 60.0)
    ref = check_against_oracle(pt, O, torch_cuda, cam, mats, sph, lights, 8, 5, iterations=1)
    tr = pt.Pathtracer(cam, 8, 5)
    tr.Init(mats, sph, lights)
    assert tr.AccelNodes() > 0
    tr.SetIntegrator(pt.INTEGRATOR_NEE_MIS)
    tr.SetAccel(pt.ACCEL_BRUTE_FORCE)
    assert render(torch_cuda, tr, 1, 1).cpu().numpy().tobytes() == ref.tobytes()
    tr.SetAccel(pt.ACCEL_AUTO)
    tr.SetAccelBudget(3)
    assert render(torch_cuda, tr, 1, 1).cpu().numpy().tobytes() == ref.tobytes()
    tr.Release()


def test_integrator_switch_and_restrictions(pt, torch_cuda):
    g = load_golden("demo_64x48_d8_it1-4")
    tr = pt.Pathtracer(g["camera_pod"], int(g["max_bounces"]), int(g["seed"]))
    tr.Init(g["materials"], g["spheres"], g["lights"])
    tr.SetIntegrator(pt.INTEGRATOR_NEE_MIS)
    nee = render(torch_cuda, tr, 1, 4).cpu().numpy()
    tr.SetIntegrator(pt.INTEGRATOR_PATH)
    assert render(torch_cuda, tr, 1, 4).cpu().numpy().tobytes() == g["image"].tobytes()
    assert nee.tobytes() != g["image"].tobytes()
    with pytest.raises(pt.PtError):
        tr.SetIntegrator(5)
    tr.SetIntegrator(pt.INTEGRATOR_NEE_MIS)
    tr.SetKernelVariant(pt.VARIANT_WAVEFRONT)
    with pytest.raises(pt.PtError) as e:
        tr.Run(torch_cuda.zeros(64 * 48 * 3, device="cuda"))
    assert e.value.code == pt.PT_ERR_STATE
    # row bands work as in the default integrator
    tr.SetKernelVariant(0)
    total = torch_cuda.zeros(64 * 48 * 3, device="cuda")
    for rank in range(3):
        tr.SetRowBands(16, rank, 3)
        total += render(torch_cuda, tr, 1, 4)
    assert total.cpu().numpy().tobytes() == nee.tobytes()
    tr.Release()
    bad = g["lights"].copy()
    bad["u_bits"][0, 0] = 99
    with pytest.raises(pt.PtError):
        pt.Pathtracer(g["camera_pod"], 4, 1).Init(g["materials"], g["spheres"], bad)


def test_same_expectation_lower_noise_at_1080p_sizes(pt, torch_cuda):
    """Demo scene, 480x270, 512 iterations with either integrator: image means within 0.5 %,
    30x30-pixel block means within 3 %, and the NEE image is the smoother one."""
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO)
    cam = pt.Camera(eye, look, up, 480, 270, 0, 0, fov)
    imgs = []
    for integrator in (pt.INTEGRATOR_PATH, pt.INTEGRATOR_NEE_MIS):
        tr = pt.Pathtracer(cam, 8, 1)
        tr.Init(mats, sph, lights)
        tr.SetIntegrator(integrator)
        imgs.append(render(torch_cuda, tr, 1, 512).cpu().numpy().reshape(270, 480, 3).astype(np.float64))
        tr.Release()
    a, b = imgs
    assert abs(b.mean() / a.mean() - 1) < 0.005
    A = a.reshape(9, 30, 16, 30, 3).mean(axis=(1, 3, 4))
    B = b.reshape(9, 30, 16, 30, 3).mean(axis=(1, 3, 4))
    assert np.abs(B / A - 1).max() < 0.03
    rough = lambda im: float(((im[100:200, 20:100] - im[100:200, 21:101]) ** 2).mean())
    assert rough(b) < 0.7 * rough(a)
