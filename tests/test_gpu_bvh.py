"""The sphere BVH against the linear scan it replaces (scene.h:71-94; the reference leaves the
acceleration structure as "TODO kd-tree", scene.h:33).

Bar: BIT-EXACT.  Intersect returns the lexicographic minimum of (t, index) over the spheres
whose fp32 test yields a valid t; the hierarchy may change the order spheres are visited in,
never the result.  Every case renders the same scene twice on the GPU (PT_ACCEL_AUTO /
PT_ACCEL_BRUTE_FORCE) and compares hit-index sequences, per-path radiance, frame buffers and
ray counts; the small cases are also checked against the CPU oracle.
"""
import numpy as np
import pytest

from test_gpu_parity import render

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    torch.cuda.set_device(0)
    return torch


def materials(pt, kinds):
    """kinds: list of (type, coef)."""
    rng = np.random.default_rng(7)
    m = np.zeros(len(kinds), pt.MATERIAL_DTYPE)
    m["color"] = rng.uniform(0.3, 0.95, (len(kinds), 3)).astype(np.float32)
    m["type"] = [k[0] for k in kinds]
    m["coef"] = [k[1] for k in kinds]
    return m


def room(pt, centre, half):
    """Six radius-1e5 walls around `centre` (the demo room's construction, main.cpp:137-149)
    plus a spherical area light under the ceiling: 7 spheres, the light is index 6."""
    c = np.asarray(centre, np.float64)
    big = 1e5
    s = np.zeros(7, pt.SPHERE_DTYPE)
    offs = [(-(big + half), 0, 0), (big + half, 0, 0), (0, -(big + half), 0), (0, big + half, 0),
            (0, 0, -(big + half)), (0, 0, big + half)]
    for i, o in enumerate(offs):
        s["pos"][i] = (c + np.array(o)).astype(np.float32)
        s["radius"][i] = big
        s["material_id"][i] = 0
        s["light_id"][i] = -1
    s["pos"][6] = (c + np.array([0, half * 0.8, 0])).astype(np.float32)
    s["radius"][6] = half * 0.15
    s["material_id"][6] = 0
    s["light_id"][6] = 0
    lights = np.zeros(1, pt.LIGHT_DTYPE)
    lights["type"] = 1
    lights["u_bits"][0, 0] = 6
    lights["intensity"] = [[12, 12, 12]]
    return s, lights


def cloud(pt, rng, n, centre, half, rmin, rmax, n_mats):
    s = np.zeros(n, pt.SPHERE_DTYPE)
    s["pos"] = (rng.uniform(-half, half, (n, 3)) + np.asarray(centre)).astype(np.float32)
    s["radius"] = rng.uniform(rmin, rmax, n).astype(np.float32)
    s["material_id"] = rng.integers(0, n_mats, n)
    s["light_id"] = -1
    return s


def compare_accel(pt, torch, cam, mats, sph, lights, depth, seed=3, iterations=2, trace_rect=None,
                  expect_bvh=True, variants=(0,)):
    """Renders with and without the BVH; returns the BVH instance's image."""
    out = {}
    for accel in (pt.ACCEL_AUTO, pt.ACCEL_BRUTE_FORCE):
        for variant in variants:
            tr = pt.Pathtracer(cam, depth, seed)
            tr.Init(mats, sph, lights)
            assert (tr.AccelNodes() > 0) == expect_bvh
            tr.SetAccel(accel)
            tr.SetKernelVariant(variant)
            tr.ResetCounters()
            img = render(torch, tr, 1, iterations).cpu().numpy()
            rays = tr.RayCount()
            trace = None
            if trace_rect is not None:
                trace = tr.DebugTrace(*trace_rect, 1)
            out[(accel, variant)] = (img, rays, trace)
            tr.Release()
    ref_img, ref_rays, ref_trace = out[(pt.ACCEL_BRUTE_FORCE, variants[0])]
    for key, (img, rays, trace) in out.items():
        if trace is not None:
            bad = np.argwhere(trace[0] != ref_trace[0])
            assert bad.size == 0, (key, bad[:4], trace[0][tuple(bad[0][:3])], ref_trace[0][tuple(bad[0][:3])])
            assert trace[1].tobytes() == ref_trace[1].tobytes(), key
        assert rays == ref_rays, key
        assert img.tobytes() == ref_img.tobytes(), key
    return ref_img


def test_synth10k_bvh_equals_scan_and_golden(pt, torch_cuda):
    """The 10 007-sphere benchmark scene at 1080p: BVH == tiled scan bit for bit, in both the
    megakernel and the one-launch-per-bounce variant."""
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_SYNTH10K, 1)
    cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
    compare_accel(pt, torch_cuda, cam, mats, sph, lights, 8, seed=1, iterations=1,
                  trace_rect=(800, 500, 1120, 532), variants=(0, pt.VARIANT_WAVEFRONT))


def test_mirror_cloud_depth_16(pt, O, torch_cuda):
    """Small mirror and glass spheres only: mirror directions are never re-normalised
    (globals.h:107-109), so |dir| drifts from 1 along a path and the sphere test reports hits
    off the geometric sphere -- the case the per-ray box margin exists for."""
    rng = np.random.default_rng(11)
    mats = materials(pt, [(0, 0.0), (1, 0.95), (2, 1.5), (1, 0.8)])
    walls, lights = room(pt, (0, 40, -20), 60)
    balls = cloud(pt, rng, 3000, (0, 40, -20), 55, 0.2, 2.5, 4)
    balls["material_id"] = rng.choice([1, 2, 3], 3000)
    sph = np.concatenate([walls, balls])
    cam = pt.Camera([0, 45, 35], [0, 38, -20], [0, 1, 0], 256, 192, 0, 0, 60.0)
    img = compare_accel(pt, torch_cuda, cam, mats, sph, lights, 16, trace_rect=(0, 0, 256, 192))
    ref, _, _, _ = O.oracle_render(cam.pod, mats, sph, lights, 3, 16, n_iterations=2,
                                   rect=(0, 90, 256, 94), n_threads=8)
    assert img.reshape(192, 256, 3)[90:94].tobytes() == ref[90:94].tobytes()


def test_equal_distance_ties_keep_the_lowest_index(pt, torch_cuda):
    """Every sphere of the cloud exists three times at scattered indices: the scan keeps the
    lowest index among equal t (strict <, scene.h:82-88), and so must the tree."""
    rng = np.random.default_rng(5)
    mats = materials(pt, [(0, 0.0), (1, 0.9), (2, 1.33), (0, 0.0)])
    walls, lights = room(pt, (0, 40, -20), 60)
    base = cloud(pt, rng, 700, (0, 40, -20), 50, 1.0, 4.0, 4)
    dup = np.concatenate([base, base, base])
    perm = rng.permutation(len(dup))
    dup = dup[perm]
    # copies share geometry but not the material: a wrong winner changes the image
    dup["material_id"] = rng.integers(0, 4, len(dup))
    sph = np.concatenate([walls, dup])
    cam = pt.Camera([0, 45, 35], [0, 38, -20], [0, 1, 0], 160, 120, 0, 0, 60.0)
    compare_accel(pt, torch_cuda, cam, mats, sph, lights, 8, trace_rect=(0, 0, 160, 120))


@pytest.mark.parametrize("case", ["far_offset", "far_camera", "tiny", "clustered", "no_walls", "flat"])
def test_awkward_geometry(pt, O, torch_cuda, case):
    rng = np.random.default_rng(100 + len(case))
    mats = materials(pt, [(0, 0.0), (1, 0.9), (2, 1.33), (0, 0.0)])
    centre, half, eye_off = np.array([0.0, 40.0, -20.0]), 60.0, np.array([0.0, 5.0, 55.0])
    rmin, rmax, n = 0.3, 1.5, 4000
    if case == "far_offset":          # large coordinates: coarse fp32 grid, large |op|-relative errors
        centre = np.array([8000.0, -3000.0, 5000.0])
    elif case == "far_camera":        # the front wall is left out, the camera sits far outside
        eye_off = np.array([0.0, 300.0, 1500.0])
    elif case == "tiny":
        rmin, rmax = 0.004, 0.05
    walls, lights = room(pt, centre, half)
    if case == "clustered":           # a few dense clumps, many overlaps
        seeds = rng.uniform(-40, 40, (12, 3))
        balls = cloud(pt, rng, n, centre, 1.0, rmin, rmax, 4)
        balls["pos"] = (centre + seeds[rng.integers(0, 12, n)] + rng.normal(0, 2.0, (n, 3))).astype(np.float32)
    elif case == "flat":              # all centres in one plane: degenerate split axis extents
        balls = cloud(pt, rng, n, centre, 50, rmin, rmax, 4)
        balls["pos"][:, 1] = np.float32(centre[1])
    else:
        balls = cloud(pt, rng, n, centre, 50, rmin, rmax, 4)
    sph = np.concatenate([walls, balls])
    if case == "no_walls":            # nothing in the always-tested list but the light: rays escape
        sph = sph[6:].copy()
        lights["u_bits"][0, 0] = 0
    if case == "far_camera":
        sph = np.concatenate([sph[:5], sph[6:]])
        lights["u_bits"][0, 0] = 5
    eye = centre + eye_off
    fov = 6.0 if case == "far_camera" else 60.0
    cam = pt.Camera(eye.tolist(), (centre + [0, -2, 0]).tolist(), [0, 1, 0], 200, 120, 0, 0, fov)
    img = compare_accel(pt, torch_cuda, cam, mats, sph, lights, 10, trace_rect=(0, 0, 200, 120))
    ref, _, _, _ = O.oracle_render(cam.pod, mats, sph, lights, 3, 10, n_iterations=2,
                                   rect=(0, 60, 200, 62), n_threads=8)
    assert img.reshape(120, 200, 3)[60:62].tobytes() == ref[60:62].tobytes()


def test_scene_the_builder_declines_falls_back_to_the_scan(pt, O, torch_cuda):
    """More than 64 spheres far larger than the rest: no hierarchy is built, the scan runs."""
    rng = np.random.default_rng(2)
    mats = materials(pt, [(0, 0.0), (1, 0.9)])
    walls, lights = room(pt, (0, 40, -20), 60)
    giants = cloud(pt, rng, 80, (0, 40, -20), 50, 30.0, 40.0, 2)
    balls = cloud(pt, rng, 900, (0, 40, -20), 50, 0.5, 1.0, 2)
    sph = np.concatenate([walls, giants, balls])
    cam = pt.Camera([0, 45, 35], [0, 38, -20], [0, 1, 0], 64, 48, 0, 0, 60.0)
    img = compare_accel(pt, torch_cuda, cam, mats, sph, lights, 6, expect_bvh=False)
    ref, _, _, _ = O.oracle_render(cam.pod, mats, sph, lights, 3, 6, n_iterations=2, n_threads=8)
    assert img.tobytes() == ref.tobytes()


@pytest.mark.parametrize("budget", [1, 7, 40, 10**9])
def test_visit_budget_does_not_change_results(pt, O, torch_cuda, budget):
    """Budget 1: practically every ray is finished by the wave-wide scan; 10^9: none is."""
    rng = np.random.default_rng(21)
    mats = materials(pt, [(0, 0.0), (1, 0.9), (2, 1.33), (0, 0.0)])
    walls, lights = room(pt, (0, 40, -20), 60)
    sph = np.concatenate([walls, cloud(pt, rng, 1200, (0, 40, -20), 50, 0.5, 3.0, 4)])
    cam = pt.Camera([0, 45, 35], [0, 38, -20], [0, 1, 0], 96, 64, 0, 0, 60.0)
    ref, ref_hits, ref_rad, ref_rays = O.oracle_render(cam.pod, mats, sph, lights, 5, 8, n_iterations=2,
                                                       want_paths=True, n_threads=8)
    for variant in (0, pt.VARIANT_WAVEFRONT):
        tr = pt.Pathtracer(cam, 8, 5)
        tr.Init(mats, sph, lights)
        assert tr.AccelNodes() > 0
        tr.SetAccelBudget(budget)
        tr.SetKernelVariant(variant)
        tr.ResetCounters()
        img = render(torch_cuda, tr, 1, 2).cpu().numpy()
        assert tr.RayCount() == ref_rays
        assert img.tobytes() == ref.tobytes(), (budget, variant)
        hits, rad, _, _ = tr.DebugTrace(0, 0, 96, 64, 2)
        assert np.array_equal(hits, ref_hits[1])
        assert rad.tobytes() == ref_rad[1].tobytes()
        tr.Release()


@pytest.mark.parametrize("size", [(16, 16), (200, 120), (333, 77)])
def test_wavefront_dynamic_fetch_and_auto_variant(pt, O, torch_cuda, size):
    """BVH scenes default to the queue-fed persistent waves (PT_VARIANT_AUTO); every way of
    scheduling paths onto lanes gives the same bits: forced megakernel, static chunks (101),
    refill thresholds 2 and 64, the default, whole paths per launch."""
    rng = np.random.default_rng(31)
    mats = materials(pt, [(0, 0.0), (1, 0.9), (2, 1.33), (0, 0.0)])
    walls, lights = room(pt, (0, 40, -20), 60)
    sph = np.concatenate([walls, cloud(pt, rng, 2500, (0, 40, -20), 50, 0.4, 2.0, 4)])
    cam = pt.Camera([0, 45, 35], [0, 38, -20], [0, 1, 0], size[0], size[1], 0, 0, 60.0)
    imgs, rays = {}, {}
    for variant in (pt.VARIANT_AUTO, 0, 100, 101, 102, 164, "whole paths 1", "whole paths 2"):
        tr = pt.Pathtracer(cam, 8, 9)
        tr.Init(mats, sph, lights)
        if str(variant).startswith("whole paths"):
            # the persistent waves follow every path to its end: from the camera ray on (1), or
            # from the first bounce on after a launch for the camera rays (2)
            mode = int(variant[-1])
            tr.SetTuning(pt.TUNE_WAVEFRONT_WHOLE_PATHS, mode)
            hits, rad, rgb, n = tr.DebugTrace(0, 0, size[0], min(size[1], 16), 3)
            tr.SetTuning(pt.TUNE_WAVEFRONT_WHOLE_PATHS, 0)
            hits0, rad0, rgb0, n0 = tr.DebugTrace(0, 0, size[0], min(size[1], 16), 3)
            assert np.array_equal(hits, hits0) and rad.tobytes() == rad0.tobytes() and rgb.tobytes() == rgb0.tobytes() and n == n0
            tr.SetTuning(pt.TUNE_WAVEFRONT_WHOLE_PATHS, mode)
        elif variant != pt.VARIANT_AUTO:
            tr.SetKernelVariant(variant)
        tr.ResetCounters()
        imgs[variant] = render(torch_cuda, tr, 2, 3).cpu().numpy().tobytes()   # iterations 2..4
        rays[variant] = tr.RayCount()
        if variant == pt.VARIANT_AUTO:
            # auto falls back to the megakernel where the wavefront variant cannot serve
            tr.SetRowBands(16, 0, 1)
            tr.SetIntegrator(pt.INTEGRATOR_NEE_MIS)
            render(torch_cuda, tr, 1, 1)
        tr.Release()
    assert len(set(imgs.values())) == 1 and len(set(rays.values())) == 1
    ref, _, _, ref_rays = O.oracle_render(cam.pod, mats, sph, lights, 9, 8, first_iteration=2, n_iterations=3,
                                          rect=(0, size[1] // 2, size[0], size[1] // 2 + 2), n_threads=8)
    got = np.frombuffer(imgs[0], np.float32).reshape(size[1], size[0], 3)
    # rows of a resumed range: the oracle accumulates from a zero buffer, so compare via a fresh GPU run
    tr = pt.Pathtracer(cam, 8, 9)
    tr.Init(mats, sph, lights)
    buf = torch_cuda.zeros(size[0] * size[1] * 3, device="cuda")
    tr.SetIteration(1)
    tr.RunN(buf, 3)
    assert buf.cpu().numpy().tobytes() == imgs[0]
    rows = slice(size[1] // 2, size[1] // 2 + 2)
    assert got[rows].tobytes() == ref[rows].tobytes()
    tr.Release()


def test_random_scenes_fuzz(pt, torch_cuda, monkeypatch):
    """40 scenes of tools/fuzz_bvh.py (random sizes, radius spreads, mirror / glass mixes, offsets,
    clumps, sheets, grids, far cameras): BVH == scan in the megakernel and both wavefront forms.
    (5 400 further seeds were run once without a mismatch.)"""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("fuzz_bvh", os.path.join(ROOT, "tools", "fuzz_bvh.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    monkeypatch.setattr("sys.argv", ["fuzz_bvh.py", "40", "777000"])
    assert fuzz.main() == 0


def test_small_scenes_have_no_bvh(pt, torch_cuda):
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_MIXED)
    tr = pt.Pathtracer(pt.Camera(eye, look, up, 32, 32, 0, 0, fov), 4, 1)
    tr.Init(mats, sph, lights)
    assert tr.AccelNodes() == 0
    with pytest.raises(pt.PtError):
        tr.SetAccel(7)
    tr.Release()


def test_two_buffer_sets_respect_the_callers_stream(pt, torch_cuda):
    """A render of several wavefront batches alternates between two buffer sets on the caller's
    stream and a stream of the instance; work the caller enqueues on its stream before and after
    an asynchronous pt_run_n is ordered against all of it: the frame is cleared by a kernel on the
    caller's stream just before, and copied by one just after, without any host synchronisation."""
    torch = torch_cuda
    rng = np.random.default_rng(77)
    mats = materials(pt, [(0, 0.0), (1, 0.9), (2, 1.33), (0, 0.0)])
    walls, lights = room(pt, (0, 40, -20), 60)
    sph = np.concatenate([walls, cloud(pt, rng, 1200, (0, 40, -20), 50, 0.5, 2.5, 4)])
    cam = pt.Camera([0, 45, 35], [0, 38, -20], [0, 1, 0], 320, 200, 0, 0, 60.0)
    tr = pt.Pathtracer(cam, 6, 3)
    tr.Init(mats, sph, lights)
    ref = render(torch, tr, 1, 7).cpu().numpy().tobytes()          # one batch, default stream
    tr.Release()
    tr = pt.Pathtracer(cam, 6, 3)
    tr.Init(mats, sph, lights)
    tr.SetMemoryLimit(2 * 56 * 320 * 200 * 4 + 2**20)              # two sets of one iteration each: seven batches
    side = torch.cuda.Stream()
    tr.SetStream(side.cuda_stream)
    buf = torch.full((320 * 200 * 3,), 123.0, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for rep in range(3):
            buf.fill_(float("nan"))                               # iteration 1 overwrites whatever is there
            tr.Reset()
            tr.RunN(buf, 7, sync=False)
            out = buf.clone()                                     # on the caller's stream, behind both sets
    side.synchronize()
    assert out.cpu().numpy().tobytes() == ref
    assert buf.cpu().numpy().tobytes() == ref
    tr.Release()


def test_row_bands_through_the_wavefront_launches(pt, torch_cuda):
    """Tile split on a BVH scene: every rank's batch holds the rows of its 16-row bands only (the
    last band of the frame is a partial one), the queue-fed launches render them, and the ranks'
    frames add up to the single-instance image bit for bit; so do the ray counts."""
    torch = torch_cuda
    rng = np.random.default_rng(5)
    mats = materials(pt, [(0, 0.0), (1, 0.9), (2, 1.33), (0, 0.0)])
    walls, lights = room(pt, (0, 40, -20), 60)
    sph = np.concatenate([walls, cloud(pt, rng, 900, (0, 40, -20), 50, 0.5, 2.5, 4)])
    cam = pt.Camera([0, 45, 35], [0, 38, -20], [0, 1, 0], 150, 107, 0, 0, 60.0)   # 6 full bands + 11 rows
    tr = pt.Pathtracer(cam, 8, 2)
    tr.Init(mats, sph, lights)
    tr.ResetCounters()
    ref = render(torch, tr, 1, 3)
    ref_rays = tr.RayCount()
    assert tr.LastLaunch() == pt.LAUNCH_WAVEFRONT_WHOLE_PATHS
    for world in (3, 8):
        total = torch.zeros_like(ref)
        rays = 0
        for rank in range(world):
            tr.SetRowBands(16, rank, world)
            tr.ResetCounters()
            assert tr.MemoryFootprint(3)[1] >= tr.MemoryFootprint(3)[0]   # also for a rank without a band
            part = render(torch, tr, 1, 3)
            if rank * 16 < 107:      # with 8 ranks the last one owns no band of this frame
                assert tr.LastLaunch() == pt.LAUNCH_WAVEFRONT_WHOLE_PATHS
            rays += tr.RayCount()
            total += part
        assert total.cpu().numpy().tobytes() == ref.cpu().numpy().tobytes(), world
        assert rays == ref_rays
    tr.Release()
