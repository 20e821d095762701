"""T2 of the parity ladder: the HIP kernels (through the C ABI) against the CPU oracle and the
golden vectors of the reference's host build.

Bar: hit-index sequences, per-path radiance, pixel colours and the running mean are BIT-EXACT
(integer work and, because both sides evaluate the same IEEE fp32 expression trees without
fused multiply-add, the floating-point results too).  The only tolerance in this file is for
PT_ACCUM_SUM vs the running mean, which sum in a different order: rtol 2e-6 * sqrt(n) + atol 1e-6.
"""
import hashlib

import numpy as np
import pytest

from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    torch.cuda.set_device(0)
    return torch


def make_tracer(pt, g, variant=0):
    tr = pt.Pathtracer(g["camera_pod"], int(g["max_bounces"]), int(g["seed"]))
    tr.Init(g["materials"], g["spheres"], g["lights"])
    tr.SetKernelVariant(variant)
    return tr


def render(torch, tr, first_iteration, n_iterations, batched=True, buf=None):
    if buf is None:
        buf = torch.zeros(tr.height * tr.width * 3, dtype=torch.float32, device="cuda")
    tr.Reset()
    tr.SetIteration(first_iteration - 1)
    if batched:
        tr.RunN(buf, n_iterations)
    else:
        for _ in range(n_iterations):
            tr.Run(buf)
    torch.cuda.synchronize()
    return buf


# The kernels a render goes through: what ships by default (PT_VARIANT_AUTO: the path pool, or the
# per-bounce wavefront launches for BVH scenes), the plain megakernel and the work-stealing one.
SHIPPED_VARIANTS = [-1, 0, 500]   # pt.VARIANT_AUTO (path pool / BVH wavefront), pt.VARIANT_REGENERATION, pt.VARIANT_STEALING


@pytest.mark.parametrize("variant", SHIPPED_VARIANTS)
@pytest.mark.parametrize("name", golden_names())
def test_image_matches_golden_bit_exact(pt, torch_cuda, name, variant):
    """Frame buffer after the fixture's iteration range == the reference host build's."""
    g = load_golden(name)
    tr = make_tracer(pt, g, variant)
    tr.ResetCounters()
    buf = render(torch_cuda, tr, int(g["first_iteration"]), int(g["n_iterations"]))
    img = buf.cpu().numpy().reshape(g["image"].shape)
    assert tr.RayCount() == int(g["ray_count"])
    assert img.tobytes() == g["image"].tobytes()
    assert tr.GetIteration() == int(g["first_iteration"]) + int(g["n_iterations"]) - 1
    tr.Release()


@pytest.mark.parametrize("variant", [-1, 100])
def test_radiance_mask_is_left_clean_between_batches_and_rectangles(pt, torch_cuda, variant):
    """The path pool and the queue-fed kernels store the radiance of the paths that met a light
    only and mark them in a mask that the resolve clears; nothing zeroes the buffers between
    batches.  Renders, parity-hook rectangles of another shape (which store every path) and
    batches cut by a memory limit share one buffer set here: every image must stay the golden one,
    and the per-path radiances of the hook must be the golden ones each time."""
    g = load_golden("demo_64x48_d8_it1-4")
    first, n = int(g["first_iteration"]), int(g["n_iterations"])
    tr = make_tracer(pt, g, variant)
    h, w = g["image"].shape[:2]
    for limit in (0, 13 * w * h * 4 * 2 + 2**20, 0):          # 0: no limit; then batches of one iteration in two sets
        tr.SetMemoryLimit(limit)
        img = render(torch_cuda, tr, first, n).cpu().numpy()
        assert img.tobytes() == g["image"].tobytes()
        hits, rad, rgb, _ = tr.DebugTrace(3, 5, 3 + 17, 5 + 9, first)
        assert np.array_equal(hits, g["hit_ids"][0][5:14, 3:20].astype(np.int32))
        assert rad.tobytes() == np.ascontiguousarray(g["path_radiance"][0][5:14, 3:20].astype(np.float32)).tobytes()
    img = render(torch_cuda, tr, first, n).cpu().numpy()
    assert img.tobytes() == g["image"].tobytes()
    tr.Release()


# + 3: whole pixel-iterations change hands even in a one-iteration launch (the default there: single paths)
@pytest.mark.parametrize("variant", SHIPPED_VARIANTS + [500 + 3, 100])
@pytest.mark.parametrize("name", golden_names())
def test_hit_ids_and_path_radiance_match_golden(pt, torch_cuda, name, variant):
    """pt_debug_trace runs the rectangle through the kernels a render of the instance would use
    (work stealing included: a stolen path is recorded by the lane that traced it), so the hit-index
    sequences and per-path radiances of the reference's host build are compared with what ships."""
    g = load_golden(name)
    tr = make_tracer(pt, g, variant)
    w, h = int(g["width"]), int(g["height"])
    hits_all, rad_all, rays = [], [], 0
    for it in range(int(g["first_iteration"]), int(g["first_iteration"]) + int(g["n_iterations"])):
        hits, rad, rgb, cnt = tr.DebugTrace(0, 0, w, h, it)
        hits_all.append(hits); rad_all.append(rad); rays += cnt
        # pixel colour is the in-order sum of the four paths times 1/4 (pathtracer.cu:94-101)
        expect = (((rad[:, :, 0] + rad[:, :, 1]) + rad[:, :, 2]) + rad[:, :, 3]) * np.float32(0.25)
        assert rgb.tobytes() == expect.astype(np.float32).tobytes()
    hits_all, rad_all = np.stack(hits_all), np.stack(rad_all)
    assert rays == int(g["ray_count"])
    assert sha(hits_all) == str(g["hit_ids_sha256"])
    assert sha(rad_all) == str(g["path_radiance_sha256"])
    if "hit_ids" in g.files:
        assert np.array_equal(hits_all, g["hit_ids"].astype(np.int32))
    tr.Release()


def test_matches_oracle_on_seeded_random_scenes(pt, O, torch_cuda):
    """Same seeded inputs through the HIP path and the oracle, sizes the oracle does in seconds."""
    from test_oracle_golden import _random_scene
    rng = np.random.default_rng(2024)
    eye, look, up = [0, 45, 79.5], [0, 35, 0], [0, 1, 0]
    for n, (w, h), mb, seed in [(1, (16, 16), 3, 1), (5, (40, 24), 6, 3), (33, (31, 17), 12, 2**40 + 5),
                                (70, (24, 24), 20, 9), (300, (24, 20), 8, 13), (600, (32, 16), 8, 77),
                                (1500, (20, 20), 6, 5)]:
        mats, sph, lights = _random_scene(rng, O, n) if n > 1 else _random_scene(rng, O, 2)
        if n == 1:
            sph = sph[:1].copy(); sph["light_id"] = 0; lights = lights[:1].copy()
        cam = pt.Camera(eye, look, up, w, h, 0, 0, 50.0)
        ref_img, ref_hits, ref_rad, ref_rays = O.oracle_render(cam.pod, mats, sph, lights, seed, mb,
                                                               first_iteration=2, n_iterations=2,
                                                               want_paths=True)
        tr = pt.Pathtracer(cam, mb, seed)
        tr.Init(mats, sph, lights)
        tr.ResetCounters()
        buf = render(torch_cuda, tr, 2, 2)
        assert buf.cpu().numpy().tobytes() == ref_img.tobytes(), (n, w, h)
        assert tr.RayCount() == ref_rays
        for k, it in enumerate((2, 3)):
            hits, rad, _, _ = tr.DebugTrace(0, 0, w, h, it)
            assert np.array_equal(hits, ref_hits[k]), (n, it)
            assert rad.tobytes() == ref_rad[k].tobytes(), (n, it)
        tr.Release()


def test_scene_with_more_materials_than_lds_holds(pt, O, torch_cuda):
    """40 spheres but 3000 materials (60 KB): the scene cannot be staged whole in LDS, so it must
    take the streaming path (materials read from device memory) -- every kernel variant and the
    NEE integrator, bit-exact against the oracle."""
    from test_oracle_golden import _random_scene
    rng = np.random.default_rng(7)
    mats4, sph, lights = _random_scene(rng, O, 40)
    mats = np.zeros(3000, O.MATERIAL_DTYPE)
    mats["color"] = rng.uniform(0.2, 0.95, (3000, 3)).astype(np.float32)
    mats["type"] = rng.integers(0, 3, 3000)
    mats["coef"] = np.where(mats["type"] == 2, 1.4, 0.8).astype(np.float32)
    sph["material_id"] = rng.integers(0, 3000, 40)
    cam = pt.Camera([0, 45, 79.5], [0, 35, 0], [0, 1, 0], 48, 32, 0, 0, 50.0)
    ref_img, _, _, ref_rays = O.oracle_render(cam.pod, mats, sph, lights, 11, 8, n_iterations=3)
    for variant in (pt.VARIANT_AUTO, 0, 1, pt.VARIANT_WAVEFRONT):
        tr = pt.Pathtracer(cam, 8, 11)
        tr.Init(mats, sph, lights)
        tr.SetKernelVariant(variant)
        tr.ResetCounters()
        buf = render(torch_cuda, tr, 1, 3)
        assert buf.cpu().numpy().tobytes() == ref_img.tobytes(), variant
        assert tr.RayCount() == ref_rays
        tr.Release()
    tr = pt.Pathtracer(cam, 8, 11)
    tr.Init(mats, sph, lights)
    tr.SetIntegrator(pt.INTEGRATOR_NEE_MIS)
    nee_ref, _, _, nee_rays = O.oracle_render(cam.pod, mats, sph, lights, 11, 8, n_iterations=2, integrator=1)
    tr.ResetCounters()
    buf = render(torch_cuda, tr, 1, 2)
    assert buf.cpu().numpy().tobytes() == nee_ref.tobytes()
    assert tr.RayCount() == nee_rays
    tr.Release()


def test_empty_and_degenerate_scenes(pt, O, torch_cuda):
    """Edge cases against the oracle, every kernel variant: no spheres at all (every path is one
    Intersect call that misses), spheres but no light, a 1x1 image, depth 0, a sphere of radius 0, a
    sphere with a NaN coordinate and one with an infinite radius, materials of unknown type (they
    scatter nothing: scene.h:179,220) with finite and infinite colours."""
    eye, look, up = [0, 45, 79.5], [0, 35, 0], [0, 1, 0]
    demo = pt.build_scene(pt.SCENE_DEMO, 1)
    mats, sph, lights = demo[0], demo[1], demo[2]
    none_s, none_m, none_l = np.zeros(0, O.SPHERE_DTYPE), np.zeros(0, O.MATERIAL_DTYPE), np.zeros(0, O.LIGHT_DTYPE)
    dark = sph.copy(); dark["light_id"] = -1
    weird = sph.copy()
    weird["radius"][6] = 0.0
    weird["pos"][7, 1] = np.nan
    weird = np.concatenate([weird, weird[:1]])
    weird["radius"][-1] = np.inf
    odd = mats.copy()
    odd["type"][int(sph["material_id"][0])] = 7
    odd["type"][int(sph["material_id"][7])] = -1
    odd["color"][int(sph["material_id"][7])] = [np.inf, 0.5, np.nan]
    cases = [  # materials, spheres, lights, (w, h), depth, iterations
        (none_m, none_s, none_l, (20, 12), 8, 5), (mats, dark, none_l, (24, 24), 8, 4), (mats, sph, lights, (1, 1), 8, 16),
        (mats, sph, lights, (33, 18), 0, 6), (mats, weird, lights, (40, 30), 8, 6), (odd, sph, lights, (40, 30), 8, 6),
    ]
    for m, s, l, (w, h), depth, n in cases:
        cam = pt.Camera(eye, look, up, w, h, 0, 0, 60.0)
        ref, _, _, ref_rays = O.oracle_render(cam.pod, m, s, l, 21, depth, n_iterations=n)
        for variant in (pt.VARIANT_AUTO, pt.VARIANT_REGENERATION, pt.VARIANT_CLASSIC, pt.VARIANT_WAVEFRONT,
                        pt.VARIANT_STEALING + 11):
            tr = pt.Pathtracer(cam, depth, 21)
            tr.Init(m, s, l)
            tr.SetKernelVariant(variant)
            tr.ResetCounters()
            buf = render(torch_cuda, tr, 1, n)
            assert buf.cpu().numpy().tobytes() == ref.tobytes(), (len(s), w, h, depth, variant)
            assert tr.RayCount() == ref_rays
            tr.Release()
    assert ref_rays > 0


@pytest.mark.parametrize("variant", SHIPPED_VARIANTS + [100])
def test_debug_trace_sub_rectangle(pt, torch_cuda, variant):
    g = load_golden("demo_64x64_d4_it1")
    tr = make_tracer(pt, g, variant)
    hits, rad, _, _ = tr.DebugTrace(5, 9, 41, 30, 1)
    assert np.array_equal(hits, g["hit_ids"][0, 9:30, 5:41].astype(np.int32))
    assert rad.tobytes() == np.ascontiguousarray(g["path_radiance"][0, 9:30, 5:41]).tobytes()
    tr.Release()


@pytest.mark.parametrize("variant", SHIPPED_VARIANTS)
def test_run_per_iteration_equals_batched_and_host_buffer(pt, torch_cuda, variant):
    """n x pt_run == pt_run_n(n) == pt_run on a host buffer (Pathtracer::Run semantics)."""
    g = load_golden("demo_64x48_d8_it1-4")
    tr = make_tracer(pt, g, variant)
    a = render(torch_cuda, tr, 1, 4, batched=True).cpu().numpy()
    b = render(torch_cuda, tr, 1, 4, batched=False).cpu().numpy()
    assert a.tobytes() == b.tobytes() == g["image"].tobytes()
    host = np.zeros(64 * 48 * 3, np.float32)
    tr.Reset()
    for k in range(4):
        tr.RunHost(host)
        assert tr.GetIteration() == k + 1
    assert host.tobytes() == g["image"].tobytes()
    # Reset(): the next Run overwrites (iteration 1 weights the old content with 0)
    tr.Reset()
    assert tr.GetIteration() == 0
    buf = torch_cuda.full((64 * 48 * 3,), 123.0, device="cuda")
    tr.Run(buf)
    one = render(torch_cuda, tr, 1, 1).cpu().numpy()
    assert buf.cpu().numpy().tobytes() == one.tobytes()
    tr.Release()


@pytest.mark.parametrize("name", golden_names())
def test_wavefront_variant_matches_golden_bit_exact(pt, torch_cuda, name):
    """The path-compaction (one launch per bounce) variant: same bits, same ray count."""
    g = load_golden(name)
    tr = make_tracer(pt, g, pt.VARIANT_WAVEFRONT)
    tr.ResetCounters()
    buf = render(torch_cuda, tr, int(g["first_iteration"]), int(g["n_iterations"]))
    assert tr.RayCount() == int(g["ray_count"])
    assert buf.cpu().numpy().tobytes() == g["image"].tobytes()
    # per-iteration launches and SUM mode go through the same batches
    buf2 = render(torch_cuda, tr, int(g["first_iteration"]), int(g["n_iterations"]), batched=False)
    assert torch_cuda.equal(buf, buf2)
    tr.Release()


def test_wavefront_renders_row_bands(pt, torch_cuda):
    """The per-bounce launches over a linear-scan scene with row bands: the two ranks' frames add up
    to the fixture's image (one iteration, so sum == mean)."""
    g = load_golden("demo_32x32_d0_it1")
    tr = make_tracer(pt, g, pt.VARIANT_WAVEFRONT)
    total = torch_cuda.zeros(32 * 32 * 3, device="cuda")
    for rank in range(2):
        tr.SetRowBands(16, rank, 2)
        tr.Reset()
        buf = torch_cuda.zeros(32 * 32 * 3, device="cuda")
        tr.Run(buf)
        assert tr.LastLaunch() == pt.LAUNCH_WAVEFRONT
        half = buf.cpu().numpy().reshape(32, 32, 3)
        assert not half[16 * (1 - rank):16 * (2 - rank)].any()      # the other rank's band stays untouched
        total += buf
    assert total.cpu().numpy().tobytes() == g["image"].tobytes()
    tr.Release()


def test_kernel_variants_agree(pt, torch_cuda):
    """Path regeneration changes scheduling only: identical bits with and without it."""
    g = load_golden("mixed_64x64_d16_it1-2")
    imgs = []
    for variant in (0, 1, 16, pt.VARIANT_WAVEFRONT):
        tr = make_tracer(pt, g, variant)
        imgs.append(render(torch_cuda, tr, 1, 2).cpu().numpy().tobytes())
        tr.Release()
    assert imgs[0] == imgs[1] == imgs[2] == imgs[3] == g["image"].tobytes()


def test_work_stealing_variant_is_bit_exact(pt, O, torch_cuda):
    """PT_VARIANT_STEALING (lanes take pixel-iterations over from other lanes of their wave)
    against the plain megakernel and the oracle: image bytes and ray counts, with and without the
    camera-ray ring (9 / 71 spheres), image sizes that leave tiles partly outside the image,
    continued renders, the sum buffer, row bands and more stolen iterations than scratch slots
    allow (40 iterations of very uneven pixels)."""
    cases = [  # scene, (w, h), depth, first iteration, iterations
        (pt.SCENE_DEMO, (64, 64), 8, 1, 12), (pt.SCENE_DEMO, (50, 37), 8, 1, 9), (pt.SCENE_DEMO, (50, 37), 4, 3, 24),
        (pt.SCENE_DEMO, (17, 9), 8, 1, 40), (pt.SCENE_MIXED, (48, 40), 16, 1, 8), (pt.SCENE_MIXED, (33, 21), 16, 5, 16),
        (pt.SCENE_DEMO, (64, 64), 8, 1, 4), (pt.SCENE_DEMO, (64, 64), 8, 1, 3),
    ]
    for scene, (w, h), depth, first, n in cases:
        mats, sph, lights, (eye, look, up, fov) = pt.build_scene(scene, 1)
        cam = pt.Camera(eye, look, up, w, h, 0, 0, fov)
        imgs, rays = {}, {}
        # + 1: without the camera-ray ring; + 11, + 13: the launch cut into 2 / 4 iteration ranges
        for variant in (pt.VARIANT_REGENERATION, pt.VARIANT_STEALING, pt.VARIANT_STEALING + 1,
                        pt.VARIANT_STEALING + 11, pt.VARIANT_STEALING + 13, pt.VARIANT_AUTO):
            tr = pt.Pathtracer(cam, depth, 5)
            tr.Init(mats, sph, lights)
            tr.SetKernelVariant(variant)
            buf = torch_cuda.full((w * h * 3,), 0.25, dtype=torch_cuda.float32, device="cuda")
            if first > 1:   # the running mean continues from what the earlier iterations left
                render(torch_cuda, tr, 1, first - 1, buf=buf)
            tr.ResetCounters()
            tr.SetIteration(first - 1)
            tr.RunN(buf, n)
            torch_cuda.cuda.synchronize()
            imgs[variant], rays[variant] = buf.cpu().numpy().tobytes(), tr.RayCount()
            tr.Release()
        for variant in imgs:
            assert imgs[variant] == imgs[pt.VARIANT_REGENERATION], (scene, w, h, variant)
            assert rays[variant] == rays[pt.VARIANT_REGENERATION]
        if first == 1 and w * h <= 64 * 64:
            ref, _, _, ref_rays = O.oracle_render(cam.pod, mats, sph, lights, 5, depth, n_iterations=n)
            assert imgs[pt.VARIANT_STEALING] == ref.tobytes(), (scene, w, h)
            assert rays[pt.VARIANT_STEALING] == ref_rays
    # a streaming scene (600 spheres scanned brute force through LDS tiles; the four waves of a
    # workgroup share the scan's barriers while each steals within itself)
    from test_oracle_golden import _random_scene
    mats, sph, lights = _random_scene(np.random.default_rng(3), O, 600)
    cam = pt.Camera([0, 45, 79.5], [0, 35, 0], [0, 1, 0], 40, 28, 0, 0, 50.0)
    ref, _, _, ref_rays = O.oracle_render(cam.pod, mats, sph, lights, 8, 6, n_iterations=12)
    for variant in (pt.VARIANT_REGENERATION, pt.VARIANT_STEALING, pt.VARIANT_STEALING + 11, pt.VARIANT_AUTO):
        tr = pt.Pathtracer(cam, 6, 8)
        tr.Init(mats, sph, lights)
        tr.SetAccel(pt.ACCEL_BRUTE_FORCE)
        tr.SetKernelVariant(variant)
        tr.ResetCounters()
        buf = render(torch_cuda, tr, 1, 12)
        assert buf.cpu().numpy().tobytes() == ref.tobytes(), variant
        assert tr.RayCount() == ref_rays
        if variant == pt.VARIANT_AUTO:   # the path pool with the scan's barriers shared by four waves
            assert tr.LastLaunch() == pt.LAUNCH_PATH_POOL
        tr.Release()

    # sum buffer and row bands; 48 iterations on a small image: the default cuts the launch into 3 ranges
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO, 1)
    cam = pt.Camera(eye, look, up, 80, 72, 0, 0, fov)
    out = {}
    for variant in (pt.VARIANT_REGENERATION, pt.VARIANT_STEALING, pt.VARIANT_STEALING + 12):
        tr = pt.Pathtracer(cam, 8, 3)
        tr.Init(mats, sph, lights)
        tr.SetKernelVariant(variant)
        tr.SetAccumMode(pt.ACCUM_SUM)
        tr.SetRowBands(16, 1, 3)
        buf = torch_cuda.zeros(80 * 72 * 3, dtype=torch_cuda.float32, device="cuda")
        tr.RunN(buf, 10)
        tr.RunN(buf, 48)
        torch_cuda.cuda.synchronize()
        out[variant] = buf.cpu().numpy().tobytes()
        tr.Release()
    assert out[pt.VARIANT_STEALING] == out[pt.VARIANT_REGENERATION]
    assert out[pt.VARIANT_STEALING + 12] == out[pt.VARIANT_REGENERATION]


def test_work_stealing_on_random_cases(pt, torch_cuda):
    """60 cases of tools/fuzz_steal.py: random frame sizes, iteration counts, depths, scenes, both
    integrators, sum buffers, row bands, BVH and brute force -- the work-stealing kernel (with and
    without iteration ranges) against the plain megakernel, bit for bit."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_steal
    for seed in range(7000, 7060):
        ok, what = fuzz_steal.one_case(seed)
        assert ok, (seed, what)


def test_tile_split_is_bit_exact(pt, torch_cuda):
    """Row bands rendered by different instances into zero-filled buffers sum to the 1-instance image."""
    g = load_golden("demo_64x48_d8_it1-4")
    total = torch_cuda.zeros(64 * 48 * 3, device="cuda")
    for rank in range(3):
        tr = make_tracer(pt, g)
        tr.SetRowBands(16, rank, 3)
        buf = render(torch_cuda, tr, 1, 4)
        img = buf.cpu().numpy().reshape(48, 64, 3)
        for band in range(3):
            rows = img[band * 16:(band + 1) * 16]
            if band % 3 != rank:
                assert not rows.any()
        total += buf
        tr.Release()
    assert total.cpu().numpy().tobytes() == g["image"].tobytes()


def test_sample_range_split_sums_to_the_mean(pt, torch_cuda):
    g = load_golden("demo_64x48_d8_it1-4")
    total = torch_cuda.zeros(64 * 48 * 3, device="cuda")
    rays = 0
    for rank in range(2):
        tr = make_tracer(pt, g)
        tr.SetAccumMode(pt.ACCUM_SUM)
        tr.ResetCounters()
        buf = torch_cuda.full((64 * 48 * 3,), 7.0, device="cuda")   # stale content must be overwritten
        render(torch_cuda, tr, 1 + 2 * rank, 2, buf=buf)
        rays += tr.RayCount()
        total += buf
        tr.Release()
    assert rays == int(g["ray_count"])
    n = 4
    np.testing.assert_allclose(total.cpu().numpy() / n, g["image"].ravel(), rtol=2e-6 * np.sqrt(n), atol=1e-6)


def test_checkpoint_resume_is_bit_exact(pt, torch_cuda, tmp_path):
    """2 iterations, save, new instance, load, 2 more == 4 uninterrupted (golden), in both
    accumulation modes; mismatching instances and damaged files are refused."""
    g = load_golden("demo_64x48_d8_it1-4")
    path = str(tmp_path / "render.ckp")
    for mode in (pt.ACCUM_RUNNING_MEAN, pt.ACCUM_SUM):
        full = make_tracer(pt, g)
        full.SetAccumMode(mode)
        whole = render(torch_cuda, full, 1, 4)
        full.Release()
        a = make_tracer(pt, g)
        a.SetAccumMode(mode)
        buf = render(torch_cuda, a, 1, 2)
        a.SaveCheckpoint(buf, path)
        a.Release()
        b = make_tracer(pt, g)
        b.SetAccumMode(mode)
        buf2 = torch_cuda.full((64 * 48 * 3,), -1.0, device="cuda")
        b.LoadCheckpoint(buf2, path)
        assert b.GetIteration() == 2
        b.RunN(buf2, 2)
        assert b.GetIteration() == 4
        assert torch_cuda.equal(buf2, whole)
        if mode == pt.ACCUM_RUNNING_MEAN:
            assert buf2.cpu().numpy().tobytes() == g["image"].tobytes()
        b.Release()
    other = pt.Pathtracer(g["camera_pod"], int(g["max_bounces"]), int(g["seed"]) + 1)
    other.Init(g["materials"], g["spheres"], g["lights"])
    other.SetAccumMode(pt.ACCUM_SUM)
    with pytest.raises(pt.PtError) as e:
        other.LoadCheckpoint(buf2, path)          # different seed
    assert e.value.code == pt.PT_ERR_STATE and "seed" in str(e.value)
    # the conditions the frame was rendered under are part of the file: another integrator,
    # another scene or another camera is refused as well
    same = make_tracer(pt, g)
    same.SetAccumMode(pt.ACCUM_SUM)
    same.SetIntegrator(pt.INTEGRATOR_NEE_MIS)
    with pytest.raises(pt.PtError) as e:
        same.LoadCheckpoint(buf2, path)
    assert e.value.code == pt.PT_ERR_STATE and "integrator" in str(e.value)
    same.SetIntegrator(pt.INTEGRATOR_PATH)
    moved = g["spheres"].copy()
    moved["pos"][6, 0] += 1.0
    same.Init(g["materials"], moved, g["lights"])
    with pytest.raises(pt.PtError) as e:
        same.LoadCheckpoint(buf2, path)
    assert e.value.code == pt.PT_ERR_STATE and "scene" in str(e.value)
    same.Init(g["materials"], g["spheres"], g["lights"])
    same.SetCamera(pt.Camera([5, 40, 70], g["look_at"], g["up"], 64, 48, 0, 0, float(g["fov"])))
    with pytest.raises(pt.PtError) as e:
        same.LoadCheckpoint(buf2, path)
    assert e.value.code == pt.PT_ERR_STATE and "camera" in str(e.value)
    same.SetCamera(g["camera_pod"])
    same.LoadCheckpoint(buf2, path)
    assert same.GetIteration() == 2
    same.Release()
    assert not __import__("os").path.exists(path + ".tmp")
    blob = open(path, "rb").read()
    open(path, "wb").write(blob[:-5])
    c = make_tracer(pt, g)
    c.SetAccumMode(pt.ACCUM_SUM)
    with pytest.raises(pt.PtError):
        c.LoadCheckpoint(buf2, path)              # truncated
    open(path, "wb").write(b"not a checkpoint" * 8)
    with pytest.raises(pt.PtError):
        c.LoadCheckpoint(buf2, path)
    with pytest.raises(pt.PtError):
        c.LoadCheckpoint(buf2, str(tmp_path / "missing.ckp"))
    assert c.GetIteration() == 0                  # failed loads leave the instance alone
    c.Release(); other.Release()


def test_camera_update_takes_effect_next_run(pt, torch_cuda):
    g = load_golden("demo_64x64_d4_it1")
    tr = make_tracer(pt, g)
    a = render(torch_cuda, tr, 1, 1).cpu().numpy()
    cam2 = pt.Camera([5, 40, 70], g["look_at"], g["up"], 64, 64, 0, 0, float(g["fov"]))
    tr.SetCamera(cam2)
    b = render(torch_cuda, tr, 1, 1).cpu().numpy()
    assert a.tobytes() != b.tobytes()
    tr.SetCamera(g["camera_pod"])
    c = render(torch_cuda, tr, 1, 1).cpu().numpy()
    assert a.tobytes() == c.tobytes() == g["image"].tobytes()
    with pytest.raises(pt.PtError):
        tr.SetCamera(pt.Camera(g["eye"], g["look_at"], g["up"], 32, 64))
    tr.Release()


def test_errors(pt, torch_cuda):
    g = load_golden("demo_32x32_d0_it1")
    tr = pt.Pathtracer(g["camera_pod"], 4, 1)
    buf = torch_cuda.zeros(32 * 32 * 3, device="cuda")
    with pytest.raises(pt.PtError) as e:
        tr.Run(buf)                      # Run before Init
    assert e.value.code == pt.PT_ERR_STATE
    bad = g["spheres"].copy()
    bad["material_id"][3] = 6
    with pytest.raises(pt.PtError) as e:
        tr.Init(g["materials"], bad, g["lights"])
    assert e.value.code == pt.PT_ERR_INVALID and "material" in str(e.value)
    bad = g["spheres"].copy()
    bad["light_id"][3] = 1
    with pytest.raises(pt.PtError):
        tr.Init(g["materials"], bad, g["lights"])
    with pytest.raises(pt.PtError):
        tr.SetRowBands(10, 0, 2)         # not a multiple of 16
    with pytest.raises(pt.PtError):
        tr.SetRowBands(16, 2, 2)
    with pytest.raises(pt.PtError):
        tr.SetAccumMode(5)
    tr.Init(g["materials"], g["spheres"], g["lights"])
    with pytest.raises(pt.PtError):
        tr.DebugTrace(0, 0, 33, 32, 1)
    with pytest.raises(pt.PtError):
        tr.DebugTrace(0, 0, 32, 32, 0)
    with pytest.raises(ValueError):
        tr.Run(torch_cuda.zeros(10, device="cuda"))
    tr.Release()
    with pytest.raises(pt.PtError):
        pt.Pathtracer(g["camera_pod"], -1, 1)


def test_full_size_properties_1080p(pt, O, torch_cuda):
    """The headline benchmark configuration (demo scene, 1080p) through size-independent properties: determinism, iteration
    composition (2+2 == 4, bit-exact), ray count == per-path Intersect calls, and a sampled
    band of rows against the oracle."""
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO)
    cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
    tr = pt.Pathtracer(cam, 8, 1)
    tr.Init(mats, sph, lights)
    tr.ResetCounters()
    a = render(torch_cuda, tr, 1, 4)
    rays4 = tr.RayCount()
    b = render(torch_cuda, tr, 1, 4)
    assert torch_cuda.equal(a, b)
    assert tr.RayCount() == 2 * rays4
    c = torch_cuda.zeros_like(a)
    tr.Reset(); tr.RunN(c, 2); tr.RunN(c, 2)
    assert torch_cuda.equal(a, c)
    assert 3.0 < rays4 / (1920 * 1080 * 4 * 4) < 4.5     # ~3.7 Intersect calls per path
    assert bool(torch_cuda.isfinite(a).all())
    # rows 536..543 against the oracle (all 1080 rows are rendered, not only 1072)
    img = a.cpu().numpy().reshape(1080, 1920, 3)
    ref, _, _, _ = O.oracle_render(cam.pod, mats, sph, lights, 1, 8, n_iterations=4,
                                   rect=(0, 536, 1920, 544), n_threads=8)
    assert img[536:544].tobytes() == ref[536:544].tobytes()
    ref, _, _, _ = O.oracle_render(cam.pod, mats, sph, lights, 1, 8, n_iterations=4,
                                   rect=(0, 1072, 1920, 1080), n_threads=8)
    assert img[1072:].tobytes() == ref[1072:].tobytes()
    tr.Release()


def _oracle_bands(O, cam, mats, sph, lights, depth, n, bands):
    return {y0: O.oracle_render(cam.pod, mats, sph, lights, 1, depth, n_iterations=n,
                                rect=(0, y0, cam.Width(), y0 + rows), n_threads=8)[0][y0:y0 + rows]
            for (y0, rows) in bands}


def test_config2_as_benchmarked_1080p_64_iterations(pt, O, torch_cuda):
    """The headline configuration exactly as bench.py times it: demo scene, 1920x1080, depth 8,
    64 iterations in ONE pt_run_n with the default variant -- the path pool, two batches of 32
    iterations on two buffer sets and two streams.  The same through the work-stealing megakernel:
    8160 tiles x 3 iteration ranges, about 24 000 workgroups for some 1 500 resident slots, so the
    ticket order, the per-tile hand-over of the frame buffer between workgroups (possibly on
    different XCDs) and the scratch-block pool all work under real oversubscription.  Against the
    plain megakernel (image bytes, ray count), a forced 4-range launch, and oracle row bands at the
    top, middle and bottom of the frame."""
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO)
    cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
    imgs, rays = {}, {}
    expect = {pt.VARIANT_AUTO: pt.LAUNCH_PATH_POOL, pt.VARIANT_REGENERATION: pt.LAUNCH_MEGAKERNEL,
              pt.VARIANT_STEALING: pt.LAUNCH_MEGAKERNEL_STEALING, pt.VARIANT_STEALING + 13: pt.LAUNCH_MEGAKERNEL_STEALING}
    for variant in expect:
        tr = pt.Pathtracer(cam, 8, 1)
        tr.Init(mats, sph, lights)
        tr.SetKernelVariant(variant)
        tr.ResetCounters()
        imgs[variant] = render(torch_cuda, tr, 1, 64).cpu().numpy().reshape(1080, 1920, 3)
        rays[variant] = tr.RayCount()
        assert tr.LastLaunch() == expect[variant], variant
        tr.Release()
    ref = imgs[pt.VARIANT_REGENERATION]
    for variant in imgs:
        assert imgs[variant].tobytes() == ref.tobytes(), variant
        assert rays[variant] == rays[pt.VARIANT_REGENERATION]
    for y0, band in _oracle_bands(O, cam, mats, sph, lights, 8, 64, [(0, 2), (539, 2), (1078, 2)]).items():
        assert imgs[pt.VARIANT_AUTO][y0:y0 + 2].tobytes() == band.tobytes(), y0


def test_config5_as_benchmarked_1080p_depth16_32_iterations(pt, O, torch_cuda):
    """The mixed-material configuration as bench.py --workload mixed16 times it (71 spheres, depth
    16 with Russian roulette, 32 iterations, default variant) against the plain megakernel and
    oracle row bands."""
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_MIXED)
    cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
    imgs, rays = {}, {}
    for variant in (pt.VARIANT_AUTO, pt.VARIANT_REGENERATION):
        tr = pt.Pathtracer(cam, 16, 1)
        tr.Init(mats, sph, lights)
        tr.SetKernelVariant(variant)
        tr.ResetCounters()
        imgs[variant] = render(torch_cuda, tr, 1, 32).cpu().numpy().reshape(1080, 1920, 3)
        rays[variant] = tr.RayCount()
        tr.Release()
    assert imgs[pt.VARIANT_AUTO].tobytes() == imgs[pt.VARIANT_REGENERATION].tobytes()
    assert rays[pt.VARIANT_AUTO] == rays[pt.VARIANT_REGENERATION]
    for y0, band in _oracle_bands(O, cam, mats, sph, lights, 16, 32, [(300, 1), (700, 2)]).items():
        assert imgs[pt.VARIANT_AUTO][y0:y0 + band.shape[0]].tobytes() == band.tobytes(), y0


def test_benchmark_sampling_matches_injected_reference_image(pt, torch_cuda):
    """256x256, depth 8, 64 iterations (the benchmark's depth and sampling): the default kernels'
    image is the reference host build's, bit for bit (fixture: oracle/gen_golden.py t3)."""
    g = load_golden("t3_libm_demo_256x256_d8_it1-64")
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO)
    cam = pt.Camera(eye, look, up, 256, 256, 0, 0, fov)
    for variant in SHIPPED_VARIANTS + [pt.VARIANT_WAVEFRONT]:
        tr = pt.Pathtracer(cam, 8, 1)
        tr.Init(mats, sph, lights)
        tr.SetKernelVariant(variant)
        tr.ResetCounters()
        img = render(torch_cuda, tr, 1, 64).cpu().numpy()
        assert tr.RayCount() == int(g["ray_count_injected"]), variant
        assert sha(img) == str(g["image_injected_sha256"]), variant
        tr.Release()


def test_config3_size_4k_tile_split_over_8_ranks(pt, torch_cuda):
    """The 4K benchmark configuration's frame (3840x2160) split into 16-row bands over 8 ranks: the sum of
    the ranks' buffers is the 1-rank image bit for bit, and the ray counts add up."""
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO)
    cam = pt.Camera(eye, look, up, 3840, 2160, 0, 0, fov)
    one = pt.Pathtracer(cam, 8, 1)
    one.Init(mats, sph, lights)
    one.ResetCounters()
    full = render(torch_cuda, one, 1, 2)
    rays_full = one.RayCount()
    one.Release()
    total = torch_cuda.zeros_like(full)
    rays = 0
    for rank in range(8):
        tr = pt.Pathtracer(cam, 8, 1)
        tr.Init(mats, sph, lights)
        tr.SetRowBands(16, rank, 8)
        tr.ResetCounters()
        total += render(torch_cuda, tr, 1, 2)
        rays += tr.RayCount()
        tr.Release()
    assert rays == rays_full
    assert torch_cuda.equal(total, full)


def test_config3_as_benchmarked_4k_128_iterations(pt, O, torch_cuda):
    """One GPU's share of the 8-GPU configuration exactly as bench.py --workload demo4k times it:
    demo scene, 3840x2160, depth 8, 128 iterations in ONE pt_run_n with the default variant -- the
    path pool, 33 M paths per iteration, batches of a few iterations alternating between the two
    buffer sets and streams some thirty times.  Two row bands (the top of the frame and one across
    the mirror and glass balls) against the oracle, bit for bit; then the same share as rank 3 of 8
    renders it under the sample-range split (iterations 385..512 into a sum buffer) against the
    oracle's sum of those iterations."""
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO)
    cam = pt.Camera(eye, look, up, 3840, 2160, 0, 0, fov)
    tr = pt.Pathtracer(cam, 8, 1)
    tr.Init(mats, sph, lights)
    img = render(torch_cuda, tr, 1, 128).cpu().numpy().reshape(2160, 3840, 3)
    assert tr.LastLaunch() == pt.LAUNCH_PATH_POOL
    for y0, band in _oracle_bands(O, cam, mats, sph, lights, 8, 128, [(0, 1), (1500, 1)]).items():
        assert img[y0:y0 + 1].tobytes() == band.tobytes(), y0
    tr.SetAccumMode(pt.ACCUM_SUM)
    img = render(torch_cuda, tr, 385, 128).cpu().numpy().reshape(2160, 3840, 3)
    want = O.oracle_render(cam.pod, mats, sph, lights, 1, 8, first_iteration=385, n_iterations=128, rect=(0, 1501, 3840, 1502),
                           accum_mode=O.ACCUM_SUM, n_threads=8)[0][1501:1502]
    assert img[1501:1502].tobytes() == want.tobytes()
    tr.Release()


def test_config4_size_synth10k_1080p_rows_against_oracle(pt, O, torch_cuda):
    """The 10 007-sphere benchmark configuration at 1080p, one iteration with the default variant
    (sphere BVH, per-bounce wavefront launches with dynamic ray fetch) against an explicit
    PT_VARIANT_WAVEFRONT instance; four rows spread over the frame are compared with the oracle
    (a linear scan) bit for bit."""
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_SYNTH10K, 1)
    cam = pt.Camera(eye, look, up, 1920, 1080, 0, 0, fov)
    tr = pt.Pathtracer(cam, 8, 1)
    tr.Init(mats, sph, lights)
    img = render(torch_cuda, tr, 1, 1).cpu().numpy().reshape(1080, 1920, 3)
    wf = make_wavefront_image(pt, torch_cuda, cam, mats, sph, lights, 8)
    assert img.tobytes() == wf.tobytes()
    for y in (0, 337, 702, 1079):
        ref, _, _, _ = O.oracle_render(cam.pod, mats, sph, lights, 1, 8, rect=(0, y, 1920, y + 1), n_threads=8)
        assert img[y].tobytes() == ref[y].tobytes(), y
    tr.Release()


def make_wavefront_image(pt, torch, cam, mats, sph, lights, depth):
    tr = pt.Pathtracer(cam, depth, 1)
    tr.Init(mats, sph, lights)
    tr.SetKernelVariant(pt.VARIANT_WAVEFRONT)
    img = render(torch, tr, 1, 1).cpu().numpy().reshape(cam.Height(), cam.Width(), 3)
    tr.Release()
    return img


def test_variants_tuning_and_memory_footprint(pt, torch_cuda):
    """pt_set_kernel_variant takes the five named variants only, experiments go through
    pt_debug_set_tuning; pt_memory_footprint predicts what a render allocates; pt_set_memory_limit
    bounds the wavefront queues (PT_VARIANT_AUTO then renders the same image with the megakernel)."""
    lib = pt.load_library()
    g = load_golden("demo_64x48_d8_it1-4")
    tr = make_tracer(pt, g, pt.VARIANT_AUTO)
    for bad in (7, 64, 101, 300, 501, 513, -2):
        assert lib.pt_set_kernel_variant(tr._h, bad) == pt.PT_ERR_INVALID
    for key, value in ((pt.TUNE_REGEN_MIN_IDLE, 0), (pt.TUNE_REGEN_MIN_IDLE, 65), (pt.TUNE_STEAL_UNIT, 2), (pt.TUNE_CAMERA_RAY_RING, 2),
                       (pt.TUNE_ITERATION_RANGES, 33), (pt.TUNE_WAVEFRONT_REFILL, 65), (99, 0)):
        assert lib.pt_debug_set_tuning(tr._h, key, value) == pt.PT_ERR_INVALID
    tr.SetKernelVariant(pt.VARIANT_STEALING)
    tr.SetTuning(pt.TUNE_ITERATION_RANGES, 2)
    tr.SetTuning(pt.TUNE_STEAL_UNIT, 4)
    img = render(torch_cuda, tr, 1, 4).cpu().numpy()
    assert img.tobytes() == g["image"].tobytes()
    tr.Release()

    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO)
    cam = pt.Camera(eye, look, up, 640, 360, 0, 0, fov)
    tr = pt.Pathtracer(cam, 8, 1)
    tr.Init(mats, sph, lights)
    # PT_VARIANT_AUTO on a scene that fits LDS: the path pool, 12 bytes of radiance per path and a
    # mask byte per pixel-iteration (a quarter of a byte per path, planned as one)
    ppi = 640 * 360 * 4
    now, nxt = tr.MemoryFootprint(8)
    assert now < 4096 and 12 * ppi * 8 + ppi * 8 // 4 <= nxt - now <= 12 * ppi * 8 + ppi * 8 // 4 + 2**20
    buf = torch_cuda.zeros(640 * 360 * 3, device="cuda")
    tr.RunN(buf, 8)
    assert tr.LastLaunch() == pt.LAUNCH_PATH_POOL
    assert tr.MemoryFootprint(8) == (nxt, nxt)
    ref8 = buf.cpu().numpy().tobytes()
    tr.Release()
    tr = pt.Pathtracer(cam, 8, 1)
    tr.Init(mats, sph, lights)
    tr.SetMemoryLimit(13 * ppi * 2 + 2**20)                        # two sets of one iteration each
    now2, nxt2 = tr.MemoryFootprint(8)
    assert 12 * ppi * 2 <= nxt2 - now2 <= 13 * ppi * 2 + 2**20
    tr.RunN(buf, 8)
    assert tr.LastLaunch() == pt.LAUNCH_PATH_POOL and buf.cpu().numpy().tobytes() == ref8
    tr.Reset()
    tr.SetMemoryLimit(12 * ppi - 1)                                # not one iteration's radiance: the megakernel,
    assert tr.MemoryFootprint(8)[1] == nxt2                        # without its 48 MiB scratch pool either
    tr.RunN(buf, 8)
    assert tr.LastLaunch() == pt.LAUNCH_MEGAKERNEL and buf.cpu().numpy().tobytes() == ref8
    tr.Release()
    # the work-stealing megakernel on request
    tr = pt.Pathtracer(cam, 8, 1)
    tr.Init(mats, sph, lights)
    tr.SetKernelVariant(pt.VARIANT_STEALING)
    now, nxt = tr.MemoryFootprint(64)
    pool = 48 * 2**20 + 1024
    tiles = 40 * 23
    assert now < 4096 and nxt == now + pool + 4 * (1 + tiles)
    assert tr.MemoryFootprint(4) == (now, now + pool)              # a short launch is not cut into ranges
    tr.RunN(buf, 64)
    assert tr.LastLaunch() == pt.LAUNCH_MEGAKERNEL_STEALING
    assert tr.MemoryFootprint(64) == (nxt, nxt)
    # the footprint does not grow with the frame: a 4K instance predicts the same pool
    cam4k = pt.Camera(eye, look, up, 3840, 2160, 0, 0, fov)
    big = pt.Pathtracer(cam4k, 8, 1)
    big.Init(mats, sph, lights)
    big.SetKernelVariant(pt.VARIANT_STEALING)
    b_now, b_next = big.MemoryFootprint(16)
    assert b_next - b_now == pool
    big.SetMemoryLimit(2**20)                                      # no room for the pool: no stealing, nothing allocated
    assert big.MemoryFootprint(16) == (b_now, b_now)
    big.Release()
    tr.Release()

    # BVH scene: PT_VARIANT_AUTO takes the persistent BVH waves -- the camera rays in a launch of
    # their own, then one launch that follows the survivors to their end: one queue of 44-byte
    # records + 12 bytes of radiance per path of a batch
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_SYNTH10K, 1)
    cam = pt.Camera(eye, look, up, 256, 128, 0, 0, fov)
    tr = pt.Pathtracer(cam, 8, 1)
    tr.Init(mats, sph, lights)
    now, nxt = tr.MemoryFootprint(2)
    ppi = 256 * 128 * 4
    paths = ppi * 2
    assert 56 * paths <= nxt - now <= 56 * paths + 2**20
    ref = render(torch_cuda, tr, 1, 2).cpu().numpy().tobytes()
    assert tr.LastLaunch() == pt.LAUNCH_WAVEFRONT_WHOLE_PATHS
    assert tr.MemoryFootprint(2) == (nxt, nxt)
    tr.SetTuning(pt.TUNE_WAVEFRONT_WHOLE_PATHS, 0)                 # one launch per bounce: two queues, 100 bytes
    now, nxt = tr.MemoryFootprint(2)
    assert 44 * paths <= nxt - now <= 44 * paths + 2**20           # the second queue
    assert render(torch_cuda, tr, 1, 2).cpu().numpy().tobytes() == ref
    assert tr.LastLaunch() == pt.LAUNCH_WAVEFRONT
    assert tr.MemoryFootprint(2) == (nxt, nxt)
    tr.Release()
    tr = pt.Pathtracer(cam, 8, 1)
    tr.Init(mats, sph, lights)
    tr.SetMemoryLimit(56 * ppi - 1)                                # no room for one iteration's queue: whole paths from
    now, nxt = tr.MemoryFootprint(2)                               # the camera ray on, 12 bytes per path, one batch
    assert 12 * paths <= nxt - now <= 12 * paths + 2**20
    assert render(torch_cuda, tr, 1, 2).cpu().numpy().tobytes() == ref
    assert tr.MemoryFootprint(2) == (nxt, nxt)
    tr.Release()
    tr = pt.Pathtracer(cam, 8, 1)
    tr.Init(mats, sph, lights)
    tr.SetMemoryLimit(12 * 256 * 128 * 4 - 1)                      # nor one iteration's radiance
    now, nxt = tr.MemoryFootprint(2)
    assert nxt == now                                              # below 48 MiB: no scratch pool either
    assert render(torch_cuda, tr, 1, 2).cpu().numpy().tobytes() == ref     # rendered by the BVH megakernel
    assert tr.MemoryFootprint(2)[0] == now
    tr.SetKernelVariant(pt.VARIANT_WAVEFRONT)
    with pytest.raises(pt.PtError) as e:
        tr.RunN(torch_cuda.zeros(256 * 128 * 3, device="cuda"), 1)
    assert e.value.code == pt.PT_ERR_NOMEM
    tr.SetMemoryLimit(56 * 256 * 128 * 4 + 2**20)                  # one iteration per batch
    tr.SetKernelVariant(pt.VARIANT_AUTO)
    assert render(torch_cuda, tr, 1, 2).cpu().numpy().tobytes() == ref
    # room for two iterations: two buffer sets of one iteration each, batches alternating between
    # them on two streams, folded into the frame in iteration order
    tr.SetMemoryLimit(0)
    ref5 = render(torch_cuda, tr, 1, 5).cpu().numpy().tobytes()
    tr.Release()
    tr = pt.Pathtracer(cam, 8, 1)
    tr.Init(mats, sph, lights)
    tr.SetMemoryLimit(2 * 56 * 256 * 128 * 4 + 2**20)
    now, nxt = tr.MemoryFootprint(5)
    assert 2 * 56 * 256 * 128 * 4 <= nxt - now <= 2 * 56 * 256 * 128 * 4 + 2**20
    assert render(torch_cuda, tr, 1, 5).cpu().numpy().tobytes() == ref5
    assert tr.MemoryFootprint(5) == (nxt, nxt)
    tr.SetTuning(pt.TUNE_WAVEFRONT_BATCH, 1)                       # 2^20 paths in flight: eight iterations, four per set
    assert render(torch_cuda, tr, 1, 5).cpu().numpy().tobytes() == ref5
    rays = []
    for mode in (1, 2, 0):                                         # every path to its end in one launch; from the
        tr.SetTuning(pt.TUNE_WAVEFRONT_WHOLE_PATHS, mode)          # first bounce on; one launch per bounce
        tr.ResetCounters()
        assert render(torch_cuda, tr, 1, 5).cpu().numpy().tobytes() == ref5
        rays.append(tr.RayCount())
    assert rays[0] == rays[1] == rays[2]
    tr.Release()
