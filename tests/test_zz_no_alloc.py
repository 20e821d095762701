"""pt_run_n_noalloc -- the entry point that allocates no device memory (its scratch is a statically
allocated device array) -- against pt_run_n through the C ABI: bit-exact frame buffers and ray
counts on two calls in a row, the second with a smaller input, so that what the first call left in
the shared scratch (occupancy bits, tickets, stolen results) must not leak into the second."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    torch.cuda.set_device(0)
    return torch


def _render(pt, torch, scene, w, h, depth, n, noalloc, integrator=0, variant=None, sum_mode=False):
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(scene, 1)
    cam = pt.Camera(eye, look, up, w, h, 0, 0, fov)
    tr = pt.Pathtracer(cam, depth, 9)
    tr.Init(mats, sph, lights)
    tr.SetIntegrator(integrator)
    if variant is not None:
        tr.SetKernelVariant(variant)
    if sum_mode:
        tr.SetAccumMode(pt.ACCUM_SUM)
    buf = torch.full((w * h * 3,), 0.75, dtype=torch.float32, device="cuda")
    tr.ResetCounters()
    run = tr.RunNNoAlloc if noalloc else tr.RunN
    run(buf, n)
    run(buf, 3)              # a continued render: short launch, single paths change hands
    torch.cuda.synchronize()
    out = buf.cpu().numpy().tobytes(), tr.RayCount(), tr.GetIteration()
    tr.Release()
    return out


def test_noalloc_equals_run_n_two_calls_in_a_row(pt, torch_cuda):
    # first call: 960x540 x 64 iterations -- 2040 tiles cut into iteration ranges (tickets in use),
    # every scratch block of the pool in use; second call: a smaller frame, fewer iterations
    for scene, (w, h), depth, n in [(pt.SCENE_DEMO, (960, 540), 8, 64), (pt.SCENE_DEMO, (200, 120), 8, 20)]:
        want = _render(pt, torch_cuda, scene, w, h, depth, n, noalloc=False)
        got = _render(pt, torch_cuda, scene, w, h, depth, n, noalloc=True)
        assert got[1] == want[1] and got[2] == want[2] == n + 3
        assert got[0] == want[0], (w, h)


def test_noalloc_leaves_device_memory_alone(pt, torch_cuda):
    """Free device memory does not move across calls, including the first chunked one (where
    pt_run_n allocates its tickets and its scratch pool)."""
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO, 1)
    cam = pt.Camera(eye, look, up, 640, 360, 0, 0, fov)
    tr = pt.Pathtracer(cam, 8, 1)
    tr.Init(mats, sph, lights)
    buf = torch_cuda.zeros(640 * 360 * 3, dtype=torch_cuda.float32, device="cuda")
    torch_cuda.cuda.synchronize()
    free0 = torch_cuda.cuda.mem_get_info()[0]
    tr.RunNNoAlloc(buf, 1)
    tr.RunNNoAlloc(buf, 64)
    torch_cuda.cuda.synchronize()
    assert torch_cuda.cuda.mem_get_info()[0] == free0
    tr.RunN(buf, 64)       # the allocating entry point, for contrast: pool + tickets
    torch_cuda.cuda.synchronize()
    assert torch_cuda.cuda.mem_get_info()[0] < free0
    tr.Release()


def test_noalloc_other_kernels_and_modes(pt, torch_cuda):
    cases = [  # scene, size, depth, iterations, integrator, variant, sum buffer
        (pt.SCENE_MIXED, (150, 90), 16, 17, 0, None, False),        # 71 spheres, no camera-ray ring
        (pt.SCENE_DEMO, (90, 50), 8, 6, 1, None, True),             # NEE + MIS, sum buffer
        (pt.SCENE_SYNTH10K, (64, 40), 8, 2, 0, None, False),        # BVH scene: the BVH megakernel, not the wavefront
        (pt.SCENE_DEMO, (33, 17), 4, 40, 0, pt.VARIANT_REGENERATION, False),   # no stealing at all
    ]
    for scene, (w, h), depth, n, integrator, variant, sum_mode in cases:
        want = _render(pt, torch_cuda, scene, w, h, depth, n, False, integrator, variant, sum_mode)
        got = _render(pt, torch_cuda, scene, w, h, depth, n, True, integrator, variant, sum_mode)
        assert got == want, (scene, w, h)


def test_noalloc_limits(pt, torch_cuda):
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO, 1)
    # 16 400 x 1 024 pixels = 1025 x 64 = 65 600 tiles: one more than the static tickets cover
    cam = pt.Camera(eye, look, up, 16400, 1024, 0, 0, fov)
    tr = pt.Pathtracer(cam, 2, 1)
    tr.Init(mats, sph, lights)
    buf = torch_cuda.zeros(16400 * 1024 * 3, dtype=torch_cuda.float32, device="cuda")
    with pytest.raises(pt.PtError) as e:
        tr.RunNNoAlloc(buf, 1)
    assert e.value.code == pt.PT_ERR_INVALID and "tiles" in str(e.value)
    assert tr.GetIteration() == 0
    tr.Release()
    del buf
    cam = pt.Camera(eye, look, up, 64, 64, 0, 0, fov)
    tr = pt.Pathtracer(cam, 2, 1)
    buf = torch_cuda.zeros(64 * 64 * 3, dtype=torch_cuda.float32, device="cuda")
    with pytest.raises(pt.PtError) as e:
        tr.RunNNoAlloc(buf, 1)                    # before Init
    assert e.value.code == pt.PT_ERR_STATE
    tr.Init(mats, sph, lights)
    tr.SetKernelVariant(pt.VARIANT_WAVEFRONT)
    with pytest.raises(pt.PtError) as e:
        tr.RunNNoAlloc(buf, 1)
    assert e.value.code == pt.PT_ERR_STATE
    tr.Release()
