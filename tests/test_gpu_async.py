"""pt_run_async / pt_wait: the reference's frame loop (Display calls Run once per iteration,
main.cpp:205-232) without the wait between the calls.  Consecutive calls alternate between the
library's two buffer sets and streams; the frame buffer must come out as from blocking calls, bit
for bit (golden vectors of the reference's host build), and every other entry point called in the
middle of a series must see the calls made so far."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    torch.cuda.set_device(0)
    return torch


def tracer(pt, g, variant=-1):
    tr = pt.Pathtracer(g["camera_pod"], int(g["max_bounces"]), int(g["seed"]))
    tr.Init(g["materials"], g["spheres"], g["lights"])
    tr.SetKernelVariant(variant)
    return tr


@pytest.mark.parametrize("name", ["demo_64x48_d8_it1-4", "mixed_64x64_d16_it1-2", "demo_256x256_d4_it1"])
@pytest.mark.parametrize("variant", [-1, 500])
def test_async_calls_match_golden(pt, torch_cuda, name, variant):
    g = load_golden(name)
    first, n = int(g["first_iteration"]), int(g["n_iterations"])
    tr = tracer(pt, g, variant)
    buf = torch_cuda.full((tr.height * tr.width * 3,), 7.0, dtype=torch_cuda.float32, device="cuda")
    tr.ResetCounters()
    tr.SetIteration(first - 1)
    for _ in range(n):
        tr.RunAsync(buf)            # one iteration per call, no wait
    tr.Wait()
    assert tr.GetIteration() == first + n - 1
    assert buf.cpu().numpy().tobytes() == g["image"].tobytes()
    assert tr.RayCount() == int(g["ray_count"])
    tr.Release()


def test_async_series_of_64_calls_equals_one_launch(pt, torch_cuda):
    """64 calls + one wait at 480x270 against pt_run_n(64); then a series whose calls differ in
    size, interrupted by entry points that have to order themselves behind it."""
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO, 1)
    cam = pt.Camera(eye, look, up, 480, 270, 0, 0, fov)
    tr = pt.Pathtracer(cam, 8, 3)
    tr.Init(mats, sph, lights)
    want = torch_cuda.zeros(480 * 270 * 3, device="cuda")
    tr.ResetCounters()
    tr.RunN(want, 64)
    rays = tr.RayCount()
    got = torch_cuda.full_like(want, -1.0)
    tr.Reset()
    tr.ResetCounters()
    for _ in range(64):
        tr.RunAsync(got)
    tr.Wait()
    assert got.cpu().numpy().tobytes() == want.cpu().numpy().tobytes()
    assert tr.RayCount() == rays and tr.GetIteration() == 64
    assert tr.LastLaunch() == pt.LAUNCH_PATH_POOL

    # calls of 1, 2, 3, ... iterations (the buffer sets grow on the way), a ray count and an image read in
    # the middle (both must see the series so far), a blocking Run, then more asynchronous calls
    mid = torch_cuda.zeros_like(want)
    tr.Reset()
    tr.RunN(mid, 6)
    mid_rays = None
    tr.Reset()
    tr.ResetCounters()
    got.fill_(5.0)
    done = 0
    for k in (1, 2, 3):
        tr.RunAsync(got, k)
        done += k
    snapshot = np.ascontiguousarray(tr.ReadImage(got), np.float32)   # no Wait: the entry point joins the series itself
    assert snapshot.tobytes() == mid.cpu().numpy().tobytes()
    mid_rays = tr.RayCount()
    tr.RunAsync(got, 10)
    tr.Run(got)                                       # blocking, one iteration
    for k in (20, 1, 1, 25):
        tr.RunAsync(got, k)
    tr.Wait()
    assert tr.GetIteration() == 64
    assert got.cpu().numpy().tobytes() == want.cpu().numpy().tobytes()
    assert tr.RayCount() == rays and 0 < mid_rays < rays
    tr.Release()


def test_async_on_a_bvh_scene_and_with_row_bands(pt, torch_cuda):
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_SYNTH10K, 1)
    cam = pt.Camera(eye, look, up, 160, 96, 0, 0, fov)
    tr = pt.Pathtracer(cam, 8, 1)
    tr.Init(mats, sph, lights)
    assert tr.AccelNodes() > 0
    want = torch_cuda.zeros(160 * 96 * 3, device="cuda")
    tr.RunN(want, 6)
    got = torch_cuda.zeros_like(want)
    tr.Reset()
    for _ in range(6):
        tr.RunAsync(got)
    tr.Wait()
    assert got.cpu().numpy().tobytes() == want.cpu().numpy().tobytes()
    # row bands of a sharded render (rank 1 of 3), sum buffer
    tr.SetAccumMode(pt.ACCUM_SUM)
    tr.SetRowBands(16, 1, 3)
    want.zero_(); got.zero_()
    tr.Reset()
    tr.RunN(want, 5)
    tr.Reset()
    for k in (2, 1, 2):
        tr.RunAsync(got, k)
    tr.Wait()
    assert got.cpu().numpy().tobytes() == want.cpu().numpy().tobytes()
    tr.Release()


def test_async_under_a_memory_limit_and_with_the_nee_integrator(pt, torch_cuda):
    """Calls that do not fit half the batch buffers, and renders that do not use them at all, are
    plain enqueues on the handle's stream: same image."""
    g = load_golden("demo_64x48_d8_it1-4")
    tr = tracer(pt, g)
    tr.SetMemoryLimit(13 * 64 * 48 * 4 + 2**16)      # room for one iteration in ONE set only
    buf = torch_cuda.zeros(64 * 48 * 3, device="cuda")
    for _ in range(4):
        tr.RunAsync(buf)
    tr.Wait()
    assert buf.cpu().numpy().tobytes() == g["image"].tobytes()
    tr.SetMemoryLimit(0)
    tr.SetIntegrator(pt.INTEGRATOR_NEE_MIS)
    want = torch_cuda.zeros_like(buf)
    tr.Reset()
    tr.RunN(want, 4)
    tr.Reset()
    for _ in range(4):
        tr.RunAsync(buf)
    tr.Wait()
    assert buf.cpu().numpy().tobytes() == want.cpu().numpy().tobytes()
    tr.Release()
