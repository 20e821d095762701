"""T1 of the parity ladder: the plain-C oracle against the golden vectors produced by the
reference's own headers + Trace() compiled for host (oracle/gen_golden.py), bit for bit.

When the host build of the reference is present (development container only) the oracle is
additionally compared with it live on inputs the fixtures do not hold.
"""
import hashlib

import numpy as np
import pytest

from conftest import golden_names, load_golden


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_golden(O, name):
    g = load_golden(name)
    img, hits, rad, rays = O.oracle_render(g["camera_pod"], g["materials"], g["spheres"], g["lights"],
                                           int(g["seed"]), int(g["max_bounces"]),
                                           int(g["first_iteration"]), int(g["n_iterations"]),
                                           want_paths=True)
    assert rays == int(g["ray_count"])
    assert sha(hits) == str(g["hit_ids_sha256"])          # bit-exact hit-index sequences
    assert sha(rad) == str(g["path_radiance_sha256"])     # bit-exact per-path radiance
    assert img.tobytes() == g["image"].tobytes()          # bit-exact running mean
    if "hit_ids" in g.files:
        assert np.array_equal(hits, g["hit_ids"].astype(np.int32))
        assert rad.tobytes() == g["path_radiance"].tobytes()


def test_oracle_camera_matches_golden(O):
    g = load_golden("cameras")
    for (w, h), pod in zip(g["sizes"], g["pods"]):
        mine = O.camera_init(g["eye"], g["look_at"], g["up"], int(w), int(h), float(g["fov"]))
        assert mine.tobytes() == pod.tobytes(), (w, h)
    mine = O.camera_init(g["eye2"], g["look_at2"], g["up2"], 320, 200, float(g["fov2"]))
    assert mine.tobytes() == g["pod2"].tobytes()


def test_struct_sizes_recorded(O):
    # sizeof of the reference's device-visible structs on x86-64 (Vec, Point, Ray, Color, BBox,
    # Primitive, Material, Light, Camera, Scene, Intersection)
    g = load_golden("cameras")
    assert list(g["sizeof"]) == [16, 12, 28, 12, 24, 48, 20, 28, 156, 48, 48]


def test_threads_do_not_change_results(O):
    g = load_golden("demo_64x48_d8_it1-4")
    args = (g["camera_pod"], g["materials"], g["spheres"], g["lights"], int(g["seed"]), int(g["max_bounces"]))
    a = O.oracle_render(*args, n_iterations=4, n_threads=1)
    b = O.oracle_render(*args, n_iterations=4, n_threads=5)
    assert a[0].tobytes() == b[0].tobytes() and a[3] == b[3]
    assert a[0].tobytes() == g["image"].tobytes()


def test_iteration_ranges_compose(O):
    """Rendering 1..2 then 3..4 into the same buffer equals rendering 1..4 (running mean)."""
    g = load_golden("demo_64x48_d8_it1-4")
    args = (g["camera_pod"], g["materials"], g["spheres"], g["lights"], int(g["seed"]), int(g["max_bounces"]))
    buf, _, _, r1 = O.oracle_render(*args, first_iteration=1, n_iterations=2)
    buf, _, _, r2 = O.oracle_render(*args, first_iteration=3, n_iterations=2, buffer=buf)
    assert buf.tobytes() == g["image"].tobytes()
    assert r1 + r2 == int(g["ray_count"])


def test_sum_mode_matches_mean_within_tolerance(O):
    g = load_golden("demo_64x48_d8_it1-4")
    args = (g["camera_pod"], g["materials"], g["spheres"], g["lights"], int(g["seed"]), int(g["max_bounces"]))
    s, _, _, _ = O.oracle_render(*args, n_iterations=4, accum_mode=O.ACCUM_SUM)
    np.testing.assert_allclose(s / 4.0, g["image"], rtol=1e-6, atol=1e-6)


def test_invalid_arguments_rejected(O):
    g = load_golden("demo_32x32_d0_it1")
    with pytest.raises(ValueError):
        O.oracle_render(g["camera_pod"], g["materials"], g["spheres"], g["lights"], 1, 4, first_iteration=0)
    with pytest.raises(ValueError):
        O.oracle_render(g["camera_pod"], g["materials"], g["spheres"], g["lights"], 1, 4, rect=(0, 0, 33, 32))
    bad = g["spheres"].copy()
    bad["material_id"][0] = 99
    with pytest.raises(ValueError):
        O.oracle_render(g["camera_pod"], g["materials"], bad, g["lights"], 1, 4)


def _random_scene(rng, O, n):
    mats = np.zeros(4, O.MATERIAL_DTYPE)
    mats["color"] = rng.uniform(0.2, 0.95, (4, 3)).astype(np.float32)
    mats["type"] = [0, 1, 2, 0]
    mats["coef"] = [0.0, 0.9, 1.33, 0.0]
    sph = np.zeros(n, O.SPHERE_DTYPE)
    sph["pos"] = rng.uniform(-30, 30, (n, 3)).astype(np.float32) + np.array([0, 35, -20], np.float32)
    sph["radius"] = rng.uniform(2, 9, n).astype(np.float32)
    sph["material_id"] = rng.integers(0, 4, n)
    sph["light_id"] = -1
    sph["light_id"][0] = 0
    sph["light_id"][n // 2] = 1
    lights = np.zeros(2, O.LIGHT_DTYPE)
    lights["type"] = 1
    lights["u_bits"][0, 0] = 0
    lights["u_bits"][1, 0] = n // 2
    lights["intensity"] = [[9, 8, 7], [3, 6, 12]]
    return mats, sph, lights


def test_oracle_matches_reference_live_on_random_scenes(O):
    if not O.have_ref():
        pytest.skip("host build of the reference is only present in the development container")
    rng = np.random.default_rng(42)
    eye, look, up = np.array([0, 45, 79.5], np.float32), np.array([0, 35, 0], np.float32), np.array([0, 1, 0], np.float32)
    for n, (w, h), mb, seed in [(5, (40, 24), 6, 3), (33, (31, 17), 12, 2**40 + 5), (70, (24, 24), 20, 9)]:
        mats, sph, lights = _random_scene(rng, O, n)
        cam = O.camera_init(eye, look, up, w, h, 50.0)
        a = O.oracle_render(cam, mats, sph, lights, seed, mb, first_iteration=2, n_iterations=2, want_paths=True)
        b = O.ref_render(eye, look, up, w, h, 50.0, mats, sph, lights, seed, mb, first_iteration=2,
                         n_iterations=2, want_paths=True)
        assert a[3] == b[3]
        assert np.array_equal(a[1], b[1])
        assert a[2].tobytes() == b[2].tobytes()
        assert a[0].tobytes() == b[0].tobytes()
