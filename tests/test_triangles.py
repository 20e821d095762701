"""The triangle primitive on the CPU side (no device): this build's extension -- the reference has spheres
only ("will be changed to triangle later", primitive.h:25-28), so there is nothing of the reference's
to pin it to (PARITY UNPINNED); what is checked here is the oracle against closed-form answers and its
own invariants.  The GPU side is compared with this oracle in test_gpu_triangles.py."""
import numpy as np
import pytest

from triangle_scenes import box_room, triangle_soup


def one(O, v0, v1, v2):
    t = np.zeros(1, O.TRIANGLE_DTYPE)
    t["v0"], t["v1"], t["v2"] = v0, v1, v2
    return t


def test_triangle_intersection_known_answers(O):
    t = one(O, [0, 0, 0], [1, 0, 0], [0, 1, 0])                      # in the plane z = 0
    assert O.triangle_intersect(t, [0.25, 0.25, 2], [0, 0, -1]) == 2.0
    assert O.triangle_intersect(t, [0.25, 0.25, -3], [0, 0, 1]) == 3.0          # both sides
    assert O.triangle_intersect(t, [0.25, 0.25, 2], [0, 0, 1]) == -1.0          # behind the ray
    assert O.triangle_intersect(t, [0.75, 0.75, 2], [0, 0, -1]) == -1.0         # u + v > 1
    assert O.triangle_intersect(t, [-0.01, 0.5, 2], [0, 0, -1]) == -1.0         # u < 0
    assert O.triangle_intersect(t, [0.5, 0.5, 2], [0, 0, -1]) == 2.0            # on the hypotenuse: u + v == 1 counts
    assert O.triangle_intersect(t, [0, 0, 2], [0, 0, -1]) == 2.0                # at a vertex
    assert O.triangle_intersect(t, [0.25, 0.25, 0.02], [0, 0, -1]) == -1.0      # t <= EPSILON = 0.03 (globals.h:49)
    assert O.triangle_intersect(t, [0.25, 0.25, 2], [0, 0, -1], tmax=2.0) == -1.0   # strictly below tmax
    assert O.triangle_intersect(t, [0.25, 0.25, 2], [1, 0, 0]) == -1.0          # parallel: det == 0
    # oblique: hit point (0.2, 0.3, 0) from (1.2, 1.3, 1) along (-1, -1, -1) / sqrt(3)
    d = np.float32(-1 / np.sqrt(3))
    assert abs(O.triangle_intersect(t, [1.2, 1.3, 1], [d, d, d]) - np.sqrt(3)) < 1e-6
    # degenerate triangles never hit, whatever the ray
    rng = np.random.default_rng(1)
    for deg in (one(O, [1, 2, 3], [1, 2, 3], [4, 5, 6]), one(O, [1, 2, 3], [1, 2, 3], [1, 2, 3]),
                one(O, [0, 0, 0], [1, 1, 1], [2, 2, 2])):
        for _ in range(200):
            o, dd = rng.normal(size=3) * 5, rng.normal(size=3)
            assert O.triangle_intersect(deg, o, dd / np.linalg.norm(dd)) == -1.0
    # NaN anywhere: no hit
    assert O.triangle_intersect(t, [np.nan, 0.25, 2], [0, 0, -1]) == -1.0
    assert O.triangle_intersect(one(O, [0, 0, 0], [np.nan, 0, 0], [0, 1, 0]), [0.25, 0.25, 2], [0, 0, -1]) == -1.0


def test_box_room_of_triangles_behaves_like_a_room(pt, O):
    mats, sph, tris, lights, (eye, look, up, fov) = box_room(pt)
    cam = pt.Camera(eye, look, up, 48, 32, 0, 0, fov).pod
    img, hits, rad, rays = O.oracle_render(cam, mats, sph, lights, 3, 8, n_iterations=2, want_paths=True, triangles=tris)
    n_prims = len(sph) + len(tris)
    assert (hits[:, :, :, :, 0] >= 0).all() and (hits < n_prims).all()      # closed room: no camera ray escapes
    traced = hits[hits != O.HIT_NOT_TRACED]
    # later rays can leave through a corner: a hit closer than EPSILON = 0.03 does not count (primitive.h:43-44,
    # globals.h:49), so a ray that starts within 0.03 of the next wall passes through it -- rarely
    assert (traced >= 0).mean() > 0.995
    assert rays == traced.size
    assert np.isfinite(img).all() and 0.05 < img.mean() < 5.0                  # lit by the quad light
    # the same walls as one quad per wall split the other way: another scene, same room -- the image agrees
    # to Monte-Carlo noise only, but the set of primitives the camera sees first must be walls and spheres
    first = hits[0, :, :, :, 0]
    assert set(np.unique(first)) <= set(range(n_prims)) and (first < len(sph)).any() and (first >= len(sph)).any()
    # spheres only through the triangle entry point == the plain entry point, bit for bit
    g_mats, g_sph, g_lights, _ = pt.build_scene(pt.SCENE_DEMO, 1)
    a = O.oracle_render(cam, g_mats, g_sph, g_lights, 3, 4)
    b = O.oracle_render(cam, g_mats, g_sph, g_lights, 3, 4, triangles=np.zeros(0, O.TRIANGLE_DTYPE))
    assert a[0].tobytes() == b[0].tobytes() and a[3] == b[3]


def test_duplicate_triangles_keep_the_lower_index(pt, O):
    rng = np.random.default_rng(5)
    mats, sph, tris, lights, (eye, look, up, fov) = triangle_soup(pt, rng, 40, 0)
    cam = pt.Camera(eye, look, up, 64, 48, 0, 0, fov).pod
    _, hits, _, _ = O.oracle_render(cam, mats, sph, lights, 1, 4, want_paths=True, triangles=tris)
    assert (hits != 4).all()            # triangle 4 is a copy of triangle 3: strict '<' never lets it win (scene.h:78-80)
    assert (hits == 3).any()
    assert not np.isin(hits, [1, 2, 5]).any()   # the degenerate ones are never hit


def test_oracle_refuses_what_triangles_do_not_support(pt, O):
    mats, sph, tris, lights, _ = box_room(pt)
    cam = pt.Camera([0, 45, 79.5], [0, 35, 0], [0, 1, 0], 8, 8, 0, 0, 60.0).pod
    with pytest.raises(ValueError):
        O.oracle_render(cam, mats, sph, lights, 1, 4, triangles=tris, integrator=O.INTEGRATOR_NEE_MIS)
    bad = tris.copy(); bad["material_id"][0] = 99
    with pytest.raises(ValueError):
        O.oracle_render(cam, mats, sph, lights, 1, 4, triangles=bad)
