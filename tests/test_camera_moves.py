"""Interactive camera (Camera::Translate / Rotate, camera.h:79-129): the product's host code and
the oracle against the reference's own methods (golden fixture from the reference's host build).

Rotate calls libm's sinf/cosf on the host in the reference too, so product and oracle must agree
bit for bit on the machine the test runs on; against the fixture, which was made on another CPU,
the comparison allows the last bits of libm to differ (rtol 1e-6)."""
import numpy as np

from conftest import load_golden

FIELDS = ("pos", "px_x", "px_y", "first_ray_dir")


def _product_moves(pt, g):
    cam = pt.Camera(g["eye"], g["look_at"], g["up"], int(g["width"]), int(g["height"]), 0.0, 0.0, float(g["fov"]))
    assert cam.IsUpdated() and not cam.IsUpdated()
    pods = [cam.pod.copy()]
    for kind, x, y, z in g["ops"]:
        (cam.Rotate if kind else cam.Translate)([x, y, z])
        assert cam.IsUpdated()
        pods.append(cam.pod.copy())
    return np.concatenate(pods)


def test_product_and_oracle_agree_bitwise(pt, O):
    g = load_golden("camera_moves")
    mine = _product_moves(pt, g)
    oracle = O.camera_moves(g["eye"], g["look_at"], g["up"], int(g["width"]), int(g["height"]), float(g["fov"]),
                            g["ops"])
    assert mine.tobytes() == oracle.tobytes()


def test_moves_match_reference_methods(pt, O):
    g = load_golden("camera_moves")
    mine = _product_moves(pt, g)
    ref = g["pods"]
    assert len(mine) == len(ref) == len(g["ops"]) + 1
    assert mine[0].tobytes() == ref[0].tobytes()            # constructor: no libm sin/cos involved
    for f in FIELDS:
        np.testing.assert_allclose(mine[f], ref[f], rtol=1e-6, atol=1e-6)
    # translation leaves the image-plane vectors untouched (camera.h:87-89)
    assert mine["px_x"][1].tobytes() == mine["px_x"][0].tobytes()
    assert mine["pos"][1].tolist() != mine["pos"][0].tolist()
    if O.have_ref():
        live = O.ref_camera_moves(g["eye"], g["look_at"], g["up"], int(g["width"]), int(g["height"]),
                                  float(g["fov"]), g["ops"])
        assert mine.tobytes() == live.tobytes()              # same machine, same libm: bit for bit


def test_moved_camera_feeds_the_tracer(pt):
    """pt_create / pt_set_camera take the pod of a moved camera (layout check, no GPU needed)."""
    cam = pt.Camera([0, 45, 79.5], [0, 35, 0], [0, 1, 0], 64, 48)
    cam.Rotate([0.05, 0.0, 0.0])
    cam.Translate([1.0, 0.0, 0.0])
    assert cam.pod.dtype == pt.CAMERA_DTYPE and cam.pod.shape == (1,)
    assert (cam.Width(), cam.Height()) == (64, 48)
