"""The C-ABI surface of libptmi.so and its host-only helpers (no GPU needed)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ptmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pt):
    lib = pt.load_library()
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"libptmi.so does not export {name}"
    # and the Python binding table covers the header
    assert sorted(n for n, _, _ in pt.ABI) == names


def test_version(pt):
    assert pt.load_library().pt_version() >= 100


def test_pod_sizes_match_reference_fields(pt):
    # Material 20 B and Light 28 B are the reference's own sizeof; the sphere drops the 24-byte
    # bounding box of the 48-byte Primitive
    g = load_golden("cameras")
    assert pt.MATERIAL_DTYPE.itemsize == g["sizeof"][6]
    assert pt.LIGHT_DTYPE.itemsize == g["sizeof"][7]
    assert pt.SPHERE_DTYPE.itemsize == g["sizeof"][5] - g["sizeof"][4]


def test_camera_matches_reference_constructor(pt):
    g = load_golden("cameras")
    for (w, h), pod in zip(g["sizes"], g["pods"]):
        cam = pt.Camera(g["eye"], g["look_at"], g["up"], int(w), int(h), 0.0, 0.0, float(g["fov"]))
        assert cam.pod.tobytes() == pod.tobytes(), (w, h)
        assert (cam.Width(), cam.Height()) == (w, h)
    cam = pt.Camera(g["eye2"], g["look_at2"], g["up2"], 320, 200, 0.0, 0.0, float(g["fov2"]))
    assert cam.pod.tobytes() == g["pod2"].tobytes()


def test_camera_rejects_bad_arguments(pt):
    with pytest.raises(pt.PtError):
        pt.Camera([0, 0, 0], [0, 0, -1], [0, 1, 0], 0, 10)


def test_demo_scene_is_the_reference_scene(pt):
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO)
    g = load_golden("demo_64x64_d4_it1")   # scene arrays as stored when the fixtures were made
    assert mats.tobytes() == g["materials"].tobytes()
    assert sph.tobytes() == g["spheres"].tobytes()
    assert lights.tobytes() == g["lights"].tobytes()
    # InitScene, main.cpp:129-164
    assert len(mats) == 6 and len(sph) == 9 and len(lights) == 1
    assert list(eye) == [0, 45, 79.5] and list(look) == [0, 35, 0] and list(up) == [0, 1, 0] and fov == 60
    assert list(mats["type"]) == [0, 0, 0, 0, 1, 2]
    assert sph["pos"][0].tolist() == [100050.0, 40.0, 0.0] and sph["radius"][0] == 1e5
    assert sph["pos"][8].tolist() == [0.0, np.float32(579.6), -40.0] and sph["radius"][8] == 500
    assert list(sph["material_id"]) == [2, 1, 3, 3, 3, 3, 4, 5, 1]
    assert list(sph["light_id"]) == [-1] * 8 + [0]
    assert lights["type"][0] == pt.AREA_LIGHT and lights["u_bits"][0, 0] == 8
    assert lights["intensity"][0].tolist() == [12, 12, 12]


def test_procedural_scenes(pt):
    mats, sph, lights, _ = pt.build_scene(pt.SCENE_SYNTH10K, 1)
    assert len(sph) == 10007 and len(lights) == 1
    assert sph["light_id"][6] == 0 and lights["u_bits"][0, 0] == 6      # walls 0..5, light 6
    assert sph["material_id"].max() < len(mats)
    frac = np.bincount(mats["type"][sph["material_id"][7:]], minlength=3) / 10000.0
    assert abs(frac[0] - 0.70) < 0.03 and abs(frac[1] - 0.15) < 0.02 and abs(frac[2] - 0.15) < 0.02
    again = pt.build_scene(pt.SCENE_SYNTH10K, 1)
    assert again[1].tobytes() == sph.tobytes()
    other = pt.build_scene(pt.SCENE_SYNTH10K, 2)
    assert other[1].tobytes() != sph.tobytes()
    g = load_golden("synth10k_32x32_d8_it1")
    assert sph.tobytes() == g["spheres"].tobytes() and mats.tobytes() == g["materials"].tobytes()

    mats, sph, lights, _ = pt.build_scene(pt.SCENE_MIXED)
    assert len(sph) == 6 + 1 + 64
    assert set(mats["type"][sph["material_id"][7:]]) == {0, 1, 2}
    with pytest.raises(pt.PtError):
        pt.build_scene(77)


def test_scene_build_capacity_check(pt):
    lib = pt.load_library()
    nm, ns, nl = ctypes.c_uint32(1), ctypes.c_uint32(1), ctypes.c_uint32(1)
    mats, sph, lights = np.zeros(1, pt.MATERIAL_DTYPE), np.zeros(1, pt.SPHERE_DTYPE), np.zeros(1, pt.LIGHT_DTYPE)
    rc = lib.pt_scene_build(pt.SCENE_DEMO, 1, mats.ctypes.data, ctypes.byref(nm), sph.ctypes.data,
                            ctypes.byref(ns), lights.ctypes.data, ctypes.byref(nl), None, None, None, None)
    assert rc == pt.PT_ERR_INVALID


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


def test_rendering_without_a_gpu_fails_loudly(pt):
    """No CPU fallback: without a device pt_create reports PT_ERR_DEVICE and says why."""
    if not _no_gpu():
        pytest.skip("a GPU is present")
    cam = pt.Camera([0, 45, 79.5], [0, 35, 0], [0, 1, 0], 32, 32)
    with pytest.raises(pt.PtError) as e:
        pt.Pathtracer(cam, 4)
    assert e.value.code == pt.PT_ERR_DEVICE
    assert "no CPU path" in str(e.value) or "gfx950" in str(e.value)


def test_null_handles_are_rejected(pt):
    lib = pt.load_library()
    assert lib.pt_destroy(None) == pt.PT_ERR_INVALID
    assert lib.pt_reset(None) == pt.PT_ERR_INVALID
    assert lib.pt_run(None, None, 1) == pt.PT_ERR_INVALID
    assert lib.pt_iteration(None) == 0
    assert lib.pt_create(None, None, 4, 1) == pt.PT_ERR_INVALID
    h = ctypes.c_void_p()
    bad = np.zeros(1, pt.CAMERA_DTYPE)   # width == height == 0
    assert lib.pt_create(ctypes.byref(h), bad.ctypes.data, 4, 1) == pt.PT_ERR_INVALID
    assert b"camera" in lib.pt_last_error(None)
