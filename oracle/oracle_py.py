"""TEST INFRASTRUCTURE ONLY -- ctypes access to the CPU oracle (libpt_oracle.so) and, in the
development container, to the host build of the reference's own trace logic
(oracle/_ref/libpt_ref*.so).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product never does.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB = os.path.join(_HERE, "libpt_oracle.so")
REF_LIB = os.path.join(_HERE, "_ref", "libpt_ref.so")
REF_LIBM_LIB = os.path.join(_HERE, "_ref", "libpt_ref_libm.so")

# Same layouts as include/ptmi.h (declared again so that the oracle does not import the product).
CAMERA_DTYPE = np.dtype([("pos", "<f4", 3), ("px_x", "<f4", 3), ("px_y", "<f4", 3),
                         ("first_ray_dir", "<f4", 3), ("width", "<u4"), ("height", "<u4")])
MATERIAL_DTYPE = np.dtype([("color", "<f4", 3), ("coef", "<f4"), ("type", "<i4")])
SPHERE_DTYPE = np.dtype([("pos", "<f4", 3), ("radius", "<f4"), ("material_id", "<u4"),
                         ("light_id", "<i4")])
LIGHT_DTYPE = np.dtype([("type", "<i4"), ("u_bits", "<u4", 3), ("intensity", "<f4", 3)])
TRIANGLE_DTYPE = np.dtype([("v0", "<f4", 3), ("v1", "<f4", 3), ("v2", "<f4", 3), ("material_id", "<u4"),
                           ("light_id", "<i4")])

ACCUM_RUNNING_MEAN, ACCUM_SUM = 0, 1
HIT_MISS, HIT_NOT_TRACED = -1, -2

_vp, _u32, _u64, _i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int
_pu64 = ctypes.POINTER(_u64)
_libs = {}


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def load_oracle() -> ctypes.CDLL:
    if "oracle" not in _libs:
        if not os.path.exists(ORACLE_LIB):
            raise ImportError(f"{ORACLE_LIB} not built: run `make -C oracle`")
        lib = ctypes.CDLL(ORACLE_LIB)
        lib.ptor_render.restype = _i32
        lib.ptor_render.argtypes = [_vp, _vp, _u32, _vp, _u32, _vp, _u32, _u64, _i32, _u32, _u32,
                                    _u32, _u32, _u32, _u32, _i32, _vp, _vp, _vp, _pu64, _i32]
        lib.ptor_camera_init.restype = None
        lib.ptor_camera_init.argtypes = [_vp, _vp, _vp, _vp, _u32, _u32, ctypes.c_float]
        lib.ptor_philox.restype = None
        lib.ptor_philox.argtypes = [_vp, _vp, _vp]
        lib.ptor_uniforms.restype = None
        lib.ptor_uniforms.argtypes = [_u64, _u32, _u32, _u32, _u32, _vp]
        lib.ptor_sincos.restype = None
        lib.ptor_sincos.argtypes = [ctypes.c_float, ctypes.POINTER(ctypes.c_float),
                                    ctypes.POINTER(ctypes.c_float)]
        _libs["oracle"] = lib
    return _libs["oracle"]


def have_ref() -> bool:
    return os.path.exists(REF_LIB) and os.path.exists(REF_LIBM_LIB)


def load_ref(libm: bool = False) -> ctypes.CDLL:
    key = "ref_libm" if libm else "ref"
    if key not in _libs:
        path = REF_LIBM_LIB if libm else REF_LIB
        if not os.path.exists(path):
            raise ImportError(f"{path} not built: run `make -C oracle ref` (needs /root/reference)")
        lib = ctypes.CDLL(path)
        lib.ref_render.restype = _i32
        lib.ref_render.argtypes = [_vp, _vp, _vp, _u32, _u32, ctypes.c_float, _vp, _u32, _vp, _u32,
                                   _vp, _u32, _u64, _i32, _u32, _u32, _u32, _u32, _u32, _u32, _i32,
                                   _vp, _vp, _vp, _pu64]
        lib.ref_camera_init.restype = None
        lib.ref_camera_init.argtypes = [_vp, _vp, _vp, _vp, _u32, _u32, ctypes.c_float]
        lib.ref_sizeof.restype = _i32
        lib.ref_sizeof.argtypes = [_i32]
        lib.ref_uses_libm_sincos.restype = _i32
        _libs[key] = lib
    return _libs[key]


def camera_init(eye, look_at, up, width, height, fov, lib=None, fn="ptor_camera_init") -> np.ndarray:
    lib = lib or load_oracle()
    pod = np.zeros(1, CAMERA_DTYPE)
    e, l, u = (np.ascontiguousarray(v, np.float32) for v in (eye, look_at, up))
    getattr(lib, fn)(_ptr(pod), _ptr(e), _ptr(l), _ptr(u), width, height, fov)
    return pod


def _outputs(width, height, rect, n_iterations, max_bounces, buffer, want_paths):
    x0, y0, x1, y1 = rect
    if buffer is None:
        buffer = np.zeros((height, width, 3), np.float32)
    hits = rad = None
    if want_paths:
        hits = np.empty((n_iterations, y1 - y0, x1 - x0, 4, max_bounces + 1), np.int32)
        rad = np.empty((n_iterations, y1 - y0, x1 - x0, 4, 3), np.float32)
    return buffer, hits, rad


INTEGRATOR_PATH, INTEGRATOR_NEE_MIS = 0, 1


def oracle_render(cam_pod, mats, prims, lights, seed, max_bounces, first_iteration=1, n_iterations=1,
                  rect=None, accum_mode=ACCUM_RUNNING_MEAN, buffer=None, want_paths=False, n_threads=1,
                  integrator=INTEGRATOR_PATH, triangles=None):
    """Returns (buffer[h,w,3], hit_ids or None, path_radiance or None, ray_count).
    triangles (TRIANGLE_DTYPE, optional): this build's extension, primitives behind the spheres."""
    lib = load_oracle()
    if triangles is not None and len(triangles):
        return _oracle_render_tri(lib, cam_pod, mats, prims, triangles, lights, seed, max_bounces, first_iteration,
                                  n_iterations, rect, accum_mode, buffer, want_paths, n_threads, integrator)
    lib.ptor_render_with.restype = _i32
    lib.ptor_render_with.argtypes = [_i32] + list(lib.ptor_render.argtypes)
    cam_pod = np.ascontiguousarray(cam_pod, CAMERA_DTYPE)
    width, height = int(cam_pod["width"].reshape(-1)[0]), int(cam_pod["height"].reshape(-1)[0])
    rect = rect or (0, 0, width, height)
    mats = np.ascontiguousarray(mats, MATERIAL_DTYPE)
    prims = np.ascontiguousarray(prims, SPHERE_DTYPE)
    lights = np.ascontiguousarray(lights, LIGHT_DTYPE)
    buffer, hits, rad = _outputs(width, height, rect, n_iterations, max_bounces, buffer, want_paths)
    cnt = _u64(0)
    rc = lib.ptor_render_with(integrator, _ptr(cam_pod), _ptr(mats), len(mats), _ptr(prims), len(prims),
                              _ptr(lights), len(lights), seed, max_bounces, first_iteration, n_iterations,
                              *rect, accum_mode, _ptr(buffer), _ptr(hits), _ptr(rad), ctypes.byref(cnt),
                              n_threads)
    if rc != 0:
        raise ValueError("ptor_render: invalid arguments")
    return buffer, hits, rad, int(cnt.value)


def _oracle_render_tri(lib, cam_pod, mats, prims, triangles, lights, seed, max_bounces, first_iteration, n_iterations,
                       rect, accum_mode, buffer, want_paths, n_threads, integrator):
    lib.ptor_render_tri.restype = _i32
    cam_pod = np.ascontiguousarray(cam_pod, CAMERA_DTYPE)
    width, height = int(cam_pod["width"].reshape(-1)[0]), int(cam_pod["height"].reshape(-1)[0])
    rect = rect or (0, 0, width, height)
    mats = np.ascontiguousarray(mats, MATERIAL_DTYPE)
    prims = np.ascontiguousarray(prims, SPHERE_DTYPE)
    tris = np.ascontiguousarray(triangles, TRIANGLE_DTYPE)
    lights = np.ascontiguousarray(lights, LIGHT_DTYPE)
    buffer, hits, rad = _outputs(width, height, rect, n_iterations, max_bounces, buffer, want_paths)
    cnt = _u64(0)
    u32 = ctypes.c_uint32
    args = [_i32(integrator), _ptr(cam_pod), _ptr(mats), u32(len(mats)), _ptr(prims), u32(len(prims)), _ptr(tris), u32(len(tris)),
            _ptr(lights), u32(len(lights)), _u64(seed), _i32(max_bounces), u32(first_iteration), u32(n_iterations),
            *[u32(v) for v in rect], _i32(accum_mode), _ptr(buffer), _ptr(hits), _ptr(rad), ctypes.byref(cnt), _i32(n_threads)]
    if lib.ptor_render_tri(*args) != 0:
        raise ValueError("ptor_render_tri: invalid arguments")
    return buffer, hits, rad, int(cnt.value)


def triangle_intersect(tri, o, d, tmin=0.03, tmax=3.402823466e+38):
    """Oracle: one ray against one triangle (TRIANGLE_DTYPE record); t or -1."""
    lib = load_oracle()
    lib.ptor_triangle_intersect.restype = ctypes.c_float
    t = np.ascontiguousarray(tri, TRIANGLE_DTYPE).reshape(1)
    o = np.ascontiguousarray(o, np.float32)
    d = np.ascontiguousarray(d, np.float32)
    return float(lib.ptor_triangle_intersect(_ptr(t), _ptr(o), _ptr(d), ctypes.c_float(tmin), ctypes.c_float(tmax)))


def ref_render(eye, look_at, up, width, height, fov, mats, prims, lights, seed, max_bounces,
               first_iteration=1, n_iterations=1, rect=None, accum_mode=ACCUM_RUNNING_MEAN,
               buffer=None, want_paths=False, libm=False):
    """The reference's own Trace()/GenerateRay()/Camera() compiled for host; same outputs."""
    lib = load_ref(libm)
    rect = rect or (0, 0, width, height)
    e, l, u = (np.ascontiguousarray(v, np.float32) for v in (eye, look_at, up))
    mats = np.ascontiguousarray(mats, MATERIAL_DTYPE)
    prims = np.ascontiguousarray(prims, SPHERE_DTYPE)
    lights = np.ascontiguousarray(lights, LIGHT_DTYPE)
    buffer, hits, rad = _outputs(width, height, rect, n_iterations, max_bounces, buffer, want_paths)
    cnt = _u64(0)
    rc = lib.ref_render(_ptr(e), _ptr(l), _ptr(u), width, height, fov, _ptr(mats), len(mats),
                        _ptr(prims), len(prims), _ptr(lights), len(lights), seed, max_bounces,
                        first_iteration, n_iterations, *rect, accum_mode, _ptr(buffer), _ptr(hits),
                        _ptr(rad), ctypes.byref(cnt))
    if rc != 0:
        raise ValueError("ref_render: invalid arguments")
    return buffer, hits, rad, int(cnt.value)


# ---- interactive camera (camera.h:79-129) ---------------------------------------------------------
CAMERA_STATE_DTYPE = np.dtype([("view", "<f4", (3, 3)), ("dist", "<f4"), ("hfov", "<f4"), ("vfov", "<f4"),
                               ("cam", CAMERA_DTYPE)])


def camera_moves(eye, look_at, up, width, height, fov, ops):
    """Oracle: camera PODs before and after each move; ops = [(kind, x, y, z)], kind 0 Translate,
    1 Rotate.  Returns an array of len(ops) + 1 CAMERA_DTYPE records."""
    lib = load_oracle()
    for name in ("ptor_camera_state_init", "ptor_camera_translate", "ptor_camera_rotate"):
        getattr(lib, name).restype = None
    lib.ptor_camera_state_init.argtypes = [_vp, _vp, _vp, _vp, _u32, _u32, ctypes.c_float]
    lib.ptor_camera_translate.argtypes = [_vp, _vp]
    lib.ptor_camera_rotate.argtypes = [_vp, _vp]
    st = np.zeros(1, CAMERA_STATE_DTYPE)
    e, l, u = (np.ascontiguousarray(v, np.float32) for v in (eye, look_at, up))
    lib.ptor_camera_state_init(_ptr(st), _ptr(e), _ptr(l), _ptr(u), width, height, fov)
    pods = [st["cam"].copy()]
    for kind, x, y, z in ops:
        v = np.array([x, y, z], np.float32)
        (lib.ptor_camera_rotate if kind else lib.ptor_camera_translate)(_ptr(st), _ptr(v))
        pods.append(st["cam"].copy())
    return np.concatenate(pods)


def ref_camera_moves(eye, look_at, up, width, height, fov, ops):
    """The reference's own Camera::Translate / Rotate (libm build of the harness)."""
    lib = load_ref(libm=True)
    lib.ref_camera_moves.restype = None
    lib.ref_camera_moves.argtypes = [_vp, _vp, _vp, _u32, _u32, ctypes.c_float, _vp, _u32, _vp]
    e, l, u = (np.ascontiguousarray(v, np.float32) for v in (eye, look_at, up))
    o = np.ascontiguousarray(ops, np.float32).reshape(-1, 4)
    pods = np.zeros(len(o) + 1, CAMERA_DTYPE)
    lib.ref_camera_moves(_ptr(e), _ptr(l), _ptr(u), width, height, fov, _ptr(o), len(o), _ptr(pods))
    return pods


# ---- direct-light sampling components (the reference's dormant code) -------------------------------
NEE_OPS = {name: i for i, name in enumerate([
    "SAMPLE_SPHERE", "SAMPLE_CONE", "CONE_PDF", "POWER_HEURISTIC", "COORD_SYSTEM", "MATERIAL", "LIGHT_L",
    "SEGMENT", "SPHERE_SAMPLE_AREA", "SPHERE_SAMPLE_FROM", "LIGHT_PDF", "INTERSECT_P", "SAMPLE_LIGHT"])}


def _nee_args(op, inputs, scene):
    x = np.zeros((len(inputs), 16), np.float32)
    a = np.asarray(inputs, np.float32)
    x[:, :a.shape[1]] = a
    mats, prims, lights = scene if scene is not None else (np.zeros(0, MATERIAL_DTYPE), np.zeros(0, SPHERE_DTYPE),
                                                           np.zeros(0, LIGHT_DTYPE))
    mats = np.ascontiguousarray(mats, MATERIAL_DTYPE)
    prims = np.ascontiguousarray(prims, SPHERE_DTYPE)
    lights = np.ascontiguousarray(lights, LIGHT_DTYPE)
    return NEE_OPS[op] if isinstance(op, str) else int(op), x, mats, prims, lights, np.zeros((len(x), 16), np.float32)


def nee_eval(op, inputs, scene=None, as_written=False):
    """Oracle: rows of `inputs` through one of the direct-light sampling functions -> [n,16]."""
    lib = load_oracle()
    lib.ptor_nee_eval.restype = _i32
    lib.ptor_nee_eval.argtypes = [_i32, _i32, _vp, _u32, _vp, _u32, _vp, _u32, _vp, _vp, _u32]
    op, x, mats, prims, lights, out = _nee_args(op, inputs, scene)
    if lib.ptor_nee_eval(op, int(as_written), _ptr(mats), len(mats), _ptr(prims), len(prims), _ptr(lights),
                         len(lights), _ptr(x), _ptr(out), len(x)) != 0:
        raise ValueError("ptor_nee_eval: invalid arguments")
    return out


def ref_nee_eval(op, inputs, scene=None):
    """The reference's own functions (host build, deterministic sin/cos injected) -> [n,16]."""
    lib = load_ref()
    lib.ref_nee_eval.restype = _i32
    lib.ref_nee_eval.argtypes = [_i32, _vp, _u32, _vp, _u32, _vp, _u32, _vp, _vp, _u32]
    op, x, mats, prims, lights, out = _nee_args(op, inputs, scene)
    if lib.ref_nee_eval(op, _ptr(mats), len(mats), _ptr(prims), len(prims), _ptr(lights), len(lights),
                        _ptr(x), _ptr(out), len(x)) != 0:
        raise ValueError("ref_nee_eval: invalid arguments")
    return out
