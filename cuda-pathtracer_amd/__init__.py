"""Python host-side mirror of the reference's ``Pathtracer`` interface over libptmi's C ABI.

The reference's host code is C++ (``pathtracer.h:38-65``, ``main.cpp:128-190``); the C++ mirror
lives in ``host/pathtracer.hpp``.  This module is the same thin layer for Python callers
(tests, ``bench.py``): ctypes bindings of ``include/ptmi.h`` plus a ``Pathtracer`` class with the
reference's method names.  PyTorch is only plumbing here: callers allocate the frame buffer
as a CUDA tensor and pass ``tensor.data_ptr()``.

There is no fallback: if ``libptmi.so`` has not been built, importing this module raises.
"""
from __future__ import annotations

import ctypes
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PTMI_LIBRARY: another build of the same library (measurement builds of the tools); there is no fallback either way
LIB_PATH = os.environ.get("PTMI_LIBRARY") or os.path.join(_HERE, "libptmi.so")

# ---- POD layouts of include/ptmi.h -------------------------------------------------------------
CAMERA_DTYPE = np.dtype([("pos", "<f4", 3), ("px_x", "<f4", 3), ("px_y", "<f4", 3),
                         ("first_ray_dir", "<f4", 3), ("width", "<u4"), ("height", "<u4")])
MATERIAL_DTYPE = np.dtype([("color", "<f4", 3), ("coef", "<f4"), ("type", "<i4")])
SPHERE_DTYPE = np.dtype([("pos", "<f4", 3), ("radius", "<f4"), ("material_id", "<u4"),
                         ("light_id", "<i4")])
# light.h:47-52: the union {Point pos; uint32 primId} is exposed as raw words
LIGHT_DTYPE = np.dtype([("type", "<i4"), ("u_bits", "<u4", 3), ("intensity", "<f4", 3)])
CAMERA_STATE_DTYPE = np.dtype([("view", "<f4", (3, 3)), ("dist", "<f4"), ("hfov", "<f4"), ("vfov", "<f4"),
                               ("pod", CAMERA_DTYPE)])
assert CAMERA_STATE_DTYPE.itemsize == 104
assert CAMERA_DTYPE.itemsize == 56 and MATERIAL_DTYPE.itemsize == 20
assert SPHERE_DTYPE.itemsize == 24 and LIGHT_DTYPE.itemsize == 28
# pt_triangle: this build's extension (the reference has spheres only, primitive.h:25-28); triangle k is
# primitive len(spheres) + k
TRIANGLE_DTYPE = np.dtype([("v0", "<f4", 3), ("v1", "<f4", 3), ("v2", "<f4", 3), ("material_id", "<u4"),
                           ("light_id", "<i4")])
assert TRIANGLE_DTYPE.itemsize == 44

DIFFUSE, SPECULAR, TRANSMISSIVE = 0, 1, 2
POINT_LIGHT, AREA_LIGHT = 0, 1
ACCUM_RUNNING_MEAN, ACCUM_SUM = 0, 1
HIT_MISS, HIT_NOT_TRACED = -1, -2
SCENE_DEMO, SCENE_SYNTH10K, SCENE_MIXED = 0, 1, 2
VARIANT_AUTO, VARIANT_REGENERATION, VARIANT_CLASSIC, VARIANT_WAVEFRONT, VARIANT_STEALING = -1, 0, 1, 100, 500
ACCEL_AUTO, ACCEL_BRUTE_FORCE = 0, 1
INTEGRATOR_PATH, INTEGRATOR_NEE_MIS = 0, 1
LAUNCH_NONE, LAUNCH_MEGAKERNEL, LAUNCH_MEGAKERNEL_STEALING, LAUNCH_WAVEFRONT, LAUNCH_WAVEFRONT_WHOLE_PATHS, LAUNCH_PATH_POOL = range(6)
TUNE_REGEN_MIN_IDLE, TUNE_CAMERA_RAY_RING, TUNE_STEAL_UNIT, TUNE_ITERATION_RANGES, TUNE_WAVEFRONT_REFILL = 0, 1, 2, 3, 4
TUNE_BVH_NODE_STEPS, TUNE_BVH_STACK, TUNE_WAVEFRONT_BATCH, TUNE_WAVEFRONT_WHOLE_PATHS, TUNE_POOL_ITERATIONS = 5, 6, 7, 8, 9
TUNE_BUFFER_SETS = 10
TUNE_POOL_WORKGROUPS = 11
PT_OK, PT_ERR_INVALID, PT_ERR_STATE, PT_ERR_DEVICE, PT_ERR_NOMEM = 0, -1, -2, -3, -4

# Every symbol include/ptmi.h declares: (name, restype, argtypes)
_vp, _u32, _u64, _i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int
_pu32, _pu64, _pf = ctypes.POINTER(_u32), ctypes.POINTER(_u64), ctypes.POINTER(ctypes.c_float)
ABI = [
    ("pt_create", _i32, [ctypes.POINTER(_vp), _vp, _i32, _u64]),
    ("pt_init_scene", _i32, [_vp, _vp, _u32, _vp, _u32, _vp, _u32]),
    ("pt_init_scene_triangles", _i32, [_vp, _vp, _u32, _vp, _u32, _vp, _u32, _vp, _u32]),
    ("pt_destroy", _i32, [_vp]),
    ("pt_last_error", ctypes.c_char_p, [_vp]),
    ("pt_run", _i32, [_vp, _vp, _i32]),
    ("pt_run_n", _i32, [_vp, _vp, _u32, _i32]),
    ("pt_run_async", _i32, [_vp, _vp, _u32]),
    ("pt_wait", _i32, [_vp]),
    ("pt_run_n_noalloc", _i32, [_vp, _vp, _u32, _i32]),
    ("pt_reset", _i32, [_vp]),
    ("pt_iteration", _u32, [_vp]),
    ("pt_set_camera", _i32, [_vp, _vp]),
    ("pt_set_stream", _i32, [_vp, _vp]),
    ("pt_set_accum_mode", _i32, [_vp, _i32]),
    ("pt_set_iteration", _i32, [_vp, _u32]),
    ("pt_set_row_bands", _i32, [_vp, _u32, _u32, _u32]),
    ("pt_ray_count", _i32, [_vp, _pu64]),
    ("pt_reset_counters", _i32, [_vp]),
    ("pt_set_kernel_variant", _i32, [_vp, _i32]),
    ("pt_debug_set_tuning", _i32, [_vp, _i32, _i32]),
    ("pt_memory_footprint", _i32, [_vp, _u32, _pu64]),
    ("pt_set_memory_limit", _i32, [_vp, _u64]),
    ("pt_save_checkpoint", _i32, [_vp, _vp, ctypes.c_char_p]),
    ("pt_load_checkpoint", _i32, [_vp, _vp, ctypes.c_char_p]),
    ("pt_set_integrator", _i32, [_vp, _i32]),
    ("pt_set_accel", _i32, [_vp, _i32]),
    ("pt_accel_nodes", _u32, [_vp]),
    ("pt_set_accel_budget", _i32, [_vp, _u32]),
    ("pt_accel_stats", _i32, [_vp, _pu64]),
    ("pt_lane_stats", _i32, [_vp, _pu64]),
    ("pt_debug_build_bvh", _i32, [_vp, _u32, _pu32, _vp, _vp, _vp]),
    ("pt_debug_bvh_grid", _i32, [_vp, _u32, _vp, _vp, _u32]),
    ("pt_debug_plan_batches", _i32, [_u32, _u32, _u32, _u32, _u32, _u32, _u32, _u64, _pu32]),
    ("pt_last_kernel_ms", _i32, [_vp, _pf]),
    ("pt_last_launch", _i32, [_vp, ctypes.POINTER(ctypes.c_int)]),
    ("pt_read_image", _i32, [_vp, _vp, _vp]),
    ("pt_debug_trace", _i32, [_vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _pu64]),
    ("pt_selftest_eval", _i32, [_vp, _i32, _vp, _vp, _vp, _u64]),
    ("pt_selftest_sample_material", _i32, [_vp, _vp, _vp, _vp, _u64]),
    ("pt_selftest_exhaustive", _i32, [_vp, _i32, _pu64]),
    ("pt_camera_state_init", _i32, [_vp, _vp, _vp, _vp, _u32, _u32, ctypes.c_float]),
    ("pt_camera_translate", _i32, [_vp, _vp]),
    ("pt_camera_rotate", _i32, [_vp, _vp]),
    ("pt_camera_init", _i32, [_vp, _vp, _vp, _vp, _u32, _u32, ctypes.c_float]),
    ("pt_scene_build", _i32, [_i32, _u64, _vp, _pu32, _vp, _pu32, _vp, _pu32, _vp, _vp, _vp, _pf]),
    ("pt_version", _i32, []),
]

_lib = None


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.

    A PyTorch-ROCm wheel ships its own libamdhip64 / libhsa-runtime64 and loads them by path.
    If libptmi.so were loaded first it would bind to the system copies under /opt/rocm, and a
    later ``import torch`` would bring a second runtime into the process: device pointers of one
    are foreign to the other, and whichever initialises second may find no device.  Mapping
    torch's copies first (same SONAME) makes libptmi.so resolve to them in either import order.
    Without PyTorch installed the system runtime is used as linked."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


def build_id() -> str:
    """Identifies the build of libptmi.so this module loads: the first 16 hex digits of the file's
    SHA-256.  Profiles record it, so that counters taken from one build are not quoted for another."""
    import hashlib
    with open(LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def load_library() -> ctypes.CDLL:
    """Load libptmi.so (built in-tree by ``make lib``); raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not built: run `make lib` (hipcc --offload-arch=gfx950); "
                              "there is no CPU fallback")
        _share_hip_runtime_with_torch()
        lib = ctypes.CDLL(LIB_PATH)
        for name, restype, argtypes in ABI:
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


class PtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libptmi error {code}: {message}")
        self.code = code


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f3(v) -> np.ndarray:
    a = np.ascontiguousarray(v, dtype=np.float32)
    if a.shape != (3,):
        raise ValueError("expected 3 floats")
    return a


class Camera:
    """``Camera(eye, lookAt, up, width, height, lensRadius, focalDistance, fov)``, camera.h:31-57.

    Lens radius and focal distance are accepted and ignored, as in the reference ("TODO depth of
    field", camera.h:68)."""

    def __init__(self, eye, look_at, up, width: int, height: int, lens_radius: float = 0.0,
                 focal_distance: float = 0.0, fov: float = 60.0):
        self.eye, self.look_at, self.up = _f3(eye), _f3(look_at), _f3(up)
        self.fov = float(fov)
        self._state = np.zeros(1, dtype=CAMERA_STATE_DTYPE)
        rc = load_library().pt_camera_state_init(_ptr(self._state), _ptr(self.eye), _ptr(self.look_at),
                                                 _ptr(self.up), width, height, self.fov)
        if rc != PT_OK:
            raise PtError(rc, "pt_camera_state_init: invalid arguments")
        self._updated = True

    @property
    def pod(self) -> np.ndarray:
        """The derived camera vectors GenerateRay reads (what pt_create / pt_set_camera take)."""
        return self._state["pod"]

    def Width(self) -> int:
        return int(self.pod["width"][0])

    def Height(self) -> int:
        return int(self.pod["height"][0])

    def Translate(self, delta) -> None:
        """camera.h:79-90: move by delta along the camera's own u, v, w axes."""
        load_library().pt_camera_translate(_ptr(self._state), _ptr(_f3(delta)))
        self._updated = True

    def Rotate(self, theta) -> None:
        """camera.h:97-129."""
        load_library().pt_camera_rotate(_ptr(self._state), _ptr(_f3(theta)))
        self._updated = True

    def IsUpdated(self) -> bool:
        """camera.h:134: reports and clears the changed flag."""
        b, self._updated = self._updated, False
        return b


def build_scene(scene_id: int = SCENE_DEMO, seed: int = 1):
    """Built-in scenes.  Returns (materials, spheres, lights, (eye, look_at, up, fov))."""
    lib = load_library()
    nm, ns, nl = _u32(0), _u32(0), _u32(0)
    eye, look_at, up = (np.zeros(3, np.float32) for _ in range(3))
    fov = ctypes.c_float(0)
    rc = lib.pt_scene_build(scene_id, seed, None, ctypes.byref(nm), None, ctypes.byref(ns), None,
                            ctypes.byref(nl), _ptr(eye), _ptr(look_at), _ptr(up), ctypes.byref(fov))
    if rc != PT_OK:
        raise PtError(rc, f"pt_scene_build: unknown scene {scene_id}")
    mats = np.zeros(nm.value, MATERIAL_DTYPE)
    sph = np.zeros(ns.value, SPHERE_DTYPE)
    lights = np.zeros(nl.value, LIGHT_DTYPE)
    rc = lib.pt_scene_build(scene_id, seed, _ptr(mats), ctypes.byref(nm), _ptr(sph), ctypes.byref(ns),
                            _ptr(lights), ctypes.byref(nl), None, None, None, None)
    if rc != PT_OK:
        raise PtError(rc, "pt_scene_build failed")
    return mats, sph, lights, (eye, look_at, up, float(fov.value))


def triangle_room():
    """The demo scene's room (main.cpp:153-161) with its walls made of triangles -- an extension: the
    reference has spheres only (primitive.h:25-28).  Twelve wall triangles whose normals point INTO the room
    (the trace loop never flips a normal, pathtracer.cu:130), a quad light of two triangles under the
    ceiling facing down, the demo scene's mirror and glass spheres and a small diffuse one.
    Returns (materials, spheres, triangles, lights, (eye, look_at, up, fov))."""
    mats = np.zeros(6, MATERIAL_DTYPE)
    mats["color"] = [[1, 1, 1], [.75, .25, .25], [.25, .25, .75], [.75, .75, .75], [.999, .999, .999], [.999, .999, .999]]
    mats["type"] = [0, 0, 0, 0, 1, 2]
    mats["coef"] = [0, 0, 0, 0, 1, 1.5]
    x0, x1, y0, y1, z0, z1 = -50.0, 50.0, 0.0, 80.0, -80.0, 80.0

    def quad(a, b, c, d, mat, light=-1):    # triangles a-c-b and a-d-c: the normal is (c - a) x (b - a)
        t = np.zeros(2, TRIANGLE_DTYPE)
        t["v0"], t["v1"], t["v2"] = [a, a], [c, d], [b, c]
        t["material_id"], t["light_id"] = mat, light
        return t

    ly = y1 - 0.5
    tris = np.concatenate([
        quad((x1, y0, z0), (x1, y1, z0), (x1, y1, z1), (x1, y0, z1), 2),      # x = +50, normal -x
        quad((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0), 1),      # x = -50, normal +x
        quad((x0, y0, z0), (x0, y1, z0), (x1, y1, z0), (x1, y0, z0), 3),      # back, normal +z
        quad((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1), 3),      # front, normal -z
        quad((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1), 3),      # floor, normal +y
        quad((x0, y1, z0), (x0, y1, z1), (x1, y1, z1), (x1, y1, z0), 3),      # ceiling, normal -y
        quad((-15, ly, -55), (-15, ly, -25), (15, ly, -25), (15, ly, -55), 0, 0),   # light, normal -y
    ])
    sph = np.zeros(3, SPHERE_DTYPE)
    sph["pos"] = [[-25, 16.5, -50], [25, 16.5, -25], [0, 6, -10]]
    sph["radius"] = [16.5, 16.5, 6]
    sph["material_id"] = [4, 5, 1]
    sph["light_id"] = -1
    lights = np.zeros(1, LIGHT_DTYPE)
    lights["type"] = AREA_LIGHT
    lights["u_bits"][0, 0] = len(sph) + 12          # the light's first triangle
    lights["intensity"] = [[12, 12, 12]]
    return mats, sph, tris, lights, (np.float32([0, 45, 79.5]), np.float32([0, 35, 0]), np.float32([0, 1, 0]), 60.0)


def debug_plan_batches(width, height, n_iterations, bytes_per_path, memory_limit=0, bands=None):
    """(rows owned, iterations per batch, buffer sets, slots per sub-queue) -- the host-side batch
    planner of the path pool / wavefront variant; bands = (band_rows, rank, world).  No device needed."""
    lib = load_library()
    out = (_u32 * 4)()
    band_rows, rank, world = bands if bands else (0, 0, 1)
    if lib.pt_debug_plan_batches(width, height, band_rows, rank, world, n_iterations, bytes_per_path, memory_limit, out) != PT_OK:
        raise ValueError("pt_debug_plan_batches: invalid arguments")
    return tuple(int(v) for v in out)


def debug_build_bvh(spheres):
    """Host-side BVH of `spheres` as pt_init_scene would build it (no device needed).
    Returns None when no hierarchy is built, else a dict: nodes [n,32] float32 (four-wide nodes in
    device layout: lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4] hi.z[4] refs[4] pad[4]), leaf_index
    [leaves,4], big_index, depth (wide levels)."""
    lib = load_library()
    sph = np.ascontiguousarray(spheres, SPHERE_DTYPE)
    counts = (_u32 * 4)()
    if lib.pt_debug_build_bvh(_ptr(sph), len(sph), counts, None, None, None) != PT_OK:
        raise ValueError("pt_debug_build_bvh: invalid arguments")
    if counts[0] == 0:
        return None
    nodes = np.zeros((counts[0], 32), np.float32)
    leaf = np.zeros((counts[1], 4), np.int32)
    big = np.zeros(counts[2], np.int32)
    lib.pt_debug_build_bvh(_ptr(sph), len(sph), counts, _ptr(nodes), _ptr(leaf), _ptr(big))
    # the 16-bit grid form the device reads: planes as origin + q * step
    grid = np.zeros(6, np.float32)
    qnodes = np.zeros((counts[0], 16), np.uint32)
    if lib.pt_debug_bvh_grid(_ptr(sph), len(sph), _ptr(grid), _ptr(qnodes), counts[0]) != PT_OK:
        raise ValueError("pt_debug_bvh_grid failed")
    return {"nodes": nodes, "leaf_index": leaf, "big_index": big, "depth": int(counts[3]),
            "grid_origin": grid[:3], "grid_step": grid[3:], "qnodes": qnodes}


class Pathtracer:
    """Mirror of the reference's ``class Pathtracer`` (pathtracer.h:38-65).

    ``Pathtracer(camera, nBounces)``, ``Init(mv, pv, lv)``, ``Run(devBuffer)``, ``Reset()``,
    ``GetIteration()``, ``Release()`` keep the reference's meaning; ``seed`` replaces the
    reference's ``time(NULL)`` cuRAND seed (pathtracer.cu:216)."""

    def __init__(self, camera, n_bounces: int, seed: int = 1):
        self._lib = load_library()
        pod = camera.pod if isinstance(camera, Camera) else np.ascontiguousarray(camera, CAMERA_DTYPE)
        self.width, self.height = int(pod["width"].reshape(-1)[0]), int(pod["height"].reshape(-1)[0])
        self.max_bounces = int(n_bounces)
        h = _vp()
        rc = self._lib.pt_create(ctypes.byref(h), _ptr(pod), n_bounces, seed)
        if rc != PT_OK:
            raise PtError(rc, self._lib.pt_last_error(None).decode())
        self._h = h

    def _check(self, rc: int):
        if rc != PT_OK:
            raise PtError(rc, self._lib.pt_last_error(self._h).decode())

    # -- the reference's five methods -----------------------------------------------------------
    def Init(self, materials, primitives, lights, triangles=None) -> bool:
        """``triangles`` (TRIANGLE_DTYPE records, optional): this build's extension, primitives behind the
        spheres; such scenes render through the path pool with the path integrator only."""
        m = np.ascontiguousarray(materials, MATERIAL_DTYPE)
        p = np.ascontiguousarray(primitives, SPHERE_DTYPE)
        l = np.ascontiguousarray(lights, LIGHT_DTYPE)
        if triangles is None or len(triangles) == 0:
            self._check(self._lib.pt_init_scene(self._h, _ptr(m), len(m), _ptr(p), len(p), _ptr(l), len(l)))
        else:
            t = np.ascontiguousarray(triangles, TRIANGLE_DTYPE)
            self._check(self._lib.pt_init_scene_triangles(self._h, _ptr(m), len(m), _ptr(p), len(p), _ptr(t), len(t),
                                                          _ptr(l), len(l)))
        return True

    def Run(self, dev_buffer) -> None:
        """``dev_buffer``: device pointer (int) or a torch CUDA tensor of width*height*3 floats."""
        self._check(self._lib.pt_run(self._h, _vp(_devptr(dev_buffer, self)), 1))

    def Reset(self) -> None:
        self._check(self._lib.pt_reset(self._h))

    def GetIteration(self) -> int:
        return int(self._lib.pt_iteration(self._h))

    def Release(self) -> None:
        if self._h:
            self._lib.pt_destroy(self._h)
            self._h = None

    # -- extensions ---------------------------------------------------------------------------
    def RunHost(self, host_buffer: np.ndarray) -> None:
        assert host_buffer.dtype == np.float32 and host_buffer.size == self.width * self.height * 3
        self._check(self._lib.pt_run(self._h, _ptr(host_buffer), 0))

    def RunN(self, dev_buffer, n_iterations: int, sync: bool = True) -> None:
        self._check(self._lib.pt_run_n(self._h, _vp(_devptr(dev_buffer, self)), n_iterations, int(sync)))

    def RunAsync(self, dev_buffer, n_iterations: int = 1) -> None:
        """Run() without the wait (pt_run_async): consecutive calls overlap on the device; Wait()
        ends the series.  Same image as Run / RunN, bit for bit."""
        self._check(self._lib.pt_run_async(self._h, _vp(_devptr(dev_buffer, self)), n_iterations))

    def Wait(self) -> None:
        self._check(self._lib.pt_wait(self._h))

    def RunNNoAlloc(self, dev_buffer, n_iterations: int, sync: bool = True) -> None:
        """RunN through the entry point that allocates no device memory (static scratch)."""
        self._check(self._lib.pt_run_n_noalloc(self._h, _vp(_devptr(dev_buffer, self)), n_iterations, int(sync)))

    def SetCamera(self, camera) -> None:
        pod = camera.pod if isinstance(camera, Camera) else np.ascontiguousarray(camera, CAMERA_DTYPE)
        self._check(self._lib.pt_set_camera(self._h, _ptr(pod)))

    def SetStream(self, stream_ptr: int) -> None:
        self._check(self._lib.pt_set_stream(self._h, _vp(stream_ptr)))

    def SetAccumMode(self, mode: int) -> None:
        self._check(self._lib.pt_set_accum_mode(self._h, mode))

    def SetIteration(self, iteration: int) -> None:
        self._check(self._lib.pt_set_iteration(self._h, iteration))

    def SetRowBands(self, band_rows: int, rank: int, world: int) -> None:
        self._check(self._lib.pt_set_row_bands(self._h, band_rows, rank, world))

    def SetKernelVariant(self, variant: int) -> None:
        """One of the VARIANT_* values.  Tests and tools also pass the shorthand codes of the
        tuning experiments, which are spelled out here as a variant plus pt_debug_set_tuning calls:
        2..64 regeneration once that many lanes are idle; 300 / 301 camera-ray ring off / on;
        500 + 1 stealing without the ring, + 2 / + 3 single paths / whole pixel-iterations change
        hands, + 10 + k: k + 1 iteration ranges; 100 + 1 wavefront with static assignment,
        100 + k (k >= 2) refill threshold k."""
        base, tunings = variant, []
        if 2 <= variant <= 64:
            base, tunings = VARIANT_REGENERATION, [(TUNE_REGEN_MIN_IDLE, variant)]
        elif variant in (300, 301):
            base, tunings = VARIANT_REGENERATION, [(TUNE_CAMERA_RAY_RING, variant - 300)]
        elif VARIANT_WAVEFRONT < variant <= VARIANT_WAVEFRONT + 64:
            k = variant - VARIANT_WAVEFRONT
            base, tunings = VARIANT_WAVEFRONT, [(TUNE_WAVEFRONT_REFILL, 0 if k == 1 else k)]
        elif variant == VARIANT_STEALING + 1:
            base, tunings = VARIANT_STEALING, [(TUNE_CAMERA_RAY_RING, 0)]
        elif variant in (VARIANT_STEALING + 2, VARIANT_STEALING + 3):
            base, tunings = VARIANT_STEALING, [(TUNE_STEAL_UNIT, 1 if variant == VARIANT_STEALING + 2 else 4)]
        elif VARIANT_STEALING + 10 <= variant <= VARIANT_STEALING + 26:
            base, tunings = VARIANT_STEALING, [(TUNE_ITERATION_RANGES, variant - (VARIANT_STEALING + 10) + 1)]
        self._check(self._lib.pt_set_kernel_variant(self._h, base))
        for key, value in tunings:
            self.SetTuning(key, value)

    def SetTuning(self, key: int, value: int) -> None:
        self._check(self._lib.pt_debug_set_tuning(self._h, key, value))

    def MemoryFootprint(self, n_iterations: int = 0):
        """(bytes held now, bytes at the peak of a RunN(n_iterations))."""
        out = (_u64 * 2)()
        self._check(self._lib.pt_memory_footprint(self._h, n_iterations, out))
        return int(out[0]), int(out[1])

    def SetMemoryLimit(self, n_bytes: int) -> None:
        self._check(self._lib.pt_set_memory_limit(self._h, n_bytes))

    def SetIntegrator(self, integrator: int) -> None:
        """INTEGRATOR_PATH: the reference's Trace; INTEGRATOR_NEE_MIS: plus direct-light sampling."""
        self._check(self._lib.pt_set_integrator(self._h, integrator))

    def SetAccel(self, mode: int) -> None:
        """ACCEL_AUTO: sphere BVH for scenes above 192 spheres; ACCEL_BRUTE_FORCE: linear scan."""
        self._check(self._lib.pt_set_accel(self._h, mode))

    def SetAccelBudget(self, visits: int) -> None:
        """Visits after which a ray is finished by a wave-wide scan (0: default)."""
        self._check(self._lib.pt_set_accel_budget(self._h, visits))

    def AccelNodes(self) -> int:
        return int(self._lib.pt_accel_nodes(self._h))

    def AccelStats(self):
        """The 32 counters of pt_accel_stats -- zeros unless the library was built with PT_BVH_STATS."""
        out = (_u64 * 32)()
        self._check(self._lib.pt_accel_stats(self._h, out))
        return tuple(int(v) for v in out)

    def RayCount(self) -> int:
        v = _u64(0)
        self._check(self._lib.pt_ray_count(self._h, ctypes.byref(v)))
        return int(v.value)

    def ResetCounters(self) -> None:
        self._check(self._lib.pt_reset_counters(self._h))

    def LastKernelMs(self) -> float:
        v = ctypes.c_float(0)
        self._check(self._lib.pt_last_kernel_ms(self._h, ctypes.byref(v)))
        return float(v.value)

    def LastLaunch(self) -> int:
        """LAUNCH_* of the last render: what PT_VARIANT_AUTO resolved to."""
        v = ctypes.c_int(0)
        self._check(self._lib.pt_last_launch(self._h, ctypes.byref(v)))
        return int(v.value)

    def ReadImage(self, dev_buffer) -> np.ndarray:
        out = np.empty((self.height, self.width, 3), np.float32)
        self._check(self._lib.pt_read_image(self._h, _vp(_devptr(dev_buffer, self)), _ptr(out)))
        return out

    def SaveCheckpoint(self, dev_buffer, path: str) -> None:
        """Frame buffer + iteration + seed/depth/size to `path`; see LoadCheckpoint."""
        self._check(self._lib.pt_save_checkpoint(self._h, _vp(_devptr(dev_buffer, self)), os.fsencode(path)))

    def LoadCheckpoint(self, dev_buffer, path: str) -> None:
        """Restores frame buffer and iteration counter; the next Run continues the render bit-exactly."""
        self._check(self._lib.pt_load_checkpoint(self._h, _vp(_devptr(dev_buffer, self)), os.fsencode(path)))

    def DebugTrace(self, x0: int, y0: int, x1: int, y1: int, iteration: int):
        """Returns (hit_ids[h,w,4,mb+1], radiance[h,w,4,3], pixel_rgb[h,w,3], ray_count)."""
        w, h = x1 - x0, y1 - y0
        hits = np.empty((h, w, 4, self.max_bounces + 1), np.int32)
        rad = np.empty((h, w, 4, 3), np.float32)
        rgb = np.empty((h, w, 3), np.float32)
        cnt = _u64(0)
        self._check(self._lib.pt_debug_trace(self._h, x0, y0, x1, y1, iteration, _ptr(hits), _ptr(rad),
                                             _ptr(rgb), ctypes.byref(cnt)))
        return hits, rad, rgb, int(cnt.value)

    def SelftestEval(self, op: int, a: np.ndarray, b: np.ndarray | None = None) -> np.ndarray:
        a = np.ascontiguousarray(a, np.float32)
        bb = None if b is None else np.ascontiguousarray(b, np.float32)
        out = np.empty_like(a)
        self._check(self._lib.pt_selftest_eval(self._h, op, _ptr(a), None if bb is None else _ptr(bb),
                                               _ptr(out), a.size))
        return out

    def SelftestSampleMaterial(self, material: np.ndarray, rows: np.ndarray) -> np.ndarray:
        """rows [n, 8] = wo.xyz, n.xyz, u, v -> [n, 8] = f.rgb, wi.xyz, pdf, 0 (one MATERIAL_DTYPE record)."""
        m = np.ascontiguousarray(material, MATERIAL_DTYPE).reshape(-1)[:1]
        x = np.ascontiguousarray(rows, np.float32).reshape(-1, 8)
        out = np.empty_like(x)
        self._check(self._lib.pt_selftest_sample_material(self._h, _ptr(m), _ptr(x), _ptr(out), len(x)))
        return out

    def SelftestExhaustive(self, op: int) -> int:
        v = _u64(0)
        self._check(self._lib.pt_selftest_exhaustive(self._h, op, ctypes.byref(v)))
        return int(v.value)

    def __del__(self):
        try:
            self.Release()
        except Exception:
            pass


def _devptr(buf, pt: "Pathtracer") -> int:
    if isinstance(buf, int):
        return buf
    # torch tensor (kept duck-typed so that importing this module does not import torch)
    if not buf.is_cuda or not buf.is_contiguous() or buf.numel() != pt.width * pt.height * 3:
        raise ValueError("frame buffer must be a contiguous CUDA float32 tensor of width*height*3")
    if str(buf.dtype) != "torch.float32":
        raise ValueError("frame buffer must be float32")
    return buf.data_ptr()
