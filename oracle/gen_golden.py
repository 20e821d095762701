#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/*.npz.

Runs ONLY in the development container: it drives oracle/_ref/libpt_ref.so, the host build of
the reference's own headers + Trace() (see oracle/ref_harness.cpp), and stores inputs and
outputs as data.  The GPU box has no reference tree; the tests there read these files.

    make -C oracle ref && python oracle/gen_golden.py

Every case stores: the camera arguments and the camera POD the reference's constructor
derived, the scene arrays, seed / depth / iteration range, and the reference's outputs:
image, per-path hit-id sequences (int16), per-path radiance and the Intersect-call count.
"""
from __future__ import annotations

import hashlib
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import oracle_py as O  # noqa: E402

pt = importlib.import_module("cuda-pathtracer_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def case(name, scene_id, width, height, max_bounces, seed, first_iteration=1, n_iterations=1,
         scene_seed=1, scene=None, store_paths=True, libm_stats=False):
    if scene is None:
        mats, sph, lights, (eye, look, up, fov) = pt.build_scene(scene_id, scene_seed)
    else:
        mats, sph, lights, (eye, look, up, fov) = scene
    cam = O.camera_init(eye, look, up, width, height, fov, lib=O.load_ref(), fn="ref_camera_init")
    img, hits, rad, rays = O.ref_render(eye, look, up, width, height, fov, mats, sph, lights, seed,
                                        max_bounces, first_iteration, n_iterations, want_paths=True)
    assert hits.min() >= -2 and hits.max() < 32767
    out = dict(eye=eye, look_at=look, up=up, fov=np.float32(fov), width=width, height=height,
               camera_pod=cam, materials=mats, spheres=sph, lights=lights, seed=np.uint64(seed),
               max_bounces=max_bounces, first_iteration=first_iteration, n_iterations=n_iterations,
               image=img, ray_count=np.uint64(rays),
               hit_ids_sha256=sha(hits), path_radiance_sha256=sha(rad))
    if store_paths:
        out["hit_ids"] = hits.astype(np.int16)
        out["path_radiance"] = rad
    if libm_stats:
        # the same reference build with libm's sinf/cosf instead of the deterministic pair:
        # bounds what the substitution does to the image (documentation, not a parity target)
        img2, hits2, _, rays2 = O.ref_render(eye, look, up, width, height, fov, mats, sph, lights,
                                             seed, max_bounces, first_iteration, n_iterations,
                                             want_paths=True, libm=True)
        out["libm_image_rmse"] = np.float64(np.sqrt(((img2.astype(np.float64) - img) ** 2).mean()))
        out["libm_mean_abs_diff_of_means"] = np.float64(abs(img2.mean(dtype=np.float64) - img.mean(dtype=np.float64)))
        out["libm_hit_mismatch_fraction"] = np.float64((hits2 != hits).mean())
        out["libm_first_hit_mismatch"] = np.int64((hits2[..., 0] != hits[..., 0]).sum())
        out["libm_ray_count"] = np.uint64(rays2)
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {width}x{height} d{max_bounces} it{first_iteration}+{n_iterations} rays={rays} "
          f"mean={img.mean():.6f} -> {os.path.getsize(path) / 1024:.0f} KiB")


def empty_scene():
    mats = np.zeros(1, O.MATERIAL_DTYPE)
    mats["color"] = 0.5
    return (mats, np.zeros(0, O.SPHERE_DTYPE), np.zeros(0, O.LIGHT_DTYPE),
            (np.array([0, 45, 79.5], np.float32), np.array([0, 35, 0], np.float32),
             np.array([0, 1, 0], np.float32), 60.0))


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    ref = O.load_ref()
    assert ref.ref_uses_libm_sincos() == 0

    # the CPU-runnable benchmark configuration: demo scene 256x256, 1 iteration, depth 4
    case("demo_256x256_d4_it1", pt.SCENE_DEMO, 256, 256, 4, seed=1, store_paths=False, libm_stats=True)
    # the same at a size whose per-path data is small enough to commit
    case("demo_64x64_d4_it1", pt.SCENE_DEMO, 64, 64, 4, seed=1)
    # running mean over iterations 1..4, and a resumed range 3..4 on a ragged size
    case("demo_64x48_d8_it1-4", pt.SCENE_DEMO, 64, 48, 8, seed=7, n_iterations=4)
    case("demo_37x21_d8_it3-4", pt.SCENE_DEMO, 37, 21, 8, seed=7, first_iteration=3, n_iterations=2)
    # depth 0 (only emission of the first hit is ever seen) and depth 16 (Russian roulette)
    case("demo_32x32_d0_it1", pt.SCENE_DEMO, 32, 32, 0, seed=3)
    case("mixed_64x64_d16_it1-2", pt.SCENE_MIXED, 64, 64, 16, seed=11, n_iterations=2)
    # 10 007 spheres: exercises the LDS-tiled scan (index order / tie-break across tiles)
    case("synth10k_32x32_d8_it1", pt.SCENE_SYNTH10K, 32, 32, 8, seed=5)
    # no primitives at all: every primary ray misses
    case("empty_16x16_d4_it1", None, 16, 16, 4, seed=1, scene=empty_scene())

    # camera PODs derived by the reference's constructor at the sizes of the benchmark configurations
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO)
    sizes = [(256, 256), (640, 480), (1920, 1080), (3840, 2160), (17, 5)]
    pods = np.concatenate([O.camera_init(eye, look, up, w, h, fov, lib=ref, fn="ref_camera_init")
                           for (w, h) in sizes])
    # an off-axis camera as well
    eye2, look2, up2 = np.array([10, 20, 60], np.float32), np.array([-5, 30, -40], np.float32), \
        np.array([0.1, 1, 0], np.float32)
    pod2 = O.camera_init(eye2, look2, up2, 320, 200, 75.0, lib=ref, fn="ref_camera_init")
    np.savez_compressed(os.path.join(GOLDEN, "cameras.npz"), eye=eye, look_at=look, up=up,
                        fov=np.float32(fov), sizes=np.array(sizes, np.uint32), pods=pods,
                        eye2=eye2, look_at2=look2, up2=up2, fov2=np.float32(75.0), pod2=pod2,
                        sizeof=np.array([ref.ref_sizeof(i) for i in range(11)], np.int32))
    # interactive camera: a keyboard/mouse session's worth of Translate / Rotate calls
    # (main.cpp:249-364 uses steps of 1.0 and 0.05 rad) through the reference's own methods
    ops = np.array([(0, 1, 0, 0), (0, 0, 0, -1), (1, 0.05, 0, 0), (1, 0, -0.05, 0), (0, -2.5, 0.5, 3),
                    (1, 0.3, 0.2, 0), (1, -0.7, 0.1, 0.4), (0, 0, 1, 0), (1, 0, 0.05, 0), (1, 0.05, 0, 0)],
                   np.float32)
    pods_moved = O.ref_camera_moves(eye, look, up, 640, 480, fov, ops)
    np.savez_compressed(os.path.join(GOLDEN, "camera_moves.npz"), eye=eye, look_at=look, up=up,
                        fov=np.float32(fov), width=640, height=480, ops=ops, pods=pods_moved)
    print("cameras.npz, camera_moves.npz written")


def nee_inputs(rng, n, sph):
    """Seeded inputs of the 13 direct-light sampling functions (rows of <= 16 floats)."""
    def unit(k):
        v = rng.normal(size=(k, 3))
        return v / np.linalg.norm(v, axis=1, keepdims=True)
    col = lambda a: np.asarray(a, np.float64).reshape(n, -1)
    pos = rng.normal(size=(n, 3)) * 50
    rad = rng.random((n, 1)) * 20 + 0.1
    # points inside, on, just outside and far outside the sphere
    pfrom = pos + unit(n) * rad * rng.choice([0.5, 1.0, 1.00001, 1.5, 3.0, 30.0], (n, 1))
    room = np.column_stack([rng.uniform(-49, 49, n), rng.uniform(1, 79, n), rng.uniform(-79, 79, n)])
    room[:n // 10] = sph["pos"][8] + unit(n // 10) * 500.0 * rng.choice([0.9, 1.0, 1.0000001], (n // 10, 1))
    ins = {
        "SAMPLE_SPHERE": rng.random((n, 2)),
        "SAMPLE_CONE": np.hstack([rng.random((n, 3)), unit(n), unit(n), unit(n)]),
        "CONE_PDF": rng.random((n, 1)),
        "POWER_HEURISTIC": np.hstack([col(rng.integers(1, 4, n)), rng.random((n, 1)) * 5,
                                      col(rng.integers(1, 4, n)), rng.random((n, 1)) * 5]),
        "COORD_SYSTEM": unit(n),
        "MATERIAL": np.hstack([col(rng.integers(0, 3, n)), rng.random((n, 3)), unit(n), unit(n), unit(n)]),
        "LIGHT_L": np.hstack([rng.random((n, 3)) * 10, unit(n), unit(n)]),
        "SEGMENT": np.hstack([rng.normal(size=(n, 3)) * 50, np.full((n, 1), 3e-2), rng.normal(size=(n, 3)) * 50,
                              col(rng.choice([0, 3e-2], n))]),
        "SPHERE_SAMPLE_AREA": np.hstack([pos, rad, rng.random((n, 2))]),
        "SPHERE_SAMPLE_FROM": np.hstack([pos, rad, pfrom, rng.random((n, 2))]),
        "LIGHT_PDF": np.hstack([np.full((n, 1), 8), room, unit(n)]),
        "INTERSECT_P": np.hstack([room, unit(n), np.full((n, 1), 3e-2), rng.random((n, 1)) * 200]),
        "SAMPLE_LIGHT": np.hstack([col(rng.integers(0, 2, n)), room, rng.random((n, 2))]),
    }
    return {k: v.astype(np.float32) for k, v in ins.items()}


def nee_components():
    """The reference's direct-light sampling functions (never called by its Trace), evaluated by the
    reference's own host build: pins the oracle's `as_written` restatement of them."""
    mats, sph, lights, _ = pt.build_scene(pt.SCENE_DEMO)
    # a second light, of the point kind, so that both branches of SampleLight are covered
    lights = np.concatenate([lights, np.zeros(1, lights.dtype)])
    lights["type"][1] = 0
    lights["u_bits"][1] = np.array([10.0, 60.0, -30.0], np.float32).view(np.uint32)
    lights["intensity"][1] = [400.0, 300.0, 200.0]
    rng = np.random.default_rng(20261111)
    out = {"materials": mats, "spheres": sph, "lights": lights}
    for op, x in nee_inputs(rng, 512, sph).items():
        out["in_" + op] = x
        out["ref_" + op] = O.ref_nee_eval(op, x, (mats, sph, lights))
    path = os.path.join(GOLDEN, "nee_components.npz")
    np.savez_compressed(path, **out)
    print(f"nee_components: {os.path.getsize(path) / 1024:.0f} KiB")


def block_means(img, n=32):
    """[n, n, 3] float64 means over the (h/n) x (w/n) pixel blocks of an image."""
    h, w, _ = img.shape
    return img.astype(np.float64).reshape(n, h // n, n, w // n, 3).mean(axis=(1, 3))


def t3_tolerance():
    """The stated fp32 tolerance of the build, at the benchmark's sampling: the reference's host
    build with libm's sinf/cosf (montecarlo.h:116-117 as written) against the same build with the
    deterministic pair this repository injects -- demo scene, 256x256, depth 8, 64 iterations --
    next to the difference between two seeds of the injected build (the Monte-Carlo noise floor)."""
    mats, sph, lights, (eye, look, up, fov) = pt.build_scene(pt.SCENE_DEMO, 1)
    w = h = 256
    args = (eye, look, up, w, h, fov, mats, sph, lights)
    inj, _, _, rays_inj = O.ref_render(*args, 1, 8, 1, 64)
    libm, _, _, rays_libm = O.ref_render(*args, 1, 8, 1, 64, libm=True)
    inj2, _, _, _ = O.ref_render(*args, 2, 8, 1, 64)
    _, hits_i, _, _ = O.ref_render(*args, 1, 8, 1, 1, want_paths=True)
    _, hits_l, _, _ = O.ref_render(*args, 1, 8, 1, 1, want_paths=True, libm=True)
    rmse = lambda a, b: float(np.sqrt(((a.astype(np.float64) - b) ** 2).mean()))
    out = dict(width=w, height=h, max_bounces=8, seed=np.uint64(1), n_iterations=64,
               block_means_injected=block_means(inj), block_means_libm=block_means(libm),
               image_injected_sha256=sha(inj), ray_count_injected=np.uint64(rays_inj),
               ray_count_libm=np.uint64(rays_libm),
               libm_image_rmse=np.float64(rmse(libm, inj)),
               seed_to_seed_image_rmse=np.float64(rmse(inj2, inj)),
               mean_injected=np.float64(inj.mean(dtype=np.float64)), mean_libm=np.float64(libm.mean(dtype=np.float64)),
               mean_other_seed=np.float64(inj2.mean(dtype=np.float64)),
               libm_hit_mismatch_fraction_it1=np.float64((hits_l != hits_i).mean()),
               libm_first_hit_mismatch_it1=np.int64((hits_l[..., 0] != hits_i[..., 0]).sum()))
    path = os.path.join(GOLDEN, "t3_libm_demo_256x256_d8_it1-64.npz")
    np.savez_compressed(path, **out)
    print({k: (float(v) if np.ndim(v) == 0 and not isinstance(v, str) else "...") for k, v in out.items()})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "t3":
        t3_tolerance()
    else:
        main()
        nee_components()
        t3_tolerance()
