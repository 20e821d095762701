"""The N>1 path on CPU: two gloo ranks shard a render by sample range and by row band and
reduce the accumulation buffers with torch.distributed.  The renderer is the CPU oracle here
(the GPU box runs the same plan with the HIP renderer in bench.py)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py as O
    sharding = importlib.import_module("cuda-pathtracer_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden("demo_64x48_d8_it1-4")
    args = (g["camera_pod"], g["materials"], g["spheres"], g["lights"], int(g["seed"]), int(g["max_bounces"]))

    def render_sum(first, n):
        img, _, _, _ = O.oracle_render(*args, first_iteration=first, n_iterations=n, accum_mode=O.ACCUM_SUM)
        return torch.from_numpy(img)

    def render_bands(r, w):
        img = np.zeros((48, 64, 3), np.float32)
        for (y0, y1) in sharding.owned_bands(48, 16, r, w):
            O.oracle_render(*args, n_iterations=4, rect=(0, y0, 64, y1), buffer=img)
        return torch.from_numpy(img)

    a = sharding.render_sample_split(render_sum, rank, world, 4, dist)
    # 3 iterations over 2 ranks: rank 0 renders two, rank 1 one (the remainder goes to the first ranks)
    a3 = sharding.render_sample_split(render_sum, rank, world, 3, dist)
    b = sharding.render_tile_split(render_bands, rank, world, dist)
    if rank == 0:
        np.save(os.path.join(out_dir, "sample.npy"), a.numpy())
        np.save(os.path.join(out_dir, "sample3.npy"), a3.numpy())
        np.save(os.path.join(out_dir, "tile.npy"), b.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = load_golden("demo_64x48_d8_it1-4")
    sample = np.load(tmp_path / "sample.npy")
    tile = np.load(tmp_path / "tile.npy")
    # tile split: bit-exact; sample split: summation order differs -> rounding tolerance
    assert tile.tobytes() == g["image"].tobytes()
    np.testing.assert_allclose(sample, g["image"], rtol=4e-6, atol=1e-6)
    # the uneven split against one rank rendering the same three iterations
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    whole, _, _, _ = O.oracle_render(g["camera_pod"], g["materials"], g["spheres"], g["lights"], int(g["seed"]),
                                     int(g["max_bounces"]), first_iteration=1, n_iterations=3)
    np.testing.assert_allclose(np.load(tmp_path / "sample3.npy"), whole, rtol=4e-6, atol=1e-6)


def test_plans():
    sharding = importlib.import_module("cuda-pathtracer_amd.sharding")
    assert sharding.sample_range(0, 8, 1024) == (1, 128)
    assert sharding.sample_range(7, 8, 1024) == (897, 128)
    # remainders go to the first ranks; every iteration exactly once; a rank may get none
    assert [sharding.sample_range(r, 3, 64) for r in range(3)] == [(1, 22), (23, 21), (44, 21)]
    assert [sharding.sample_range(r, 5, 3) for r in range(5)] == [(1, 1), (2, 1), (3, 1), (4, 0), (4, 0)]
    for world, total in ((8, 1000), (7, 64), (3, 2), (4, 0)):
        covered = [i for r in range(world) for f, n in [sharding.sample_range(r, world, total)] for i in range(f, f + n)]
        assert covered == list(range(1, total + 1))
    with pytest.raises(ValueError):
        sharding.sample_range(8, 8, 1)
    bands = [sharding.owned_bands(1080, 16, r, 8) for r in range(8)]
    rows = sorted(y for b in bands for (y0, y1) in b for y in range(y0, y1))
    assert rows == list(range(1080))                       # every row owned exactly once
    assert bands[3][0] == (48, 64) and bands[3][-1][1] <= 1080
    assert sharding.owned_bands(40, 16, 2, 3) == [(32, 40)]
    assert sharding.owned_bands(16, 16, 1, 2) == []
    with pytest.raises(ValueError):
        sharding.owned_bands(100, 10, 0, 2)
