"""Host-side planning of the path pool / wavefront batches (pt_debug_plan_batches; no device needed):
row-band ownership against sharding.owned_bands, and the invariants of the batch sizes the renders
rely on -- whole iterations, within the memory limit, two buffer sets only when there are at least two
batches, fewer than 2^32 paths per batch, sub-queues that hold their share of a batch."""
import importlib

import numpy as np
import pytest

sharding = importlib.import_module("cuda-pathtracer_amd.sharding")


@pytest.mark.parametrize("height,band_rows,world", [(1080, 16, 8), (107, 16, 3), (107, 16, 8), (16, 16, 4), (2160, 32, 5), (1, 16, 2)])
def test_owned_rows_partition_the_frame(pt, height, band_rows, world):
    total = 0
    for rank in range(world):
        rows = pt.debug_plan_batches(64, height, 1, 12, bands=(band_rows, rank, world))[0]
        bands = sharding.owned_bands(height, band_rows, rank, world)
        assert rows == sum(y1 - y0 for y0, y1 in bands), (rank, bands)
        total += rows
    assert total == height
    assert pt.debug_plan_batches(64, height, 1, 12)[0] == height


@pytest.mark.parametrize("per_path", [12, 56, 100])
def test_batch_sizes(pt, per_path):
    rng = np.random.default_rng(per_path)
    for _ in range(400):
        w, h = int(rng.integers(1, 4000)), int(rng.integers(1, 2200))
        n = int(rng.choice([1, 2, 5, 64, 1000]))
        ppi = w * h * 4
        limit = int(rng.choice([0, per_path * ppi - 1, per_path * ppi, 3 * per_path * ppi + 5, 2**33, 2**40]))
        rows, batch, sets, sub_cap = pt.debug_plan_batches(w, h, n, per_path, limit)
        assert rows == h
        budget = limit if limit else (64 << 20) * 100
        if limit and limit < per_path * ppi:
            assert (batch, sets) == (0, 0)                      # not one iteration fits: the caller falls back
            continue
        assert 1 <= batch <= n
        assert batch * ppi < 2**32                              # a batch's paths are numbered with 32 bits
        if limit or per_path * ppi <= budget:                   # without a limit one iteration always goes
            assert sets * batch * ppi * per_path <= budget
        if sets == 2:
            assert batch < n                                    # two sets only when the render is more than one batch
        blocks = -(-batch * ppi // 256)
        assert sub_cap % 256 == 0 and sub_cap * 64 >= blocks * 256 and (sub_cap - 256) * 64 < blocks * 256


def test_rejects_nonsense(pt):
    for args in ((0, 10, 1, 12), (10, 0, 1, 12), (10, 10, 1, 0)):
        with pytest.raises(ValueError):
            pt.debug_plan_batches(*args)
    with pytest.raises(ValueError):
        pt.debug_plan_batches(10, 10, 1, 12, bands=(16, 3, 3))
    assert pt.debug_plan_batches(10, 10, 0, 12)[1:3] == (0, 0)


def test_frames_outside_the_index_arithmetic_are_not_planned(pt):
    """The pool and wavefront kernels divide path numbers through a float reciprocal, exact for
    quotients below 2^20 (rows, iterations of a batch) and divisors below 2^30 (band_rows * world):
    taller frames and wider band periods are not given to them (batch 0: the megakernel renders),
    and no batch holds 2^20 iterations or more."""
    assert pt.debug_plan_batches(16, (1 << 20) - 1, 1, 12)[1:3] == (1, 1)
    assert pt.debug_plan_batches(16, 1 << 20, 1, 12)[1:3] == (0, 0)
    assert pt.debug_plan_batches(1, 5_000_000, 4, 12)[1:3] == (0, 0)
    assert pt.debug_plan_batches(64, 1080, 4, 12, bands=(16, 0, (1 << 26) - 1))[1] > 0
    assert pt.debug_plan_batches(64, 1080, 4, 12, bands=(16, 0, 1 << 26))[1:3] == (0, 0)
    # a tiny frame and millions of iterations: the memory would hold them all, the division would not
    rows, batch, sets, _ = pt.debug_plan_batches(2, 2, 1 << 24, 12)
    assert rows == 2 and 1 <= batch < (1 << 20) and sets in (1, 2)
