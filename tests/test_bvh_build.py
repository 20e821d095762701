"""Host-side build of the sphere BVH (csrc/pt_bvh.cpp) through the pt_debug_build_bvh test hook:
structural invariants the traversal relies on.  No device needed."""
import numpy as np
import pytest


def scene(pt, rng, n, half=50.0, rmin=0.3, rmax=1.2, n_big=6):
    s = np.zeros(n + n_big, pt.SPHERE_DTYPE)
    s["pos"][:n_big] = rng.normal(size=(n_big, 3)) * 1e5
    s["radius"][:n_big] = 1e5
    s["pos"][n_big:] = rng.uniform(-half, half, (n, 3))
    s["radius"][n_big:] = rng.uniform(rmin, rmax, n)
    s["light_id"] = -1
    return s


def check(pt, sph):
    b = pt.debug_build_bvh(sph)
    if b is None:
        return None
    nodes, leaf, big = b["nodes"], b["leaf_index"], b["big_index"]
    NONE = 0x7fffffff
    refs = nodes[:, 24:28].copy().view(np.int32)                  # [node][child]
    used_slot = refs != NONE
    # slots 0 and 1 are always used (the traversal tests them unconditionally); unused slots hold NaN boxes
    assert used_slot[:, :2].all()
    # every sphere is in exactly one leaf slot or in the always-tested list
    used = leaf[leaf != NONE]
    everything = np.sort(np.concatenate([used, big]))
    assert np.array_equal(everything, np.arange(len(sph)))
    # every inner node except the root and every leaf is referenced exactly once
    live = refs[used_slot]
    inner_refs = np.sort(live[live >= 0])
    assert np.array_equal(inner_refs, np.arange(1, len(nodes)))
    leaf_refs = np.sort(~live[live < 0])
    assert np.array_equal(leaf_refs, np.arange(len(leaf)))
    # child boxes contain the spheres below them; depth is what the stack is sized from
    lo = nodes[:, 0:12].reshape(-1, 3, 4).transpose(0, 2, 1)      # [node][child][xyz]
    hi = nodes[:, 12:24].reshape(-1, 3, 4).transpose(0, 2, 1)
    assert np.isnan(lo[~used_slot]).all() and np.isnan(hi[~used_slot]).all()

    bounded = np.setdiff1d(np.arange(len(sph)), big)
    extent = float(np.abs(sph["pos"][bounded]).max() + np.abs(sph["radius"][bounded]).max())

    def walk(ref, depth):
        if ref < 0:
            idx = leaf[~ref]
            idx = idx[idx != NONE]
            c, r = sph["pos"][idx].astype(np.float64), np.abs(sph["radius"][idx].astype(np.float64))
            return (c - r[:, None]).min(0), (c + r[:, None]).max(0), 0
        d = 0
        mins, maxs = [], []
        for k in range(4):
            if refs[ref, k] == NONE:
                continue
            cl, ch, cd = walk(int(refs[ref, k]), depth + 1)
            assert (lo[ref, k] <= cl).all() and (hi[ref, k] >= ch).all(), (ref, k)
            # ... and not by much: slack is 1e-4 + 1e-6 of the cloud's size
            tol = 0.01 + 1e-5 * extent
            assert (cl - lo[ref, k]).max() < tol and (hi[ref, k] - ch).max() < tol
            mins.append(cl); maxs.append(ch); d = max(d, cd)
        return np.min(mins, axis=0), np.max(maxs, axis=0), d + 1

    _, _, depth = walk(0, 0)
    assert depth == b["depth"] <= 28

    # The 16-bit grid form the device reads: no plane of a used box sits on the grid's clamp values
    # (a clamped plane would make the box tighter than the traversal's error bound assumes), and
    # every quantised box contains its float box, as the device decodes it (origin + q * step in fp32
    # arithmetic loses less than half a step, which the outward rounding by two steps covers).
    q = b["qnodes"]
    assert np.array_equal(q[:, 12:16].view(np.int32), refs)
    planes = q[:, :12].reshape(-1, 6, 2)                          # [node][plane][pair]
    q4 = np.stack([planes[..., 0] & 0xffff, planes[..., 0] >> 16, planes[..., 1] & 0xffff, planes[..., 1] >> 16], -1)
    qlo, qhi = q4[:, 0:3].transpose(0, 2, 1), q4[:, 3:6].transpose(0, 2, 1)      # [node][child][xyz]
    assert (qlo[used_slot] > 0).all() and (qhi[used_slot] < 65535).all()
    o, st = b["grid_origin"].astype(np.float64), b["grid_step"].astype(np.float64)
    assert (o + (qlo[used_slot] + 1) * st <= lo[used_slot]).all()
    assert (o + (qhi[used_slot] - 1) * st >= hi[used_slot]).all()
    return b


def test_invariants_on_random_scenes(pt):
    rng = np.random.default_rng(1)
    assert check(pt, scene(pt, rng, 5)) is not None      # few spheres of very different sizes
    for n in (193, 1000, 10000):
        b = check(pt, scene(pt, rng, n))
        assert b is not None and len(b["big_index"]) == 6
        # cuts leave multiples of the leaf size on the left: leaves are full
        assert len(b["leaf_index"]) == -(-n // 4)
        # both sides of a binary cut hold between 3/8 and 5/8 of the range, and a four-wide node
        # folds at least ... one binary level: the wide depth is bounded by the binary bound,
        # and is about half of it
        binary_bound = np.ceil(np.log(max(n / 4, 1)) / np.log(8 / 5)) + 1
        assert b["depth"] <= binary_bound
        assert b["depth"] <= np.ceil(np.log2(max(n / 4, 2))) * 0.75 + 1
        # most child slots are in use
        refs = b["nodes"][:, 24:28].copy().view(np.int32)
        assert (refs != 0x7fffffff).mean() > 0.7


def test_degenerate_inputs(pt):
    rng = np.random.default_rng(2)
    s = scene(pt, rng, 500)
    s["pos"][6:] = [1.0, 2.0, 3.0]                   # all centres coincide
    check(pt, s)
    s = scene(pt, rng, 500)
    s["pos"][6:, 1] = 0.0                            # coplanar
    s["pos"][6:, 0] = np.round(s["pos"][6:, 0])      # many equal coordinates on the split axis
    check(pt, s)
    s = scene(pt, rng, 300)
    s["radius"][10] = -2.0                           # the sphere test only sees r*r
    s["radius"][11] = 0.0
    check(pt, s)
    s = scene(pt, rng, 300)
    s["radius"][20] = np.inf
    s["pos"][21] = [np.nan, 0, 0]
    b = check(pt, s)
    assert {20, 21} <= set(b["big_index"].tolist())  # unboundable: always tested
    # A flat, elongated cloud: the builder's slack on every box follows the cloud's diagonal
    # (1e-6 * 2e4 here) and exceeds 1/64 of the short axes' extent; the grid must still hold every plane.
    s = scene(pt, rng, 600, n_big=0)
    s["pos"][:, 0] = rng.uniform(-2e4, 2e4, 600)
    s["pos"][:, 1:] = rng.uniform(-1e-3, 1e-3, (600, 2))
    s["radius"] = 1e-3
    assert check(pt, s) is not None
    # too many spheres that are far larger than the rest: no hierarchy, the scan is used
    s = scene(pt, rng, 400, n_big=80)
    assert pt.debug_build_bvh(s) is None
    assert pt.debug_build_bvh(scene(pt, rng, 3, n_big=0)) is None
    assert pt.debug_build_bvh(np.zeros(0, pt.SPHERE_DTYPE)) is None
