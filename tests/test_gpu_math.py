"""Device arithmetic against the host: the short correctly rounded sqrt / reciprocal / division
sequences and the deterministic sin/cos must be bit-identical to IEEE fp32 on the host (numpy)
and to the oracle's sin/cos, because every later bounce depends on the last bit."""
import ctypes

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tracer(pt):
    g = load_golden("demo_32x32_d0_it1")
    tr = pt.Pathtracer(g["camera_pod"], 1, 1)
    yield tr
    tr.Release()


def _interesting_floats(rng, n):
    bits = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32).copy()
    special = np.array([0.0, -0.0, 1.0, -1.0, 2.0, 0.5, 3.0, 1e-45, 1e-38, 1.17549435e-38, 3.4028235e38,
                        np.inf, -np.inf, np.nan, 2.0**-100, 2.0**100, 2.0**-101, 2.0**99, 1e10, 1e-10,
                        0.99999994, 1.0000001, 16777216.0, 1e5 * 1e5], np.float32)
    x[:special.size] = special
    return x


def same_bits(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    nan = np.isnan(a) & np.isnan(b)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | nan))


def test_sqrt_rcp_div_match_numpy_bitwise(tracer):
    rng = np.random.default_rng(7)
    n = 1 << 22
    x = _interesting_floats(rng, n)
    with np.errstate(all="ignore"):
        assert same_bits(tracer.SelftestEval(0, x), np.sqrt(x))
        assert same_bits(tracer.SelftestEval(1, x), np.float32(1.0) / x)
        y = _interesting_floats(rng, n)[::-1].copy()
        assert same_bits(tracer.SelftestEval(2, x, y), x / y)
        # operands in the range the trace loop produces: magnitudes around 1e-7 .. 1e6
        a = (rng.standard_normal(n) * 10.0 ** rng.uniform(-7, 6, n)).astype(np.float32)
        b = (rng.standard_normal(n) * 10.0 ** rng.uniform(-7, 6, n)).astype(np.float32)
        assert same_bits(tracer.SelftestEval(2, a, b), a / b)
        assert same_bits(tracer.SelftestEval(0, np.abs(a)), np.sqrt(np.abs(a)))
        assert same_bits(tracer.SelftestEval(1, b), np.float32(1.0) / b)
        # concentric-disk quotients: multiples of 2^-23 in [-1, 1] with |num| <= |den|
        num = (rng.integers(-2**23, 2**23, n) * 2.0**-23).astype(np.float32)
        den = (rng.integers(-2**23, 2**23, n) * 2.0**-23).astype(np.float32)
        assert same_bits(tracer.SelftestEval(2, num, den), num / den)


@pytest.mark.parametrize("op", [0, 1, 2, 3])
def test_short_sequences_equal_ieee_expansion_exhaustively(tracer, op):
    """All 2^32 bit patterns (sqrt, 1/x, and 3: the sphere test's square root, whose one-comparison
    gate may let +infinity through to the short sequence, where NaN is as good as +infinity) and 2^32
    random pairs (a/b), compared on the device."""
    assert tracer.SelftestExhaustive(op) == 0


def test_sincos_matches_oracle_bitwise(tracer, O):
    lib = O.load_oracle()
    rng = np.random.default_rng(3)
    # theta = t * (pi/4) for t in [0, 8], as ConcentricSampleDisk produces, plus a denser sample
    t = np.concatenate([rng.uniform(0, 8, 200000), np.arange(9), np.linspace(0, 8, 4097)]).astype(np.float32)
    theta = (t * np.float32(np.float32(3.14159265358979323846) / np.float32(4.0))).astype(np.float32)
    s_dev = tracer.SelftestEval(3, theta)
    c_dev = tracer.SelftestEval(4, theta)
    s, c = ctypes.c_float(), ctypes.c_float()
    s_ref, c_ref = np.empty_like(theta), np.empty_like(theta)
    for i, x in enumerate(theta):
        lib.ptor_sincos(float(x), ctypes.byref(s), ctypes.byref(c))
        s_ref[i], c_ref[i] = s.value, c.value
    assert same_bits(s_dev, s_ref) and same_bits(c_dev, c_ref)


def _material_rows(rng, n):
    """wo, n, u, v: unit normals, directions towards and away from the surface, and samples on the
    lattice u01() produces, with the disk mapping's region boundaries and its origin over-represented."""
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    wo = rng.normal(size=(n, 3)); wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    grazing = rng.random(n) < 0.1     # nearly tangent directions: total internal reflection, tiny cosines
    wo[grazing] -= (np.sum(wo[grazing] * nrm[grazing], axis=1, keepdims=True) * nrm[grazing]) * 0.999
    wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    u = (rng.integers(0, 2**24, n) * 2.0**-24)
    v = (rng.integers(0, 2**24, n) * 2.0**-24)
    k = n // 8
    v[:k] = u[:k]                      # sx == sy
    v[k:2 * k] = 1.0 - u[k:2 * k]      # sx == -sy
    u[2 * k:2 * k + 64] = 0.5          # sx == 0 ...
    v[2 * k + 32:2 * k + 96] = 0.5     # ... sy == 0, both for 32 rows: the origin
    u[2 * k + 96:2 * k + 128] = 0.0
    v[2 * k + 112:2 * k + 144] = 0.0
    rows = np.zeros((n, 8), np.float32)
    rows[:, 0:3], rows[:, 3:6], rows[:, 6], rows[:, 7] = wo, nrm, u, v
    return rows


@pytest.mark.parametrize("mtype,coef", [(0, 0.0), (1, 0.9), (2, 1.5), (2, 2.4), (7, 0.3)])
def test_sample_material_matches_oracle_bitwise(pt, tracer, O, mtype, coef):
    """SampleMaterial as the kernels evaluate it (the staged BSDF of pt_device.hpp) against the
    oracle's statement-by-statement version (scene.h:177-221): f, wi and pdf bit for bit."""
    lib = O.load_oracle()
    rows = _material_rows(np.random.default_rng(11 + mtype), 200000)
    mat = np.zeros(1, pt.MATERIAL_DTYPE)
    mat["color"], mat["coef"], mat["type"] = [[0.75, 0.25, 0.9]], coef, mtype
    dev = tracer.SelftestSampleMaterial(mat, rows)
    ref = np.zeros_like(rows)
    lib.ptor_sample_material_eval.restype = ctypes.c_int32
    lib.ptor_sample_material_eval.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_uint32]
    omat = np.ascontiguousarray(mat, O.MATERIAL_DTYPE)
    assert lib.ptor_sample_material_eval(omat.ctypes.data, rows.ctypes.data, ref.ctypes.data, len(rows)) == 0
    cols = slice(0, 3) if mtype > 2 else slice(0, 7)   # unknown type: only f (black) is defined
    bad = np.argwhere(~((dev[:, cols].view(np.uint32) == ref[:, cols].view(np.uint32))
                        | (np.isnan(dev[:, cols]) & np.isnan(ref[:, cols]))))
    assert bad.size == 0, (bad[:4], rows[bad[0, 0]], dev[bad[0, 0]], ref[bad[0, 0]])
