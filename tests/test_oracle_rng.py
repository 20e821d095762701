"""T0 of the parity ladder: the counter-based RNG and the deterministic sin/cos of the oracle.

Philox4x32-10 is pinned by the Random123 known-answer vectors (kat_vectors of Random123 v1.09,
philox4x32 10 rounds); the float map and the sin/cos pair by their definitions.
"""
import ctypes

import numpy as np
import pytest

KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox_py(ctr, key):
    """Independent pure-Python evaluation of Philox4x32-10."""
    c, k = list(ctr), list(key)
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xffffffff, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xffffffff]
        k = [(k[0] + 0x9E3779B9) & 0xffffffff, (k[1] + 0xBB67AE85) & 0xffffffff]
    return tuple(c)


@pytest.mark.parametrize("ctr,key,expect", KAT)
def test_philox_known_answers(O, ctr, key, expect):
    lib = O.load_oracle()
    c, k, out = np.array(ctr, np.uint32), np.array(key, np.uint32), np.zeros(4, np.uint32)
    lib.ptor_philox(c.ctypes.data, k.ctypes.data, out.ctypes.data)
    assert tuple(int(v) for v in out) == expect
    assert philox_py(ctr, key) == expect


def test_uniform_map_and_stream_layout(O):
    lib = O.load_oracle()
    rng = np.random.default_rng(0)
    for _ in range(200):
        seed = int(rng.integers(0, 2**63))
        pixel, it, slot, stream = (int(v) for v in rng.integers(0, 2**32, 4))
        out = np.zeros(4, np.float32)
        lib.ptor_uniforms(seed, pixel, it, slot, stream, out.ctypes.data)
        words = philox_py((pixel, it, slot, stream), (seed & 0xffffffff, seed >> 32))
        expect = np.array([(w >> 8) * 2.0**-24 for w in words], np.float32)
        assert out.tobytes() == expect.tobytes()
        assert (out >= 0).all() and (out < 1).all()


def test_sincos_accuracy_and_quadrants(O):
    lib = O.load_oracle()
    s, c = ctypes.c_float(), ctypes.c_float()
    # ConcentricSampleDisk calls with theta = t*(pi/4), t in [0, 8]
    xs = np.concatenate([np.linspace(0, 2 * np.pi, 20001), np.arange(9) * (np.pi / 4)]).astype(np.float32)
    worst = 0.0
    for x in xs:
        lib.ptor_sincos(float(x), ctypes.byref(s), ctypes.byref(c))
        es, ec = np.sin(np.float64(x)), np.cos(np.float64(x))
        # absolute error in units of 2^-24 (values are <= 1): a few ulp of 1.0
        worst = max(worst, abs(s.value - es) / 2.0**-24, abs(c.value - ec) / 2.0**-24)
        assert abs(s.value * s.value + c.value * c.value - 1.0) < 4e-7
    assert worst < 2.0, worst
    lib.ptor_sincos(0.0, ctypes.byref(s), ctypes.byref(c))
    assert (s.value, c.value) == (0.0, 1.0)
