"""The PRODUCT's numerics header, compiled for the host, equals the oracle's independent
restatement bit for bit (RNG streams, dlog, dexp, fixed-point weights, offspring boundaries).
This is the CPU-side half of the GPU parity argument: same header -> same IEEE operations."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hc():
    L = C.CDLL(os.path.join(ROOT, "discretepomp.jl_amd", "libdpomp_hostcheck.so"))
    L.hc_uniforms.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int64, C.POINTER(C.c_double)]
    for n in ("hc_dlog", "hc_dexp", "hc_ordered_roundtrip"):
        getattr(L, n).restype = C.c_double
        getattr(L, n).argtypes = [C.c_double]
    L.hc_weight_shift.argtypes = [C.c_int64]
    L.hc_fixed_weight.restype = C.c_uint64
    L.hc_fixed_weight.argtypes = [C.c_double, C.c_double, C.c_int]
    L.hc_offspring_end.restype = C.c_int64
    L.hc_offspring_end.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.c_double]
    L.hc_log_mean_weight.restype = C.c_double
    L.hc_log_mean_weight.argtypes = [C.c_double, C.c_uint64, C.c_int, C.c_int64]
    L.hc_ordered_key.restype = C.c_uint64
    L.hc_ordered_key.argtypes = [C.c_double]
    return L


def test_streams_bitwise(orc):
    L = _hc()
    for args in [(1, 0, 0, 0, 0), (20261003, 7, 12345, 3, 0), ((1 << 64) - 1, 4000000000, 1 << 31, 19, 1), (9, 1, 2, 3, 2)]:
        out = np.empty(500)
        L.hc_uniforms(*args, 500, out.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.array_equal(out, orc.uniforms(*args, 500))


def test_dlog_dexp_bitwise(orc):
    L = _hc()
    rng = np.random.default_rng(2)
    xs = np.concatenate([rng.random(20000), 1 - rng.random(2000) * 1e-6, rng.random(2000) * 1e-12, [2.0 ** -53, 1 - 2.0 ** -53, 1.0, 7.0, 1e300]])
    assert np.array_equal(np.array([L.hc_dlog(float(x)) for x in xs]), orc.dlog(xs))
    xe = np.concatenate([-37.0 * rng.random(20000), [0.0, -37.0]])
    assert np.array_equal(np.array([L.hc_dexp(float(x)) for x in xe]), orc.dexp(xe))


def test_weights_and_offspring_bitwise(orc):
    L, O = _hc(), orc.lib()
    rng = np.random.default_rng(4)
    for P in (1, 3, 1024, 1025, 1 << 20, (1 << 20) + 1):
        assert L.hc_weight_shift(P) == O.orc_weight_shift(P)
    g = -40 * rng.random(5000)
    for sh in (42, 52):
        assert [L.hc_fixed_weight(float(v), 0.0, sh) for v in g] == [O.orc_fixed_weight(float(v), 0.0, sh) for v in g]
    for v in (float("-inf"), float("nan")):
        assert L.hc_fixed_weight(v, 0.0, 42) == O.orc_fixed_weight(v, 0.0, 42) == 0
    P = 1 << 20
    W = int(rng.integers(1 << 55, 1 << 60))
    for c in [0, 1, W // 3, W - 1, W] + [int(x) for x in rng.integers(0, W, 2000)]:
        assert L.hc_offspring_end(c, W, P, 0.37) == O.orc_offspring_end(c, W, P, 0.37)


def test_ordered_key_is_monotone_and_invertible():
    L = _hc()
    vals = [-np.inf, -1e300, -3.5, -1e-300, -0.0, 0.0, 1e-300, 2.5, 1e300, np.inf]
    keys = [L.hc_ordered_key(v) for v in vals]
    assert all(a <= b for a, b in zip(keys, keys[1:])) and keys[0] > 0
    for v in vals:
        assert L.hc_ordered_roundtrip(v) == v


def test_log_mean_weight_matches_direct_formula():
    L = _hc()
    import math
    for P, W, sh, m in [(1024, 300 << 52, 52, -2.1), (1 << 20, (1 << 57) + 12345, 42, -1.7)]:
        want = m + math.log(W / 2.0 ** sh / P)
        assert abs(L.hc_log_mean_weight(m, W, sh, P) - want) < 1e-13


def test_ess_decision_bitwise(orc):
    L, O = _hc(), orc.lib()
    L.hc_ess_resample.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, C.c_int64]
    rng = np.random.default_rng(6)
    for _ in range(3000):
        P = int(rng.integers(1, 1 << 20))
        w = rng.integers(0, 1 << 42, size=min(P, 64), dtype=np.uint64)
        W = int(w.sum())
        s2 = sum(int(x) * int(x) for x in w)
        frac = float(rng.random())
        assert L.hc_ess_resample(W, s2 >> 64, s2 & ((1 << 64) - 1), frac, P) == O.orc_ess_resample(W, s2 >> 64, s2 & ((1 << 64) - 1), frac, P)
    assert L.hc_ess_resample(0, 0, 0, 0.5, 10) == 1
