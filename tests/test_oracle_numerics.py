"""Pins the oracle's numerical building blocks against independent references."""
import ctypes as C
import math
import os

import mpmath as mp
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_philox4x32_10_known_answers(orc):
    # Random123 kat_vectors, philox4x32 10 rounds
    assert orc.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert orc.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert orc.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def _py_stream(seed, seg, p, k, domain, n):
    """Independent big-integer restatement: Philox seeding + sfc64 (12 discarded outputs) + (2k+1)/2^53."""
    M = (1 << 64) - 1
    c = [p, seg, k, domain]
    key = [seed & 0xFFFFFFFF, seed >> 32]
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ key[0], p1 & 0xFFFFFFFF, (p0 >> 32) ^ c[3] ^ key[1], p0 & 0xFFFFFFFF]
        key = [(key[0] + 0x9E3779B9) & 0xFFFFFFFF, (key[1] + 0xBB67AE85) & 0xFFFFFFFF]
    a, b, cc, ctr = c[0] | (c[1] << 32), c[2] | (c[3] << 32), 0x9E3779B97F4A7C15, 1
    out = []
    for i in range(n + 12):
        t = (a + b + ctr) & M
        ctr += 1
        a = b ^ (b >> 11)
        b = (cc + (cc << 3)) & M
        cc = ((((cc << 24) | (cc >> 40)) & M) + t) & M
        if i >= 12:
            out.append((2 * (t >> 12) + 1) / 2.0 ** 53)
    return np.array(out)


def test_sfc64_matches_numpy_bit_generator(orc):
    # numpy.random.SFC64 is an independent implementation of the published generator: same state -> same outputs
    rng = np.random.default_rng(3)
    for _ in range(5):
        a, b, c = (int(v) for v in rng.integers(0, 1 << 63, 3, dtype=np.uint64))
        bg = np.random.SFC64()
        st = bg.state
        st["state"]["state"] = np.array([a, b, c, 1], dtype=np.uint64)
        bg.state = st
        assert orc.sfc64_raw(a, b, c, 1, 64) == [int(v) for v in bg.random_raw(64)]


def test_stream_matches_big_integer_restatement(orc):
    for args in [(1, 0, 0, 0, 0), (20261003, 7, 12345, 3, 0), ((1 << 64) - 1, 4000000000, 1 << 31, 19, 1)]:
        assert np.array_equal(orc.uniforms(*args, 64), _py_stream(*args, 64))


def test_uniforms_are_open_interval_and_uniform(orc):
    u = np.concatenate([orc.uniforms(5, 0, p, 0, 0, 256) for p in range(400)])
    assert u.min() > 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 4 * math.sqrt(1 / 12 / len(u))
    hist, _ = np.histogram(u, bins=20, range=(0, 1))
    chi2 = ((hist - len(u) / 20) ** 2 / (len(u) / 20)).sum()
    assert chi2 < 60  # 19 dof, p ~ 1e-6
    # streams of neighbouring particles / observations are uncorrelated
    a = np.array([orc.uniforms(5, 0, p, 0, 0, 1)[0] for p in range(4000)])
    assert abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 0.08


def _max_ulp(vals, refs):
    worst = 0.0
    for v, r in zip(vals, refs):
        rf = float(r)
        worst = max(worst, float(abs(mp.mpf(float(v)) - r) / abs(np.spacing(rf))))
    return worst


def test_dlog_within_one_and_a_half_ulp(orc):  # measured 1.43
    mp.mp.prec = 200
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.random(8000), 1 - rng.random(2000) * 2.0 ** -7, 1 + rng.random(1000) * 2.0 ** -7,
                         rng.random(1000) * 1e-10,
                         [2.0 ** -53, 1 - 2.0 ** -53, 0.5, 0.6875, 0.6874999999, 1.375, 1.3749999999, 1048576.0, 3.7e18, 1e-300]])
    assert _max_ulp(orc.dlog(xs), [mp.log(mp.mpf(float(x))) for x in xs]) < 1.5
    assert orc.dlog([1.0])[0] == 0.0


def test_log_table_entries_rederived():
    """Both generated tables (the oracle's and the product's) equal an independent mpmath evaluation
    of their definition: invc = RN(1/c_i), logc = RN(-log(invc)), c = 1 on the subintervals touching 1."""
    import re
    import struct
    mp.mp.prec = 300
    OFF = 0x3FE6000000000000
    val = lambda b: mp.mpf(struct.unpack("<d", struct.pack("<Q", b))[0])
    want = []
    for i in range(512):
        lo, hi = val(OFF + (i << 43)), val(OFF + ((i + 1) << 43))
        if lo == 1 or hi == 1:
            want.append((1.0, 0.0))
        else:
            invc = float(2 / (lo + hi))
            want.append((invc, float(-mp.log(mp.mpf(invc)))))
    for path in ("oracle/orc_log_table.h", "discretepomp.jl_amd/csrc/dpomp_log_table.h"):
        text = open(os.path.join(ROOT, path)).read()
        rows = re.findall(r"\{(-?0x[0-9a-f.]+p[+-]\d+), (-?0x[0-9a-f.]+p[+-]\d+)\}", text)
        got = [(float.fromhex(a), float.fromhex(b)) for a, b in rows]
        assert got == want, path


def test_dexp_within_one_ulp(orc):
    mp.mp.prec = 200
    xs = np.concatenate([-37.0 * np.random.default_rng(1).random(6000), [0.0, -37.0, -1e-300, -0.34657359, -0.3465736]])
    assert _max_ulp(orc.dexp(xs), [mp.exp(mp.mpf(float(x))) for x in xs]) < 1.0
    assert orc.dexp([0.0])[0] == 1.0


def test_fixed_point_weights_and_offspring_boundaries(orc):
    L = orc.lib()
    assert [L.orc_weight_shift(p) for p in (1, 2, 1024, 1025, 1 << 20, (1 << 20) + 1, 1 << 30)] == [52, 52, 52, 51, 42, 41, 32]
    inf = float("inf")
    assert L.orc_fixed_weight(-inf, 0.0, 42) == 0 and L.orc_fixed_weight(float("nan"), 0.0, 42) == 0
    assert L.orc_fixed_weight(-3.0, -3.0, 42) == 1 << 42
    assert L.orc_fixed_weight(-41.0, -3.0, 42) == 0
    # boundaries: monotone, N(0) = 0, N(W) = P, and every slot gets exactly one ancestor
    rng = np.random.default_rng(3)
    for P in (1, 7, 1000):
        w = rng.integers(0, 1 << 40, P).astype(np.uint64)
        w[rng.random(P) < 0.3] = 0
        w[0] = max(int(w[0]), 1)
        c = np.cumsum(w)
        W = int(c[-1])
        for r in (2.0 ** -53, 0.31, 1 - 2.0 ** -53):
            N = np.array([L.orc_offspring_end(int(ci), W, P, r) for ci in c])
            assert L.orc_offspring_end(0, W, P, r) == 0 and N[-1] == P and np.all(np.diff(N) >= 0)
            counts = np.diff(np.concatenate([[0], N]))
            assert counts.sum() == P and np.all(counts[w == 0] == 0)
            # against the reference rule in exact rationals: slot i -> first j with c_j > (i + r) W / P
            from fractions import Fraction
            anc = np.repeat(np.arange(P), counts)
            for i in rng.integers(0, P, 20):
                u = (Fraction(int(i)) + Fraction(r)) * W / P
                j = int(np.searchsorted(c.astype(object), u, side="right"))
                assert abs(int(anc[i]) - j) <= 1  # equal except when u sits within rounding of a boundary
