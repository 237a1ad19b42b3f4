"""dpomp_pf_loglik_static (the entry point that allocates nothing) against dpomp_pf_loglik, through the C ABI.

Bar: the log-likelihoods and status words are equal bit for bit -- both run the same kernels, one on allocated
buffers, the other on buffers carved from the library's static device array.  The array keeps the previous call's
contents, so the calls come in a row and the second is smaller than the first.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(pf, theta, P, seed, seg_base=0):
    """(allocating entry point, static entry point): log-likelihoods and status words of the same call."""
    lib, th = pf._lib, pf._theta_matrix(theta)
    B = th.shape[0]
    out = []
    for fn in (lib.dpomp_pf_loglik, lib.dpomp_pf_loglik_static):
        ll, st = np.full(B, np.nan), np.full(B, 0xFFFFFFFF, dtype=np.uint32)
        rc = fn(pf._h, th.ctypes.data_as(C.POINTER(C.c_double)), B, P, seed, seg_base,
                ll.ctypes.data_as(C.POINTER(C.c_double)), st.ctypes.data_as(C.POINTER(C.c_uint32)))
        assert rc == 0, lib.dpomp_last_error(pf._h).decode()
        out.append((ll, st, pf.last_event_count()))
    return out


def _same(a, b):
    assert a[0].tobytes() == b[0].tobytes(), (a[0], b[0])  # bit for bit, -inf included
    assert np.array_equal(a[1], b[1])
    assert a[2] == b[2]


def test_static_entry_point_equals_allocating_one_twice_in_a_row(dp, pooley):
    pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), pooley)
    theta = np.array([[0.003, 0.0032, 0.0025], [0.1, 0.1, 0.08]])
    a, s = _pair(pf, theta, 50000, seed=7)        # 3 filters x 50000 particles: multi-kernel path, 49 tiles each
    _same(a, s)
    assert np.all(np.isfinite(s[0]))
    a2, s2 = _pair(pf, theta[:, :2], 3000, seed=8)  # smaller call on the same array (stale contents behind it)
    _same(a2, s2)
    a3, s3 = _pair(pf, theta[:, :1], 200, seed=9)   # whole-filter-in-one-workgroup path
    _same(a3, s3)
    a4, s4 = _pair(pf, theta, 50000, seed=7)        # and the first call again: identical to itself
    _same(a4, s4)
    assert a4[0].tobytes() == a[0].tobytes()
    pf.close()


@pytest.mark.parametrize("rs_type,ess", [(2, 0.0), (3, 0.0), (1, 0.5), (2, 0.5)])
def test_static_entry_point_other_resamplers_and_ess_mode(dp, pooley, rs_type, ess):
    pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), pooley, rs_type=rs_type)
    if ess:
        pf.set_ess_threshold(ess)
    theta = np.array([[0.003, 0.004], [0.1, 0.15]])
    _same(*_pair(pf, theta, 20000, seed=3))
    _same(*_pair(pf, theta[:, :1], 5000, seed=4))
    pf.close()


def test_static_entry_point_seir_and_two_handles_share_the_array(dp, make_obs):
    """Handles of different models use the same per-GPU array one after the other."""
    import os
    from conftest import GOLDEN
    seir_obs = dp.get_observations(os.path.join(GOLDEN, "seir_synthetic.csv"))
    seir = dp.ParticleFilter(dp.generate_model("SEIR", [100, 0, 1, 0]), seir_obs)
    sis = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), dp.get_observations(os.path.join(GOLDEN, "pooley.csv")))
    th_seir = np.array([[0.005, 0.004], [0.2, 0.25], [0.1, 0.1]])
    _same(*_pair(seir, th_seir, 65536, seed=5))
    _same(*_pair(sis, np.array([[0.003], [0.1]]), 4096, seed=6))
    _same(*_pair(seir, th_seir[:, :1], 1000, seed=5))
    seir.close()
    sis.close()


def test_static_entry_point_rejects_what_does_not_fit(dp, pooley):
    pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), pooley)
    lib = pf._lib
    th = pf._theta_matrix([0.003, 0.1])
    ll = np.zeros(1)

    def call(B, P, theta=th):
        return lib.dpomp_pf_loglik_static(pf._h, theta.ctypes.data_as(C.POINTER(C.c_double)), B, P, 1, 0,
                                          ll.ctypes.data_as(C.POINTER(C.c_double)), None)
    assert call(1, 1 << 20) == dp._abi.EINVAL      # 36 B per particle x 2^20 > 16 MiB
    msg = lib.dpomp_last_error(pf._h).decode()
    assert "static scratch" in msg and "dpomp_pf_loglik" in msg, msg
    big = np.ascontiguousarray(np.tile(th, (dp._abi.STATIC_MAX_SEGMENTS + 1, 1)))
    ll = np.zeros(dp._abi.STATIC_MAX_SEGMENTS + 1)
    assert call(dp._abi.STATIC_MAX_SEGMENTS + 1, 8, big) == dp._abi.EINVAL
    assert "maximum" in lib.dpomp_last_error(pf._h).decode()
    # the largest SIS call that fits (36 B per particle + control block) still runs and agrees
    _same(*_pair(pf, [0.003, 0.1], 460000, seed=2))
    pf.close()


def test_static_entry_point_makes_no_allocating_call(dp, pooley):
    """The property the entry point exists for: across its calls the library makes no hipMalloc / hipHostMalloc /
    hipEventCreate / hipStreamCreate (every such call site of the library bumps dpomp_debug_alloc_count), and the free
    device memory does not move -- whatever the shape, resampler or mode, first call of a shape included.  The
    allocating entry point does move the counter on the first call of a new shape."""
    import torch
    lib = dp._abi.load_library()
    assert lib.dpomp_static_max_bytes() == dp._abi.STATIC_MAX_BYTES
    handles = []
    for rs_type, ess in [(1, 0.0), (2, 0.0), (3, 0.0), (1, 0.5), (2, 0.5)]:
        pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), pooley, rs_type=rs_type)
        if ess:
            pf.set_ess_threshold(ess)
        handles.append(pf)
    theta = np.array([[0.003, 0.0032, 0.0025], [0.1, 0.1, 0.08]])
    ll = np.zeros(400)
    big = np.ascontiguousarray(np.tile(np.array([0.003, 0.1]), (400, 1)))

    def static_call(pf, th, B, P, seed):
        rc = lib.dpomp_pf_loglik_static(pf._h, th.ctypes.data_as(C.POINTER(C.c_double)), B, P, seed, 0,
                                        ll.ctypes.data_as(C.POINTER(C.c_double)), None)
        assert rc == 0, lib.dpomp_last_error(pf._h).decode()

    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    n0 = lib.dpomp_debug_alloc_count()
    for pf in handles:
        th = pf._theta_matrix(theta)
        static_call(pf, th, 3, 30000, 11)       # multi-kernel path, several tiles per filter
        static_call(pf, th, 2, 777, 12)         # a new, smaller shape
        static_call(pf, th, 1, 200, 13)         # whole-filter kernel (systematic, non-adaptive) / one tile
        static_call(pf, big, 300, 64, 14)       # hundreds of small filters in one call (rows are padded to 1024 slots)
        static_call(pf, th, 3, 30000, 15)       # the first shape again
    assert lib.dpomp_debug_alloc_count() == n0, "dpomp_pf_loglik_static made an allocating HIP call"
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0, "free device memory moved across dpomp_pf_loglik_static calls"
    # the hook itself works: the allocating entry point allocates on the first call of a shape, not on the second
    handles[0].loglik(theta, 12345, seed=1)
    n1 = lib.dpomp_debug_alloc_count()
    assert n1 > n0
    handles[0].loglik(theta, 12345, seed=2)
    assert lib.dpomp_debug_alloc_count() == n1
    for pf in handles:
        pf.close()
