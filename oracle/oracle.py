"""ctypes binding of oracle/liboracle_dpomp.so -- TEST INFRASTRUCTURE, not product code.

The oracle consumes the public ``dpomp_model_desc`` POD (include/dpomp.h); tests build it with
the product's host-side ``ModelDescriptor`` so both sides see the same model and data.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle_dpomp.so")

MODE_SPEC, MODE_FAITHFUL = 0, 1
DOMAIN_PROPAGATE, DOMAIN_RESAMPLE, DOMAIN_SIMULATE = 0, 1, 2


class Trace(C.Structure):
    _fields_ = [("per_obs_ll", C.POINTER(C.c_double)), ("logw", C.POINTER(C.c_double)),
                ("offspring_end", C.POINTER(C.c_int64)), ("ancestors", C.POINTER(C.c_int64)),
                ("states", C.POINTER(C.c_int64)), ("nkeep", C.c_int64), ("particle_events", C.POINTER(C.c_int32))]


def build(force: bool = False):
    src = os.path.join(HERE, "dpomp_oracle.c")
    deps = [src, os.path.join(HERE, "..", "include", "dpomp.h"), os.path.join(HERE, "orc_log_table.h")]
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["make", "-C", HERE, "-B", "liboracle_dpomp.so"], stdout=subprocess.DEVNULL)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        dp = C.POINTER(C.c_double)
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.orc_uniforms.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int64, dp]
        L.orc_dlog.restype = C.c_double
        L.orc_dlog.argtypes = [C.c_double]
        L.orc_dexp.restype = C.c_double
        L.orc_dexp.argtypes = [C.c_double]
        L.orc_weight_shift.argtypes = [C.c_int64]
        L.orc_fixed_weight.restype = C.c_uint64
        L.orc_fixed_weight.argtypes = [C.c_double, C.c_double, C.c_int]
        L.orc_offspring_end.restype = C.c_int64
        L.orc_offspring_end.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.c_double]
        L.orc_rates.argtypes = [C.c_void_p, dp, C.POINTER(C.c_int64), dp]
        L.orc_choose_event.argtypes = [dp, C.c_int, C.c_double]
        L.orc_obs_loglik.restype = C.c_double
        L.orc_obs_loglik.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        L.orc_pf_partial.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int64, dp, C.c_int, C.c_int, C.c_uint64,
                                     C.c_uint32, C.c_int, C.c_int, C.c_int, dp, C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint64), C.POINTER(Trace)]
        L.orc_pf_partial_adaptive.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int64, dp, C.c_int, C.c_int, C.c_uint64,
                                              C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(Carry), dp,
                                              C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(Trace)]
        L.orc_ess_resample.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, C.c_int64]
        L.orc_pf_loglik.argtypes = [C.c_void_p, C.c_int64, dp, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int, dp,
                                    C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(Trace)]
        L.orc_simulate_events.restype = C.c_int64
        L.orc_simulate_events.argtypes = [C.c_void_p, dp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64, dp,
                                          C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
        L.orc_set_user_rates.argtypes = [C.c_void_p]
        L.orc_set_user_obs.argtypes = [C.c_void_p]
        L.orc_simulate.argtypes = [C.c_void_p, C.c_int64, dp, C.c_uint64, C.c_uint32, C.POINTER(C.c_int64)]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return list(o)


def sfc64_raw(a, b, c, counter, n):
    """n raw outputs of the stream generator from an explicit state (a, b, c, counter)."""
    st = (C.c_uint64 * 4)(a, b, c, counter)
    f = lib().orc_stream_next
    f.restype = C.c_uint64
    f.argtypes = [C.POINTER(C.c_uint64)]
    return [int(f(st)) for _ in range(n)]


def uniforms(seed, seg, p, k, domain, n):
    out = np.empty(n)
    lib().orc_uniforms(seed, seg, p, k, domain, n, _dp(out))
    return out


def dlog(x):
    f = lib().orc_dlog
    return np.array([f(float(v)) for v in np.atleast_1d(x)])


def dexp(x):
    f = lib().orc_dexp
    return np.array([f(float(v)) for v in np.atleast_1d(x)])


def rates(desc, theta, x):
    out = np.empty(desc.n_events)
    th = np.ascontiguousarray(theta, dtype=np.float64)
    xx = np.ascontiguousarray(x, dtype=np.int64)
    lib().orc_rates(C.addressof(desc.c), _dp(th), _ip(xx), _dp(out))
    return out


def choose_event(cum, u):
    c = np.ascontiguousarray(cum, dtype=np.float64)
    return lib().orc_choose_event(_dp(c), len(c), float(u))


def obs_loglik(desc, k, x):
    xx = np.ascontiguousarray(x, dtype=np.int64)
    return lib().orc_obs_loglik(C.addressof(desc.c), int(k), _ip(xx))


class PFResult:
    pass


class Carry(C.Structure):
    _fields_ = [("active", C.c_int), ("l_prev", C.c_double), ("a", C.POINTER(C.c_double))]


class CarryState:
    """Carried log-weights of the ESS-adaptive mode between incremental calls."""

    def __init__(self, P):
        self.a = np.zeros(P)
        self.c = Carry(0, 0.0, _dp(self.a))


def pf_partial(desc, pop, theta, ymin, ymax, seed=1, seg=0, mode=MODE_SPEC, rs_type=1, nthreads=1, trace=False,
               nkeep=64, ess_frac=0.0, carry=None):
    """partial_log_likelihood! on a P x S int64 population (modified in place; pass a
    Fortran-ordered or (S,P)-transposed array: memory layout must be column-major P x S)."""
    P = pop.shape[0]
    S, T = desc.n_states, desc.n_obs
    flat = np.ascontiguousarray(pop.T, dtype=np.int64)  # (S, P) C-order == P x S column-major
    th = np.ascontiguousarray(theta, dtype=np.float64)
    ll = C.c_double()
    st = C.c_uint32()
    ev = C.c_uint64()
    r = PFResult()
    tr = None
    if trace:
        r.per_obs_ll = np.zeros(T)
        r.logw = np.zeros(P)
        r.offspring_end = np.zeros(P, dtype=np.int64)
        r.ancestors = np.zeros(P, dtype=np.int64)
        r.states = np.zeros((T, min(nkeep, P), S), dtype=np.int64)
        r.particle_events = np.zeros((T, P), dtype=np.int32)
        tr = Trace(_dp(r.per_obs_ll), _dp(r.logw), _ip(r.offspring_end), _ip(r.ancestors), _ip(r.states), min(nkeep, P),
                   r.particle_events.ctypes.data_as(C.POINTER(C.c_int32)))
    if ess_frac > 0.0 and carry is None:
        carry = CarryState(P)
    rc = lib().orc_pf_partial_adaptive(C.addressof(desc.c), _ip(flat), P, _dp(th), ymin, ymax, seed, seg, mode, rs_type,
                                       nthreads, float(ess_frac), C.byref(carry.c) if carry is not None else None,
                                       C.byref(ll), C.byref(st), C.byref(ev), C.byref(tr) if tr is not None else None)
    if rc != 0:
        raise RuntimeError(f"orc_pf_partial failed: {rc}")
    pop[:, :] = flat.T
    r.ll, r.status, r.events = ll.value, st.value, ev.value
    r.carry = carry
    return r


def pf_loglik(desc, P, theta, seed=1, seg=0, mode=MODE_SPEC, rs_type=1, nthreads=1, trace=False, nkeep=64, ess_frac=0.0):
    pop = np.zeros((P, desc.n_states), dtype=np.int64)
    r = pf_partial(desc, pop, theta, 1, desc.n_obs, seed, seg, mode, rs_type, nthreads, trace, nkeep, ess_frac)
    r.pop = pop
    return r


def simulate(desc, P, theta, seed=1, seg=0):
    out = np.empty((desc.n_obs, desc.n_states, P), dtype=np.int64)
    th = np.ascontiguousarray(theta, dtype=np.float64)
    lib().orc_simulate(C.addressof(desc.c), P, _dp(th), seed, seg, _ip(out))
    return out


def simulate_events(desc, theta, seed=1, seg=0, p=0, max_events=196000):
    th = np.ascontiguousarray(theta, dtype=np.float64)
    times = np.empty(max_events)
    types = np.empty(max_events, dtype=np.int32)
    states = np.empty((desc.n_obs, desc.n_states), dtype=np.int64)
    n = lib().orc_simulate_events(C.addressof(desc.c), _dp(th), seed, seg, p, max_events, _dp(times),
                                  types.ctypes.data_as(C.POINTER(C.c_int32)), _ip(states))
    m = max_events if n < 0 else n
    return n, times[:m].copy(), types[:m].copy(), states


_user_libs = []


def set_user_rates_from_source(source: str, workdir: str):
    """Compile a rate source (the text handed to dpomp_create_with_source) for the host and register it as the
    oracle's rate function for DPOMP_BUILTIN_USER models."""
    src = os.path.join(workdir, "user_rates.cpp")
    so = os.path.join(workdir, f"libuser_rates_{len(_user_libs)}.so")
    has_obs = "dpomp_user_obs" in source
    with open(src, "w") as f:
        f.write('#include <cmath>\n#define __device__\n#define __forceinline__ inline\n' + source +
                '\nextern "C" void orc_user_rates_entry(const double* th, const int* x, double* r) { dpomp_user_rates(th, x, r); }\n' +
                ('extern "C" double orc_user_obs_entry(const double* th, const int* x, const long long* y, int ny) '
                 '{ return dpomp_user_obs(th, x, y, ny); }\n' if has_obs else ''))
    fma = ["-mfma"] if " fma" in open("/proc/cpuinfo").read() else []
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", *fma, "-fPIC", "-shared", "-o", so, src])
    ul = C.CDLL(so)
    _user_libs.append(ul)
    lib().orc_set_user_rates(C.cast(ul.orc_user_rates_entry, C.c_void_p))
    lib().orc_set_user_obs(C.cast(ul.orc_user_obs_entry, C.c_void_p) if has_obs else None)


def max_threads():
    return lib().orc_max_threads()
