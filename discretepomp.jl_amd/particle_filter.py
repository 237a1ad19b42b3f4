"""The particle filter operator: host mirror of src/hmm_particle_filter.jl over the C ABI.

* ``get_particle_filter_lpdf``  src/hmm_utils.jl:281-284  -> closure theta -> log p^(y | theta)
* ``get_log_pdf_fn``            src/hmm_particle_filter.jl:87-101
* ``estimate_likelihood``       src/hmm_particle_filter.jl:79-84
* ``partial_log_likelihood``    src/hmm_particle_filter.jl:39-76 (on a device-resident population)

All arithmetic happens in libdpomp_hip.so on the GPU; this module only marshals arguments.
"""
from __future__ import annotations

import ctypes as C
import warnings
import weakref
from typing import Optional, Sequence

import numpy as np

from . import _abi
from .model import DPOMPModel, ModelDescriptor, Observation

# defaults of the reference (src/DiscretePOMP.jl:50-52)
C_DF_PF_P = 200
C_DF_ESS_CRIT = 0.3


class EventCapWarning(RuntimeWarning):
    """A particle reached ``max_events_per_interval``: its weight was zeroed, which the reference never does (its
    filter has no cap and would keep looping), so the returned log-likelihood is biased for those filters."""


class DpompError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"dpomp error {code}: {message}")
        self.code = code


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Population:
    """Device-resident populations of B independent filters (the ``pop::Array{Int64,2}`` that
    ``run_pibis`` keeps per theta-particle, src/hmm_ibis.jl:26-30)."""

    def __init__(self, pf: "ParticleFilter", B: int, P: int):
        self._pf = pf
        self.B, self.P = int(B), int(P)
        self._ptr = C.c_void_p()
        pf._check(pf._lib.dpomp_pop_alloc(pf._h, self.B, self.P, C.byref(self._ptr)))
        pf._pops.add(self)

    def close(self):
        if self._ptr and self._pf._h:
            self._pf._lib.dpomp_pop_free(self._pf._h, self._ptr)
            self._ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def download(self, b: int = 0) -> np.ndarray:
        """Segment ``b`` as a P x S Int64 matrix (rows = particles), like the reference's ``pop``."""
        S = self._pf.n_states
        out = np.empty((S, self.P), dtype=np.int64)  # column-major P x S == C-order S x P
        self._pf._check(self._pf._lib.dpomp_pop_download(self._pf._h, self._ptr, b,
                                                         out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out.T

    def upload(self, b: int, pop: np.ndarray):
        S = self._pf.n_states
        arr = np.ascontiguousarray(np.asarray(pop, dtype=np.int64).T)
        if arr.shape != (S, self.P):
            raise ValueError(f"expected a {self.P} x {S} matrix")
        self._pf._check(self._pf._lib.dpomp_pop_upload(self._pf._h, self._ptr, b,
                                                       arr.ctypes.data_as(C.POINTER(C.c_int64))))

    def gather(self, ancestors: Sequence[int]):
        """``pop[b] <- pop[ancestors[b]]`` (0-based), the theta-level resampling copy of
        src/hmm_ibis.jl:71-75."""
        a = np.ascontiguousarray(ancestors, dtype=np.int64)
        if a.shape != (self.B,):
            raise ValueError("one ancestor per segment")
        self._pf._check(self._pf._lib.dpomp_pop_gather(self._pf._h, self._ptr, a.ctypes.data_as(C.POINTER(C.c_int64))))

    def state_tensor(self, for_write: bool = False):
        """Zero-copy ``torch`` view of the device-resident populations, int32 ``[B, S, stride]`` (lanes
        >= P of the last axis are padding).  Valid until the next resampling / gather on this population."""
        import torch
        ptr, stride = C.c_void_p(), C.c_int64()
        self._pf._check(self._pf._lib.dpomp_pop_device_state(self._pf._h, self._ptr, 1 if for_write else 0,
                                                            C.byref(ptr), C.byref(stride)))

        class _View:
            __cuda_array_interface__ = {"shape": (self.B, self._pf.n_states, int(stride.value)), "typestr": "<i4",
                                        "data": (int(ptr.value), False), "version": 2, "strides": None}
        return torch.as_tensor(_View(), device=torch.device("cuda", self._pf.device))

    def export_state(self):
        """Populations as a tensor for migration between ranks (a view, no copy)."""
        return self.state_tensor(False)

    def import_state(self, t):
        self.state_tensor(True).copy_(t)  # accepts a host tensor too (copied H2D)
        import torch
        torch.cuda.current_stream(self._pf.device).synchronize()

    def export_rows(self, rows):
        """Copy of the populations ``rows`` (segment indices) as an int32 tensor [len(rows), S, stride] on the device:
        what this rank sends when theta-particles migrate."""
        import torch
        idx = torch.as_tensor(np.asarray(rows, dtype=np.int64), device=torch.device("cuda", self._pf.device))
        return self.state_tensor(False).index_select(0, idx)

    def import_rows(self, slots, rows):
        """``pop[slots[i]] <- rows[i]`` (device or host tensor of the shape :meth:`export_rows` returns)."""
        import torch
        dev = torch.device("cuda", self._pf.device)
        idx = torch.as_tensor(np.asarray(slots, dtype=np.int64), device=dev)
        self.state_tensor(True).index_copy_(0, idx, rows.to(dev))
        torch.cuda.current_stream(self._pf.device).synchronize()

    def take(self, other: "Population", take: Sequence[int]):
        """``self[b] <- other[take[b]]`` where ``take[b] >= 0`` (accepted PMMH proposals,
        src/hmm_ibis.jl:109)."""
        t = np.ascontiguousarray(take, dtype=np.int64)
        if t.shape != (self.B,):
            raise ValueError("one entry per segment")
        self._pf._check(self._pf._lib.dpomp_pop_take(self._pf._h, self._ptr, other._ptr, t.ctypes.data_as(C.POINTER(C.c_int64))))

    def debug(self, what: int, b: int = 0) -> np.ndarray:
        out = np.empty(self.P, dtype=np.float64 if what == _abi.DBG_LOGW else np.int32)
        self._pf._check(self._pf._lib.dpomp_debug_download(self._pf._h, self._ptr, what, b, out.ctypes.data_as(C.c_void_p)))
        return out


class ParticleFilter:
    """A model + data set bound to one GPU (``get_private_model``, src/DiscretePOMP.jl:96-99)."""

    def __init__(self, model: DPOMPModel, obs_data: Sequence[Observation], device: int = 0, rs_type: int = 1,
                 essc: float = C_DF_ESS_CRIT, force_generic: bool = False, ess_adaptive: bool = False,
                 on_event_cap: str = "warn", jit: Optional[bool] = None):
        self._lib = _abi.load_library()
        self._h = C.c_void_p()
        self.device = int(device)
        self.model = model
        self.desc = ModelDescriptor(model, obs_data, force_generic=force_generic, jit=jit)
        self.n_states, self.n_params, self.n_obs = self.desc.n_states, self.desc.n_params, self.desc.n_obs
        self.essc = essc  # accepted and unused, as in the reference (src/hmm_particle_filter.jl:63-68)
        # a rate table outside the catalogue is compiled at create time (cached; jit=False: interpret it on the device;
        # None: the library default, on unless DPOMP_GENERIC_JIT=0) -- a flag of this model's descriptor, so the choice
        # touches no process-wide state
        rc = self._create()
        if rc != _abi.OK:
            self._h = C.c_void_p()
            raise DpompError(rc, self._lib.dpomp_last_global_error().decode())
        self._finish_init(rs_type, essc, ess_adaptive, on_event_cap)

    def _create(self) -> int:
        device = self.device
        if self.desc.rate_source is not None:  # arbitrary rate code: compiled with hipRTC into the kernels
            rc = self._lib.dpomp_create_with_source(self.desc.byref(), int(device), self.desc.rate_source.encode(),
                                                    C.byref(self._h))
        else:
            rc = self._lib.dpomp_create(self.desc.byref(), int(device), C.byref(self._h))
        return rc

    def _finish_init(self, rs_type, essc, ess_adaptive, on_event_cap):
        self.rs_type = rs_type
        self._check(self._lib.dpomp_set_resampler(self._h, int(rs_type)))
        self._pops = weakref.WeakSet()  # live populations: freed with the handle if the caller did not
        self.ess_frac = 0.0
        self.last_status: Optional[np.ndarray] = None
        self.status_accum = 0  # OR of the status words of every filter call on this handle (drivers report it)
        if on_event_cap not in ("warn", "raise", "ignore"):
            raise ValueError("on_event_cap must be 'warn', 'raise' or 'ignore'")
        self.on_event_cap = on_event_cap
        if ess_adaptive:  # opt-in: make `essc` do what its name says
            self.set_ess_threshold(essc)

    # -- plumbing ---------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != _abi.OK:
            raise DpompError(rc, self._lib.dpomp_last_error(self._h).decode())

    def close(self):
        if self._h:
            for pop in list(self._pops):
                pop.close()
            self._lib.dpomp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _note_status(self, status: np.ndarray):
        """Keeps the per-segment status words of the call and reacts to DPOMP_ST_EVENT_CAP: a capped particle's
        weight is zero, so the estimate of its filter is not the reference's (states are int32 on the device,
        values stay below 2^30 by the cap on events and the checks at create/upload)."""
        self.last_status = status
        self.status_accum |= int(np.bitwise_or.reduce(status)) if status.size else 0
        if self.on_event_cap != "ignore" and np.any(status & _abi.ST_EVENT_CAP):
            n = int(np.count_nonzero(status & _abi.ST_EVENT_CAP))
            msg = (f"{n} of {status.size} filters had particles stopped at max_events_per_interval "
                   f"(weight zeroed; the reference has no such cap): their log-likelihoods are biased")
            if self.on_event_cap == "raise":
                raise DpompError(_abi.ESTATE, msg)
            warnings.warn(msg, EventCapWarning, stacklevel=3)

    @property
    def evaluator(self) -> int:
        """How rates are evaluated: 1 compiled-in (catalogue), 2 table interpreted on the device, 3 table compiled at
        create time, 4 user rate source."""
        return int(self._lib.dpomp_get_evaluator(self._h))

    def set_stream(self, raw_stream: Optional[int]):
        """Launch on an existing HIP stream, e.g. ``torch.cuda.current_stream().cuda_stream``."""
        self._check(self._lib.dpomp_set_stream(self._h, C.c_void_p(raw_stream or 0)))

    def set_resampler(self, rs_type: int):
        """1 systematic (the reference's working default), 2 stratified, 3 multinomial."""
        self._check(self._lib.dpomp_set_resampler(self._h, int(rs_type)))
        self.rs_type = int(rs_type)

    def set_ess_threshold(self, ess_frac: float):
        """ESS-adaptive resampling (0 = the reference's behaviour: resample at every weighted observation)."""
        self._check(self._lib.dpomp_set_ess_threshold(self._h, float(ess_frac)))
        self.ess_frac = float(ess_frac)

    def set_propagate_variant(self, rpl: int):
        """Propagation-kernel variant (0 default, 1 static, 2/4 LDS queue, 16/32 compaction, 33 compaction with small
        filters laid end to end over the workgroups); never changes a result."""
        self._check(self._lib.dpomp_set_propagate_variant(self._h, int(rpl)))

    def _theta_matrix(self, theta) -> np.ndarray:
        th = np.asarray(theta, dtype=np.float64)
        if th.ndim == 1:
            th = th[:, None]
        if th.shape[0] != self.n_params:
            raise ValueError(f"theta must be a length-{self.n_params} vector or a {self.n_params} x B matrix (column per filter)")
        return np.ascontiguousarray(th.T)  # [B][K] == column-major K x B

    # -- the operator -------------------------------------------------------------------
    def loglik(self, theta, np_particles: int = C_DF_PF_P, seed: int = 1, seg_base: int = 0) -> np.ndarray:
        """``estimate_likelihood`` for every column of ``theta`` (K x B): B independent filters of
        ``np_particles`` particles over all observations."""
        th = self._theta_matrix(theta)
        B = th.shape[0]
        out = np.empty(B, dtype=np.float64)
        status = np.zeros(B, dtype=np.uint32)
        self._check(self._lib.dpomp_pf_loglik(self._h, _dptr(th), B, int(np_particles), int(seed) & (2**64 - 1),
                                              int(seg_base), _dptr(out), status.ctypes.data_as(C.POINTER(C.c_uint32))))
        self._note_status(status)
        return out

    def loglik_static(self, theta, np_particles: int = C_DF_PF_P, seed: int = 1, seg_base: int = 0) -> np.ndarray:
        """:meth:`loglik` through ``dpomp_pf_loglik_static``: same results, but the call allocates no device
        memory (working buffers live in the library's static device array; for callers that must not allocate).
        Raises :class:`DpompError` when the call does not fit that array."""
        th = self._theta_matrix(theta)
        B = th.shape[0]
        out = np.empty(B, dtype=np.float64)
        status = np.zeros(B, dtype=np.uint32)
        self._check(self._lib.dpomp_pf_loglik_static(self._h, _dptr(th), B, int(np_particles), int(seed) & (2**64 - 1),
                                                     int(seg_base), _dptr(out), status.ctypes.data_as(C.POINTER(C.c_uint32))))
        self._note_status(status)
        return out

    def alloc_population(self, B: int, np_particles: int) -> Population:
        return Population(self, B, np_particles)

    def partial_log_likelihood(self, pop: Population, theta, ymin: int, ymax: int, seed: int = 1,
                               seg_base: int = 0) -> np.ndarray:
        """``partial_log_likelihood!(pop, model, theta, fn_rs, essc, ymin, ymax)`` for every segment of
        ``pop`` (1-based ``ymin``/``ymax`` as in the reference)."""
        th = self._theta_matrix(theta)
        if th.shape[0] != pop.B:
            raise ValueError("one theta column per population segment")
        out = np.empty(pop.B, dtype=np.float64)
        status = np.zeros(pop.B, dtype=np.uint32)
        self._check(self._lib.dpomp_pf_partial(self._h, pop._ptr, _dptr(th), int(ymin), int(ymax),
                                               int(seed) & (2**64 - 1), int(seg_base), _dptr(out),
                                               status.ctypes.data_as(C.POINTER(C.c_uint32))))
        self._note_status(status)
        return out

    def simulate(self, theta, n_sims: int = 1, seed: int = 1, seg_base: int = 0) -> np.ndarray:
        """States of ``n_sims`` forward simulations per theta column at every observation time:
        array [B, T, S, n_sims]."""
        th = self._theta_matrix(theta)
        B = th.shape[0]
        out = np.empty((B, self.n_obs, self.n_states, int(n_sims)), dtype=np.int64)
        self._check(self._lib.dpomp_simulate(self._h, _dptr(th), B, int(n_sims), int(seed) & (2**64 - 1), int(seg_base),
                                             out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    def simulate_events(self, theta, n_sims: int = 1, seed: int = 1, seg: int = 0, max_events: int = 196000):
        """Forward simulations with their event histories.  Returns (counts[n], times[n][..], types[n][..],
        states[T, S, n]); ``counts[i] == -1`` marks a trajectory truncated at ``max_events`` (MAX_TRAJ)."""
        th = np.ascontiguousarray(theta, dtype=np.float64)
        if th.shape != (self.n_params,):
            raise ValueError(f"theta must have length {self.n_params}")
        n = int(n_sims)
        counts = np.empty(n, dtype=np.int64)
        times = np.empty((max_events, n), dtype=np.float64)
        types = np.empty((max_events, n), dtype=np.int32)
        states = np.empty((self.n_obs, self.n_states, n), dtype=np.int64)
        self._check(self._lib.dpomp_simulate_events(self._h, _dptr(th), n, int(seed) & (2**64 - 1), int(seg), int(max_events),
                                                    counts.ctypes.data_as(C.POINTER(C.c_int64)), _dptr(times),
                                                    types.ctypes.data_as(C.POINTER(C.c_int32)),
                                                    states.ctypes.data_as(C.POINTER(C.c_int64))))
        m = np.where(counts < 0, max_events, counts)
        return counts, [times[: m[i], i].copy() for i in range(n)], [types[: m[i], i].copy() for i in range(n)], states

    # -- inspection ---------------------------------------------------------------------
    def lane_utilisation(self) -> np.ndarray:
        """Per observation interval: the fraction of lanes a one-particle-per-lane schedule kept busy in the most
        recent call (-1 where unknown); what the automatic propagate variant decides on."""
        out = np.empty(self.n_obs)
        u = C.c_double()
        for k in range(self.n_obs):
            self._check(self._lib.dpomp_get_lane_utilisation(self._h, k + 1, C.byref(u)))
            out[k] = u.value
        return out

    def last_event_count(self) -> int:
        ev = C.c_uint64()
        self._check(self._lib.dpomp_last_event_count(self._h, C.byref(ev)))
        return int(ev.value)

    def set_profiling(self, on):
        """False/0 = off; True/1 = time the kernels of every filter call; n > 1 = of every n-th call."""
        self._check(self._lib.dpomp_set_profiling(self._h, int(on)))

    def null_bracket_us(self) -> float:
        """Median elapsed time of an empty HIP-event bracket (what profiling adds to each kernel's duration)."""
        ms = C.c_double()
        self._check(self._lib.dpomp_get_null_bracket_ms(self._h, C.byref(ms)))
        return float(ms.value) * 1e3

    def kernel_stats(self) -> dict:
        out = {}
        for kid in range(self._lib.dpomp_kernel_count()):
            n, ms = C.c_int64(), C.c_double()
            self._check(self._lib.dpomp_get_kernel_stat(self._h, kid, C.byref(n), C.byref(ms)))
            out[self._lib.dpomp_kernel_name(kid).decode()] = (int(n.value), float(ms.value))
        return out

    def selftest_numerics(self, x: np.ndarray, seed: int = 1):
        x = np.ascontiguousarray(x, dtype=np.float64)
        n = x.shape[0]
        u, lg, ex = np.empty(n), np.empty(n), np.empty(n)
        self._check(self._lib.dpomp_selftest_numerics(self._h, n, int(seed), _dptr(x), _dptr(u), _dptr(lg), _dptr(ex)))
        return u, lg, ex

    def selftest_division(self, a: np.ndarray, b: np.ndarray):
        """Device quotients a / b: (event loop's in-range division, plain division)."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        fast, plain = np.empty(a.shape[0]), np.empty(a.shape[0])
        self._check(self._lib.dpomp_selftest_division(self._h, a.shape[0], _dptr(a), _dptr(b), _dptr(fast), _dptr(plain)))
        return fast, plain


def check_rate_source(model: DPOMPModel, obs_data: Sequence[Observation]) -> str:
    """Compile-only check (needs no GPU) of a :class:`RateSource` model, or of the evaluator a rate-table model is
    turned into at create time.  Returns the compiler log; raises :class:`DpompError` with the diagnostics if the
    source does not compile."""
    lib = _abi.load_library()
    desc = ModelDescriptor(model, obs_data)
    log = C.create_string_buffer(1 << 16)
    rc = lib.dpomp_check_rate_source(desc.byref(), desc.rate_source.encode() if desc.rate_source is not None else None, log, len(log))
    if rc != _abi.OK:
        raise DpompError(rc, log.value.decode(errors="replace") or lib.dpomp_last_global_error().decode())
    return log.value.decode(errors="replace")


def get_particle_filter_lpdf(model: DPOMPModel, obs_data: Sequence[Observation], np: int = C_DF_PF_P,
                             rs_type: int = 1, essc: float = C_DF_ESS_CRIT, seed: int = 1, device: int = 0,
                             ess_adaptive: bool = False):
    """Generate a SMC log-likelihood estimating function for ``model`` -- a.k.a. a particle filter
    (src/hmm_utils.jl:281-284, src/hmm_particle_filter.jl:87-101).

    Returns ``comp_log_pdf(parameters) -> float``.  The reference advances Julia's global RNG on
    every call; here call ``n`` of the closure uses the stream ``(seed, n)``, or pass
    ``comp_log_pdf(parameters, seed=...)`` to pin it.  ``rs_type``: 1 systematic (default), 2 stratified,
    3 multinomial.  ``essc`` is accepted and ignored as in the reference unless ``ess_adaptive=True``, which turns
    it into a real ESS threshold (resample only when ESS < essc * np).
    """
    pf = ParticleFilter(model, obs_data, device=device, rs_type=rs_type, essc=essc, ess_adaptive=ess_adaptive)
    n_particles = int(np)
    state = {"calls": 0}

    def comp_log_pdf(parameters, seed: Optional[int] = None) -> float:
        if seed is None:
            s = (int(comp_log_pdf.base_seed) * 0x9E3779B97F4A7C15 + state["calls"]) & (2**64 - 1)
            state["calls"] += 1
        else:
            s = int(seed)
        return float(pf.loglik(parameters, n_particles, seed=s)[0])

    comp_log_pdf.base_seed = seed
    comp_log_pdf.filter = pf
    return comp_log_pdf
