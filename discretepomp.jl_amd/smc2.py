"""SMC^2 (particle-filter IBIS, Chopin et al. 2013): host driver over the batched device filter.

Mirror of ``run_pibis`` (src/hmm_ibis.jl:12-135) and its public wrappers ``run_smc2_analysis``
(src/DiscretePOMP.jl:251-257) / ``run_ibis_analysis`` (:289-302).  The reference loops over the
theta-particles and calls ``partial_log_likelihood!`` once per particle; here every such loop is
ONE call on a bank of device-resident populations (one segment per theta-particle):

    hmm_ibis.jl:53-56   incremental weights      -> bank.partial_log_likelihood(pop, theta, obs_min, obs_i)
    hmm_ibis.jl:70-79   theta-level resampling   -> rs_systematic on the host, pop.gather(nidx) on the device
    hmm_ibis.jl:83-116  PMMH mutation            -> all proposals drawn first, their filters run as one batch
                                                    on a second bank (pop_f), accepted ones installed with pop.take

Quirks kept from the reference: independent proposals are accepted with ``exp(aw_f - aw[p])`` and no
proposal-density correction (:104); the "standard" evidence update after ``w .*= gx`` uses the already
updated weights (:121); the jump scalar moves by ``alpha`` / ``0.999`` per accept / reject (:110-112).
One deliberate difference: with conditional proposals (``ind_prop = false``) the reference adapts ``tj``
*inside* the sweep over particles, so proposal p depends on the fate of proposal p-1; here a sweep's
proposals are drawn with the ``tj`` of its start and the accumulated factor is applied afterwards.
"""
from __future__ import annotations

import math
import os
import time
from dataclasses import dataclass
from typing import Optional

import numpy as np

from .model import DPOMPModel
from .particle_filter import C_DF_ESS_CRIT, C_DF_PF_P, ParticleFilter

# defaults of the reference (src/DiscretePOMP.jl:47-52)
C_DF_SMC2_P = 4000
C_ACCEPTANCE_ALPHA = 1.002
C_ALG_NM_SMC2 = "SMC2"


@dataclass
class ImportanceSample:
    """src/cmn_structs.jl:34-41."""
    mu: np.ndarray
    cv: np.ndarray
    theta: np.ndarray    # K x N
    weight: np.ndarray
    run_time: int        # ns
    bme: np.ndarray      # [-ln evidence (standard), -ln evidence (Rao-Blackwellised)]
    acceptance_rate: float = float("nan")
    n_resample: int = 0
    status: int = 0      # OR of the filter status bits of every call (1 = event cap hit, 2 = all weights zero)
    migration: Optional[dict] = None  # distributed runs: populations this rank received over all resamplings, block size


def compute_ess(w: np.ndarray) -> float:
    """src/hmm_particle_filter.jl:4-6."""
    return float(np.sum(w) ** 2 / np.sum(w ** 2))


def compute_is_mu_covar(theta: np.ndarray, w: np.ndarray):
    """Weighted mean and covariance, src/cmn.jl:91-99 (normalised by sum(w), no bias correction)."""
    sw = np.sum(w)
    mu = theta @ w / sw
    d = theta - mu[:, None]
    cv = (d * w) @ d.T / sw
    return mu, cv


def rs_systematic(w: np.ndarray, rng: np.random.Generator) -> np.ndarray:
    """theta-level systematic resampling, src/hmm_resample.jl:44-62; 0-based ancestor indices."""
    n = len(w)
    cw = np.cumsum(w)
    u = (rng.random() / n + np.arange(n) / n) * cw[-1]
    out = np.empty(n, dtype=np.int64)
    j = 0
    for i in range(n):
        while u[i] > cw[j] and j < n - 1:
            j += 1
        out[i] = j
    return out


class _Proposal:
    """``get_prop_density`` / ``get_mv_param`` (src/hmm_cmn.jl:13-18, :33-42): zero-mean MvNormal kept as its
    Cholesky factor; a non-positive-definite update keeps the old density."""

    def __init__(self, k: int):
        self.chol = np.eye(k)

    def update(self, cv: np.ndarray):
        try:
            self.chol = np.linalg.cholesky((cv + cv.T) / 2)
        except np.linalg.LinAlgError:
            pass  # "particle degeneracy problem": keep the old proposal

    def draw(self, rng: np.random.Generator, scale: float, centre: np.ndarray, n: int) -> np.ndarray:
        z = rng.standard_normal((len(self.chol), n))
        return centre + scale * (self.chol @ z)

    def logpdf_centred(self, d: np.ndarray) -> np.ndarray:
        """log N(d | 0, Sigma) up to the normalising constant, column-wise."""
        z = np.linalg.solve(self.chol, d)
        return -0.5 * np.sum(z * z, axis=0)


def _log(x: float) -> float:
    """Julia's log: -Inf at 0 (math.log raises there), e.g. when every incremental weight underflowed."""
    return math.log(x) if x > 0.0 else (-math.inf if x == 0.0 else math.nan)


def run_pibis(bank, prior, theta: np.ndarray, ess_rs_crit: float, ind_prop: bool, alpha: float, npf: int,
              n_props: int = 1, seed: int = 1, rng: Optional[np.random.Generator] = None, verbose: bool = True,
              proposal_correction: bool = False, distributed=False, group=None) -> ImportanceSample:
    """SMC^2 over the theta-particles ``theta`` (K x N).  ``bank`` is a :class:`ParticleFilter` (or any
    object with ``alloc_population``, ``partial_log_likelihood``, ``n_obs``, ``desc.obs_id``).

    ``proposal_correction`` (not in the reference): with independent proposals the reference accepts with
    ``exp(aw_f - aw[p])`` alone (src/hmm_ibis.jl:104), which leaves out ``q(theta)/q(theta_f)`` and biases
    the sample towards the proposal mean (its own test value of -ln evidence is 19.98 where the exact one is
    20.1676).  ``False`` reproduces the reference; ``True`` adds the missing ratio.

    ``distributed``: ``True`` shards the theta-particles over the process group (:class:`parallel.ShardedBank`);
    ``"loopback"`` additionally sends populations whose ancestor is local through the collective (test hook: the
    all-to-all then carries data even with one rank)."""
    start = time.perf_counter_ns()
    rng = rng or np.random.default_rng(seed)
    theta = np.array(theta, dtype=np.float64, order="C")
    K, N = theta.shape
    ess_crit = ess_rs_crit * N
    obs_id = np.asarray(bank.desc.obs_id)
    T = bank.n_obs

    w = np.ones(N)
    aw = np.array([prior.logpdf(theta[:, i]) for i in range(N)])
    from .parallel import LocalBank, ShardedBank
    if getattr(bank, "ess_frac", 0.0) > 0.0:
        # theta-level copies (gather / take) move states only, not the log-weights an ESS-adaptive filter carries
        raise ValueError("run_pibis needs a filter that resamples at every observation (ess_adaptive=False)")
    sb = ShardedBank(bank, N, npf, group, loopback=(distributed == "loopback")) if distributed else LocalBank(bank, N, npf)
    try:
        mu, cv = np.zeros(K), np.zeros((K, K))
        k_log = [0, 0]
        bme = np.zeros(2)
        propd = _Proposal(K)
        tj = 0.2
        call = [0]
        n_resample = 0

        def next_seed():  # every filter call gets its own stream key (the reference advances a global RNG)
            call[0] += 1
            return (int(seed) * 0x9E3779B97F4A7C15 + call[0]) & (2 ** 64 - 1)

        obs_min = 1
        for obs_i in range(1, T + 1):
            if obs_id[obs_i - 1] <= 0:
                continue
            gx = sb.increment(theta, obs_min, obs_i, next_seed())  # :53-56
            aw += gx
            gx = np.exp(gx)
            lml = _log(float(np.sum(w * gx) / np.sum(w)))  # :59
            bme[0] += lml
            w *= gx
            mu, cv = compute_is_mu_covar(theta, w)
            if compute_ess(w) < ess_crit:  # :65
                n_resample += 1
                propd.update(cv)
                nidx = rs_systematic(w, rng)  # :70
                theta = np.ascontiguousarray(theta[:, nidx])
                aw = aw[nidx]
                sb.resample(nidx)  # :74
                mlr = float(np.mean(gx[nidx])) * (math.exp(lml) if lml > -math.inf else 0.0)
                k_log[0] += N
                mtd_gx = gx[nidx].copy()
                for _ in range(n_props):  # :85
                    centre = mu[:, None] if ind_prop else theta
                    theta_f = propd.draw(rng, 1.0 if ind_prop else tj, centre, N)  # :87
                    prtf = np.array([prior.logpdf(theta_f[:, p]) for p in range(N)])
                    ok = np.isfinite(prtf)
                    if not ok.any():
                        continue
                    th_run = np.where(ok[None, :], theta_f, theta)  # invalid proposals run a harmless theta, then are ignored
                    if obs_i == 1:  # :92-94
                        gx_f = sb.propose(th_run, 1, 1, next_seed())
                        aw_f = gx_f.copy()
                    else:  # :95-100
                        aw_f = sb.propose(th_run, 1, obs_i - 1, next_seed())
                        gx_f = sb.propose(th_run, obs_i, obs_i, next_seed())
                        aw_f = aw_f + gx_f
                    aw_f = aw_f + prtf
                    log_ratio = aw_f - aw
                    if proposal_correction and ind_prop:
                        log_ratio = log_ratio + propd.logpdf_centred(theta - mu[:, None]) - propd.logpdf_centred(theta_f - mu[:, None])
                    u = rng.random(N)
                    with np.errstate(over="ignore", invalid="ignore"):
                        accept = ok & (np.exp(log_ratio) > u)  # :104
                    accept &= ~np.isnan(aw_f)
                    if accept.any():
                        mtd_gx[accept] = np.exp(gx_f[accept])
                        theta[:, accept] = theta_f[:, accept]
                        aw[accept] = aw_f[accept]
                        sb.accept(accept)  # pop[p] .= pop_f  :109
                    n_acc = int(accept.sum())
                    k_log[1] += n_acc
                    tj *= alpha ** n_acc * 0.999 ** int(ok.sum() - n_acc)  # :111-113, applied per sweep
                bme[1] += _log(mlr / float(np.mean(mtd_gx))) if np.mean(mtd_gx) > 0 else math.nan  # :118
                w[:] = 1.0
            else:
                bme[1] += _log(float(np.sum(w * gx) / np.sum(w)))  # :121 (with the updated w, as in the reference)
            obs_min = obs_i + 1
        mu, cv = compute_is_mu_covar(theta, w)
        out = ImportanceSample(mu, cv, theta, w, time.perf_counter_ns() - start, -bme,
                               acceptance_rate=(k_log[1] / k_log[0]) if k_log[0] else float("nan"), n_resample=n_resample,
                               status=int(getattr(bank, "status_accum", 0)),
                           migration={"rows_received": sb.total_rows_received, "block": sb.hi - sb.lo} if distributed else None)
        if verbose:
            ar = f"{100.0 * out.acceptance_rate:.3g}" if k_log[0] else "n/a"
            print(f"- finished in {out.run_time / 1e9:.1f} seconds (AR = {ar}%)")
    finally:
        sb.close()  # frees both device populations, also when a filter call raised
    return out


def run_smc2_analysis(model: DPOMPModel, obs_data, np_: int = C_DF_SMC2_P, npf: int = C_DF_PF_P,
                      ess_rs_crit: float = C_DF_ESS_CRIT, ind_prop: bool = True, alpha: float = C_ACCEPTANCE_ALPHA,
                      seed: int = 1, device: Optional[int] = None, verbose: bool = True, proposal_correction: bool = False,
                      distributed: bool = False) -> ImportanceSample:
    """``run_smc2_analysis(model, obs_data; np, npf, ess_rs_crit, ind_prop, alpha)`` (src/DiscretePOMP.jl:251-257).
    (``np_`` because ``np`` is numpy here.)"""
    rng = np.random.default_rng(seed)
    if device is None:  # one process per GPU: a rank of a distributed run drives the GPU of its LOCAL_RANK
        device = int(os.environ.get("LOCAL_RANK", "0")) if distributed else 0
    bank = ParticleFilter(model, obs_data, device=device)
    theta_init = model.prior.rand(rng, np_)  # rand(mdl.prior, np)
    if verbose:
        print(f"Running: {np_}-particle SMC^2 analysis (model: {model.model_name})")
    try:
        return run_pibis(bank, model.prior, theta_init, ess_rs_crit, ind_prop, alpha, npf, seed=seed, rng=rng,
                         verbose=verbose, proposal_correction=proposal_correction, distributed=distributed)
    finally:
        bank.close()


def run_ibis_analysis(model: DPOMPModel, obs_data, algorithm: str = C_ALG_NM_SMC2, np_: Optional[int] = None,
                      ind_prop: Optional[bool] = None, ess_rs_crit: Optional[float] = None,
                      alpha: float = C_ACCEPTANCE_ALPHA, npf: int = C_DF_PF_P, seed: int = 1, device: Optional[int] = None,
                      verbose: bool = True, proposal_correction: bool = False, distributed: bool = False
                      ) -> ImportanceSample:
    """``run_ibis_analysis`` (src/DiscretePOMP.jl:289-302).  Only the default algorithm, SMC^2, sits on
    the particle-filter path; "MBPI" (model-based-proposal IBIS) does not use the filter and is out of scope."""
    if algorithm != C_ALG_NM_SMC2:
        raise NotImplementedError("only algorithm='SMC2' uses the particle filter; MBP-IBIS is outside this path")
    return run_smc2_analysis(model, obs_data, np_ if np_ is not None else C_DF_SMC2_P, npf,
                             ess_rs_crit if ess_rs_crit is not None else C_DF_ESS_CRIT,
                             True if ind_prop is None else ind_prop, alpha, seed, device, verbose, proposal_correction,
                             distributed)
