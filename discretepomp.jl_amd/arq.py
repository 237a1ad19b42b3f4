"""ARQ-MCMC: grid-quantised Metropolis-Hastings with a likelihood cache -- the reference's working
PF-inside-MCMC path (its `arqmcmc` testset pins `imp_sample.mu[1] ~ 0.003217`, test/runtests.jl:30-37).

Mirror of the ARQMCMC sub-module as far as it touches the particle filter:

* ``get_grid_point!``        src/arq_alg_std.jl:4-41   (lazy ``model.pdf(theta)`` = the particle filter, cached
                                                        per grid cell, re-estimated up to ``sample_limit`` times)
* ``arq_met_hastings!``      src/arq_alg_std.jl:44-88  (one chain; refresh after 100 straight rejections :67-74)
* ``get_theta_val``, ``get_theta_f``, ``adapt_jw!``, ``get_initial_sample``, ``@init_inner_mcmc``
                             src/arq_alg_cmn.jl:26-136
* ``run_inner_mcmc_analysis``  src/arq_main.jl:55-87   (chains run one after another and SHARE the cache;
                                                        importance sample over the visited cells, :74-80)
* public wrapper             src/DiscretePOMP.jl:342-350

The chains are inherently sequential (each step needs its predecessor) and on a 20 x 25 grid the whole analysis
needs only a few hundred distinct filter evaluations, each one GPU call through ``get_particle_filter_lpdf``;
everything else is host bookkeeping, as in the reference.  The data-augmented variant (arq_alg_daug.jl) is
commented out in the reference and not mirrored.
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass
from typing import Callable, Dict, Optional, Tuple

import numpy as np

from .pmcmc import RejectionSample, gelman_sre, handle_rej_samples
from .smc2 import ImportanceSample, compute_is_mu_covar

# src/arq_main.jl:6-15, src/arq_alg_cmn.jl:48-51
C_DF_MCMC_STEPS = 50000
C_DF_MCMC_ADAPT = 0.2
C_DF_ARQ_SL = 1
C_DF_ARQ_SR = 50
C_DF_ARQ_MC = 5
C_DF_ARQ_AR = 0.33
C_DF_ARQ_JT = 0.0
Q_JUMP = 0.1
Q_J_MIN = 2
N_ADAPT_PERIODS = 100
C_DF_ARQ_CJ = 10
Q_REJECT_TRIGGER = 100
C_LAR_J_MP = 0.2


@dataclass
class ARQModel:
    """src/arq_structs.jl: target log-density (here: the particle filter) + grid geometry."""
    pdf: Callable
    sample_interval: np.ndarray
    sample_offset: np.ndarray


@dataclass
class GridPoint:
    sample: np.ndarray
    log_likelihood: float
    visited: int
    sampled: int


@dataclass
class ARQMCMCSample:
    imp_sample: ImportanceSample
    samples: RejectionSample
    sample_interval: np.ndarray
    sample_limit: int
    sample_dispersal: int
    adapt_period: int
    sre: np.ndarray
    run_time: int
    fx: np.ndarray
    sample_cache: Dict[Tuple[int, ...], GridPoint]


class _Sampler:
    def __init__(self, model: ARQModel, prior, sample_limit, sample_dispersal, jitter, grid, rng):
        self.m, self.prior, self.limit, self.disp, self.jitter = model, prior, sample_limit, sample_dispersal, jitter
        self.grid, self.rng = grid, rng

    def theta_val(self, idx):  # get_theta_val, arq_alg_cmn.jl:26-32
        out = self.m.sample_offset + np.asarray(idx) * self.m.sample_interval
        if self.jitter > 0.0:
            out = out + (self.rng.random(len(out)) * 2 - 1) * self.jitter * self.m.sample_interval
        return out

    def grid_point(self, idx, burn_in):  # get_grid_point!, arq_alg_std.jl:4-41 -> (GridPoint, prior, computed)
        key = tuple(int(v) for v in idx)
        x = self.grid.get(key)
        if x is not None:
            visited, sampled, val = x.visited, x.sampled, x.sample
        else:
            visited, sampled, val = 0, 0, self.theta_val(idx)
        pr = self.prior(val)
        if pr == -math.inf:
            return GridPoint(val, -math.inf, visited, sampled), pr, False
        if visited < (1 if burn_in else self.limit):
            ll = float(self.m.pdf(val))
            if x is not None:  # running mean of the likelihood estimates of this cell (:26-27)
                ll = math.log(math.exp(x.log_likelihood) + (math.exp(ll) - math.exp(x.log_likelihood)) / visited)
            visited += 1
            comp = True
        else:
            ll, comp = x.log_likelihood, False
        if not burn_in:
            sampled += 1
        out = GridPoint(val, ll, visited, sampled)
        self.grid[key] = out
        return out, pr, comp

    def theta_f(self, idx, j_w, max_dist, min_dist=1):  # get_theta_f, arq_alg_cmn.jl:36-45
        out = np.zeros(len(idx), dtype=np.int64)
        d = max_dist if min_dist == max_dist else int(self.rng.integers(min_dist, max_dist + 1))
        pw = j_w / j_w.sum()
        while np.abs(out).sum() != d:
            p = int(self.rng.choice(len(idx), p=pw))
            out[p] += 1 if self.rng.random() < 0.5 else -1
        return out + idx

    def initial_sample(self, fx, disp=None):  # get_initial_sample, arq_alg_cmn.jl:100-107
        disp = disp or self.disp
        while True:
            idx = self.rng.integers(1, disp + 1, len(self.m.sample_interval))
            x, pr, comp = self.grid_point(idx, True)
            if pr != -math.inf:
                fx[0] += int(comp)
                return idx, (x, pr)
            disp += 1


def _adapt_jw(j_w, lar_j, j, accepted, a_h, i, tgt_ar, mc_idx):  # adapt_jw!, arq_alg_cmn.jl:54-81 (i is 1-based)
    recent = int(accepted[i - a_h:i].sum())
    if j == Q_J_MIN and recent == 0:
        j = int(round(C_DF_ARQ_CJ * (i / a_h))) if int(accepted[:i].sum()) == 1 else lar_j
    else:
        j = max(int(round(j * ((recent / a_h) / tgt_ar))), Q_J_MIN)
    sd = mc_idx[:, :i].std(axis=1, ddof=1)
    if sd.sum() == 0.0:
        sd[:] = 1.0
    else:
        sd[sd == 0.0] = sd[sd > 0.0].min()
    j_w[:] = sd
    return j


def _arq_met_hastings(samples, mc, s: _Sampler, steps, adapt_period, tgt_ar):  # arq_alg_std.jl:44-88
    fx = np.zeros(3, dtype=np.int64)
    theta_i, (xi, pri) = s.initial_sample(fx)
    n = len(theta_i)
    lar_j = int(round(C_LAR_J_MP * s.disp * n))
    a_h = int(max(steps / N_ADAPT_PERIODS, 100))
    j = int(round(Q_JUMP * s.disp * n))
    j_w = np.ones(n)
    mc_idx = np.zeros((n, steps), dtype=np.int64)
    accepted = np.zeros(steps, dtype=bool)
    samples[:, 0, mc] = xi.sample
    mc_idx[:, 0] = theta_i
    accepted[0] = True
    for i in range(1, steps):  # Julia i = 2:steps
        ji = i + 1
        theta_f = s.theta_f(theta_i, j_w, j, 1)
        xf, prf, comp = s.grid_point(theta_f, ji < a_h)
        fx[1] += int(comp)
        with np.errstate(over="ignore", invalid="ignore"):
            mh = math.exp(min(700.0, prf - pri + xf.log_likelihood - xi.log_likelihood)) if prf != -math.inf else 0.0
        if mh > 1.0 or mh > s.rng.random():
            accepted[i] = True
            samples[:, i, mc] = xf.sample
            mc_idx[:, i] = theta_f
            theta_i, xi, pri = theta_f, xf, prf
        else:
            samples[:, i, mc] = samples[:, i - 1, mc]
            mc_idx[:, i] = mc_idx[:, i - 1]
            if ji > Q_REJECT_TRIGGER and not accepted[i - Q_REJECT_TRIGGER:i + 1].any():  # refresh a stuck estimate
                xi, pri, comp = s.grid_point(theta_i, False)
                fx[2] += int(comp)
        if ji % a_h == 0:
            j = _adapt_jw(j_w, lar_j, j, accepted, a_h, ji, tgt_ar, mc_idx)
    return int(fx.sum()), float(accepted.mean()), float(accepted[adapt_period:].mean())


def run_inner_mcmc_analysis(model: ARQModel, prior, steps, burnin, chains, tgt_ar, grid, sample_limit=C_DF_ARQ_SL,
                            sample_dispersal=C_DF_ARQ_SR, jitter=C_DF_ARQ_JT, rng=None, verbose=True) -> ARQMCMCSample:
    """src/arq_main.jl:55-87."""
    start = time.perf_counter_ns()
    rng = rng or np.random.default_rng(1)
    n_theta = len(model.sample_interval)
    samples = np.zeros((n_theta, steps, chains))
    fx = np.zeros(chains, dtype=np.int64)
    s = _Sampler(model, prior, sample_limit, sample_dispersal, jitter, grid, rng)
    for c in range(chains):  # sequential, sharing the cache (:65-71)
        calls, ar, aar = _arq_met_hastings(samples, c, s, steps, burnin, tgt_ar)
        fx[c] = calls
        if verbose:
            print(f" chain {c + 1} - complete (calls to f(θ) := {calls}; AAR := {100 * aar:.1f}%)")
    rejs = handle_rej_samples(samples, burnin)
    theta = np.stack([g.sample for g in grid.values()], axis=1)            # collect_theta_weight, arq_utils.jl:6-14
    w = np.array([math.exp(g.log_likelihood) for g in grid.values()])
    mu, cv = compute_is_mu_covar(theta, w)
    bme = np.array([-math.log(w.sum() / len(w)), -math.log(w.sum() / (len(w) ** (1 / n_theta)))])  # :79
    imp = ImportanceSample(mu, cv, theta, w, 0, bme)
    out = ARQMCMCSample(imp, rejs, model.sample_interval, sample_limit, sample_dispersal, burnin, gelman_sre(samples, burnin),
                        time.perf_counter_ns() - start, fx, grid)
    if verbose:
        print(f"- finished in {out.run_time / 1e9:.1f} seconds. (Iμ = {np.round(mu, 6)}; Rμ = {np.round(rejs.mu, 6)})")
    return out


def run_arq_mcmc_analysis(model, obs_data, sample_interval, sample_offset=None, sample_dispersal: int = C_DF_ARQ_SR,
                          sample_limit: int = C_DF_ARQ_SL, n_chains: int = C_DF_ARQ_MC, steps: int = C_DF_MCMC_STEPS,
                          burnin: Optional[int] = None, tgt_ar: float = C_DF_ARQ_AR, np_: int = 200, ess_crit: float = 0.3,
                          seed: int = 1, device: int = 0, sample_cache: Optional[dict] = None, verbose: bool = True
                          ) -> ARQMCMCSample:
    """``run_arq_mcmc_analysis(model, obs_data, sample_interval; ...)`` (src/DiscretePOMP.jl:342-350):
    the target density is the particle filter with ``np_`` particles (``get_log_pdf_fn``, :313)."""
    from .particle_filter import get_particle_filter_lpdf
    sample_interval = np.asarray(sample_interval, dtype=np.float64)
    sample_offset = sample_interval / 2 if sample_offset is None else np.asarray(sample_offset, dtype=np.float64)
    burnin = int(steps * C_DF_MCMC_ADAPT) if burnin is None else burnin
    pdf = get_particle_filter_lpdf(model, obs_data, np=np_, essc=ess_crit, seed=seed, device=device)
    if verbose:
        print(f"ARQ model initialised: {model.model_name}")
    grid = {} if sample_cache is None else sample_cache
    try:
        return run_inner_mcmc_analysis(ARQModel(pdf, sample_interval, sample_offset), lambda th: model.prior.logpdf(th), steps,
                                       burnin, n_chains, tgt_ar, grid, sample_limit, sample_dispersal, rng=np.random.default_rng(seed),
                                       verbose=verbose)
    finally:
        pdf.filter.close()
