"""Particle-marginal Metropolis-Hastings: the PF-inside-MCMC driver.

The reference ships this path only as dead code: ``run_pmcmc`` (src/hmm_mcmc.jl:349-365) calls the
commented-out ``generic_mcmc!`` (:166-211), a non-existent ``model.fn_log_prior`` and a 5-argument
``estimate_likelihood``.  This module follows their *intent*, line for line where it exists:

    target              log prior(theta) + PF estimate of log p(y | theta)        (:355-357)
    initial proposal    MvNormal(diag(0.1 * theta0^2))  (0.1 where theta0 == 0)   (:172-175)
    proposal            theta + c * MvNormal draw, c0 = C_INITIAL = 0.1           (:7, :176, :181)
    accept              ll_f != -Inf and (exp(ll_f - ll_i) > rand())              (:183-189)
    adaptation          while i < adapt_period: c *= 1.002 on accept, 0.999 on reject;
                        every adapt_period/10 steps the proposal covariance becomes the sample
                        covariance of the chain so far                             (:196-208)
    result              samples[theta, step, chain], post-adaptation mean/covariance (handle_rej_samples,
                        src/cmn.jl:8-17) and a Gelman-Rubin scale-reduction point estimate

MI355X-first difference: the chains advance in lockstep and every iteration evaluates ALL chains'
proposals with ONE batched filter call (chain = segment); across GPUs the chains are sharded with
``parallel.sharded_loglik`` and only their log-likelihood estimates are all-gathered.
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np

C_INITIAL = 0.1            # src/hmm_mcmc.jl:7
C_DF_MCMC_STEPS = 50000    # src/DiscretePOMP.jl:42
C_DF_MCMC_ADAPT = 0.2      # :43


@dataclass
class RejectionSample:
    """src/cmn_structs.jl:13-17 (theta index; step; chain)."""
    theta: np.ndarray
    mu: np.ndarray
    cv: np.ndarray


@dataclass
class MCMCSample:
    """src/hmm_structs.jl:147-152."""
    samples: RejectionSample
    adapt_period: int
    sre: np.ndarray          # Gelman-Rubin potential scale reduction, point estimate per parameter
    run_time: int            # ns
    acceptance_rate: np.ndarray = None  # per chain, after adaptation
    status: int = 0          # OR of the filter status bits of every call (1 = event cap hit, 2 = all weights zero)


def handle_rej_samples(theta: np.ndarray, ap: int = 0) -> RejectionSample:
    """src/cmn.jl:8-17: mean and covariance of the pooled post-adaptation samples."""
    K = theta.shape[0]
    flat = theta[:, ap:, :].reshape(K, -1)
    return RejectionSample(theta, flat.mean(axis=1), np.atleast_2d(np.cov(flat)))


def gelman_sre(theta: np.ndarray, discard: int) -> np.ndarray:
    """Point estimate of the scale reduction factor (the middle column of the reference's
    ``gelman_diagnostic``, src/cmn.jl:20-88, without its F-distribution interval)."""
    x = theta[:, discard:, :]
    K, n, m = x.shape
    if m < 2 or n < 2:
        return np.full(K, np.nan)
    means = x.mean(axis=1)
    W = x.var(axis=1, ddof=1).mean(axis=1)
    B = n * means.var(axis=1, ddof=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.sqrt((n - 1) / n + (1 + 1 / m) * B / (n * W))


def run_pmcmc(evaluate: Callable[[np.ndarray, int], np.ndarray], prior, theta_init: np.ndarray,
              steps: int = C_DF_MCMC_STEPS, adapt_period: Optional[int] = None, seed: int = 1,
              rng: Optional[np.random.Generator] = None, verbose: bool = True) -> MCMCSample:
    """Adaptive PMMH for the ``C`` chains started at the columns of ``theta_init`` (K x C).

    ``evaluate(theta (K x C), call_seed) -> ll[C]`` runs one particle filter per column, e.g.
    ``lambda th, s: pf.loglik(th, 200, seed=s)``.
    """
    start = time.perf_counter_ns()
    rng = rng or np.random.default_rng(seed)
    theta_init = np.asarray(theta_init, dtype=np.float64)
    K, C = theta_init.shape
    if adapt_period is None:
        adapt_period = int(steps * C_DF_MCMC_ADAPT)
    adapt_interval = max(1, adapt_period // 10)
    samples = np.zeros((K, steps, C))
    samples[:, 0, :] = theta_init
    call = [0]

    def next_seed():
        call[0] += 1
        return (int(seed) * 0x9E3779B97F4A7C15 + call[0]) & (2 ** 64 - 1)

    lp = np.array([prior.logpdf(theta_init[:, c]) for c in range(C)])
    ll_i = lp + np.asarray(evaluate(theta_init, next_seed()))
    chol = np.stack([np.diag(np.sqrt(0.1 * np.where(theta_init[:, c] == 0.0, 1.0, theta_init[:, c] ** 2)))
                     for c in range(C)])  # [C][K][K]
    scale = np.full(C, C_INITIAL)
    acc_post = np.zeros(C)
    for i in range(1, steps):
        cur = samples[:, i - 1, :]
        z = rng.standard_normal((C, K))
        prop = cur + (scale[:, None] * np.einsum("cij,cj->ci", chol, z)).T  # get_mv_param(propd, c, theta)
        lp_f = np.array([prior.logpdf(prop[:, c]) for c in range(C)])
        ok = np.isfinite(lp_f)
        ll_f = np.full(C, -np.inf)
        if ok.any():
            th_run = np.where(ok[None, :], prop, cur)  # out-of-support proposals are rejected without a filter result
            ll_f = np.where(ok, lp_f + np.asarray(evaluate(th_run, next_seed())), -np.inf)
        u = rng.random(C)
        with np.errstate(over="ignore", invalid="ignore"):
            accepted = np.isfinite(ll_f) & (np.exp(ll_f - ll_i) > u)
        samples[:, i, :] = np.where(accepted[None, :], prop, cur)
        ll_i = np.where(accepted, ll_f, ll_i)
        if i < adapt_period:
            scale *= np.where(accepted, 1.002, 0.999)
            if i % adapt_interval == 0:
                for c in range(C):
                    cov = np.atleast_2d(np.cov(samples[:, : i + 1, c]))
                    if cov.sum() == 0:
                        if verbose:
                            print("warning: low acceptance rate detected in adaptation period")
                        continue
                    try:
                        chol[c] = np.linalg.cholesky(cov)
                    except np.linalg.LinAlgError:
                        pass
        else:
            acc_post += accepted
    rejs = handle_rej_samples(samples, adapt_period)
    out = MCMCSample(rejs, adapt_period, gelman_sre(samples, adapt_period), time.perf_counter_ns() - start,
                     acc_post / max(1, steps - adapt_period))
    if verbose:
        print(f"- finished in {out.run_time / 1e9:.1f} seconds. E(x) := {out.samples.mu}")
    return out


def run_pmcmc_analysis(model, obs_data, n_chains: int = 3, initial_parameters: Optional[np.ndarray] = None,
                       steps: int = C_DF_MCMC_STEPS, adapt_period: Optional[int] = None, np_: int = 200,
                       seed: int = 1, device: int = 0, distributed: bool = False, verbose: bool = True) -> MCMCSample:
    """Public wrapper in the shape of ``run_mcmc_analysis`` (src/DiscretePOMP.jl:185-194) for the particle
    filter path.  ``distributed=True`` shards the chains over the initialised ``torch.distributed`` group."""
    from .particle_filter import ParticleFilter
    rng = np.random.default_rng(seed)
    if initial_parameters is None:
        initial_parameters = model.prior.rand(rng, n_chains)  # rand(model.prior, n_chains)
    pf = ParticleFilter(model, obs_data, device=device)
    if distributed:
        from . import parallel

        def evaluate(th, s):
            return parallel.sharded_loglik(lambda t, base: pf.loglik(t, np_, seed=s, seg_base=base), th)
    else:
        def evaluate(th, s):
            return pf.loglik(th, np_, seed=s)
    if verbose:
        print(f"Running PMCMC analysis: {initial_parameters.shape[1]} x {steps} samples")
    try:
        out = run_pmcmc(evaluate, model.prior, initial_parameters, steps, adapt_period, seed=seed, rng=rng, verbose=verbose)
        out.status = pf.status_accum
        return out
    finally:
        pf.close()
