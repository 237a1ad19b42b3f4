"""PF-inside-MCMC driver (intent of run_pmcmc / generic_mcmc!, src/hmm_mcmc.jl:166-211, :349-365)."""
import json
import os

import numpy as np
import pytest

from helpers import threaded

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EXACT = json.load(open(os.path.join(GOLDEN, "pf_golden.json")))["exact_pooley_sis_evidence_u001_u05"]


def test_driver_samples_a_known_target(dp):
    """Driver logic alone (proposal, accept rule, scale and covariance adaptation): with a noise-free
    Gaussian 'likelihood' the chains must reproduce its mean and covariance."""
    m = np.array([0.004, 0.2])
    cov = np.array([[1.0e-6, 6.0e-5], [6.0e-5, 9.0e-3]])
    prec = np.linalg.inv(cov)
    prior = dp.UniformProductPrior([0.0, 0.0], [0.01, 0.5])

    def evaluate(th, seed):
        d = th - m[:, None]
        return -0.5 * np.einsum("ic,ij,jc->c", d, prec, d)

    init = np.array([[0.003, 0.005, 0.0045], [0.1, 0.3, 0.25]])
    out = dp.run_pmcmc(evaluate, prior, init, steps=30000, adapt_period=6000, seed=2, verbose=False)
    assert out.samples.theta.shape == (2, 30000, 3) and out.adapt_period == 6000
    assert np.allclose(out.samples.mu, m, rtol=0.03)
    assert np.allclose(out.samples.cv, cov, rtol=0.2)
    assert np.all(out.sre < 1.05) and np.all(out.acceptance_rate > 0.1) and np.all(out.acceptance_rate < 0.7)
    assert np.all(prior.logpdf(out.samples.theta[:, i, c]) > -np.inf for i in range(0, 30000, 997) for c in range(3))


def test_pmmh_with_the_oracle_filter(dp, orc, pooley):
    desc = dp.ModelDescriptor(dp.generate_model("SIS", [100, 1]), pooley)
    prior = dp.UniformProductPrior([0.0, 0.0], [0.01, 0.5])

    def evaluate(th, seed):
        return np.array(threaded(lambda c: orc.pf_loglik(desc, 64, th[:, c], seed=seed, seg=c).ll, th.shape[1]))

    # the scale moves by x1.002 / x0.999 per step from c0 = 0.1 (reference constants), so adaptation needs
    # a few thousand steps; the reference's defaults are 50000 steps with 10000 of adaptation
    init = np.array([[0.003, 0.0035], [0.1, 0.12]])
    out = dp.run_pmcmc(evaluate, prior, init, steps=5000, adapt_period=2500, seed=4, verbose=False)
    assert abs(out.samples.mu[0] - EXACT["beta_mean"]) < 0.1 * EXACT["beta_mean"]
    assert abs(out.samples.mu[1] - EXACT["gamma_mean"]) < 0.15 * EXACT["gamma_mean"]
    assert np.all(out.sre < 1.3) and np.all(out.acceptance_rate < 0.8)


def test_rejection_sample_and_sre_helpers(dp):
    rng = np.random.default_rng(0)
    th = rng.normal(size=(2, 400, 3)) + np.array([1.0, -2.0])[:, None, None]
    rs = dp.handle_rej_samples(th, 100)
    assert np.allclose(rs.mu, [1.0, -2.0], atol=0.15) and rs.cv.shape == (2, 2)
    assert np.all(np.abs(dp.gelman_sre(th, 100) - 1.0) < 0.05)
    th[:, :, 0] += 5.0  # one chain elsewhere -> no convergence
    assert np.all(dp.gelman_sre(th, 100) > 1.5)


@pytest.mark.gpu
def test_pmmh_on_gpu_four_chains(dp, pooley):
    """The 4-chain PMMH benchmark shape on the Pooley data: 4 chains, batched proposals, one filter call per iteration."""
    model = dp.generate_model("SIS", [100, 1])
    model.prior = dp.UniformProductPrior([0.0, 0.0], [0.01, 0.5])
    init = np.array([[0.003, 0.0035, 0.0028, 0.004], [0.1, 0.12, 0.09, 0.14]])
    out = dp.run_pmcmc_analysis(model, pooley, initial_parameters=init, steps=12000, adapt_period=4000, np_=512, seed=3, verbose=False)
    assert abs(out.samples.mu[0] - EXACT["beta_mean"]) < 0.04 * EXACT["beta_mean"]
    assert abs(out.samples.mu[1] - EXACT["gamma_mean"]) < 0.06 * EXACT["gamma_mean"]
    assert np.all(out.sre < 1.1)
