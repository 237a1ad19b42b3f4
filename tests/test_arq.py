"""ARQ-MCMC (src/arq_*.jl): the reference's working PF-inside-MCMC sampler.  Its testset pins
imp_sample.mu[1] ~ 0.003217 at Julia seed 1 (test/runtests.jl:30-37); the exact posterior mean is 0.003246."""
import json
import math
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EXACT = json.load(open(os.path.join(GOLDEN, "pf_golden.json")))["exact_pooley_sis_evidence_u001_u05"]
REFERENCE_SEED1_MU1 = 0.003217


def test_arq_on_a_known_density(dp):
    """Sampler logic alone: a noise-free Gaussian log-density on the grid; importance-sample and chain means
    must recover its mean, the cache must serve repeat visits."""
    m = np.array([0.004, 0.2])
    sd = np.array([0.0008, 0.05])
    calls = [0]

    def pdf(th):
        calls[0] += 1
        return float(-0.5 * np.sum(((th - m) / sd) ** 2))

    prior = dp.UniformProductPrior([0.0, 0.0], [0.01, 0.5])
    model = dp.ARQModel(pdf, np.array([0.0005, 0.02]), np.array([0.00025, 0.01]))
    out = dp.run_inner_mcmc_analysis(model, prior.logpdf, 8000, 2000, 3, 0.33, {}, rng=np.random.default_rng(2), verbose=False)
    assert calls[0] == len(out.sample_cache) <= 20 * 25      # sample_limit = 1: one evaluation per visited cell
    assert calls[0] == int(out.fx.sum())
    assert np.allclose(out.imp_sample.mu, m, rtol=0.03) and np.allclose(out.samples.mu, m, rtol=0.05)
    assert np.all(out.sre < 1.1)
    grid_mass = sum(math.exp(g.log_likelihood) for g in out.sample_cache.values())
    assert abs(out.imp_sample.bme[0] + math.log(grid_mass / len(out.sample_cache))) < 1e-12


def test_arq_with_the_oracle_filter(dp, orc, pooley):
    desc = dp.ModelDescriptor(dp.generate_model("SIS", [100, 1]), pooley)
    prior = dp.UniformProductPrior([0.0, 0.0], [0.01, 0.5])
    calls = [0]

    def pdf(th):
        calls[0] += 1
        return orc.pf_loglik(desc, 200, th, seed=calls[0]).ll

    model = dp.ARQModel(pdf, np.array([0.0005, 0.02]), np.array([0.00025, 0.01]))
    out = dp.run_inner_mcmc_analysis(model, prior.logpdf, 10000, 2000, 2, 0.33, {}, rng=np.random.default_rng(4), verbose=False)
    # this short configuration (2 chains x 10000 steps) measured over 14 seeds: mu[0] = 0.00315 +- 0.00008; 3.5 sd bands
    assert abs(out.imp_sample.mu[0] - 0.00315) < 0.00028
    assert abs(out.imp_sample.mu[0] - EXACT["beta_mean"]) < 0.00038
    assert abs(out.imp_sample.mu[0] - REFERENCE_SEED1_MU1) < 0.00035


@pytest.mark.gpu
def test_arq_reference_configuration_on_gpu(dp, pooley):
    """The reference's arqmcmc testset: prior U(0,0.01) x U(0,0.5), sample_interval [0.0005, 0.02], 5 chains x
    50000 steps, 200-particle filter."""
    model = dp.generate_model("SIS", [100, 1])
    model.prior = dp.UniformProductPrior([0.0, 0.0], [0.01, 0.5])
    out = dp.run_arq_mcmc_analysis(model, pooley, [0.0005, 0.02], seed=1, verbose=False)
    assert out.samples.theta.shape == (2, 50000, 5) and int(out.fx.sum()) == len(out.sample_cache)
    # measured over 10 seeds on MI355X (tools/arq_spread.py): mu[0] = 0.003204 +- 0.000074, mu[1] = 0.1064 +- 0.0026;
    # the reference's seed-1 value 0.003217 sits at +0.2 sd.  3.5 sd bands:
    assert abs(out.imp_sample.mu[0] - REFERENCE_SEED1_MU1) < 0.00026, out.imp_sample.mu
    assert abs(out.imp_sample.mu[0] - EXACT["beta_mean"]) < 0.0003
    assert abs(out.imp_sample.mu[1] - EXACT["gamma_mean"]) < 0.011
    assert 200 < len(out.sample_cache) <= 500   # a few hundred filter evaluations in total
