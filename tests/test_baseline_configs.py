"""The SEIR and Lotka-Volterra benchmark configurations (3, 4 and 5 of the benchmark list) at their own workload (one
GPU's share of it), through the drivers that name them:

  3  SEIR, 2^18 particles, 4-chain particle-marginal MH                   (run_pmcmc_analysis)
  4  SEIR, run_smc2 with 512 theta-particles x 2^16 over 8 GPUs           (one rank's share: 64 x 2^16, run_pibis)
  5  Lotka-Volterra, 2^20 particles, 8 independent chains on 8 GPUs       (one rank's share: 1 chain x 2^20)

Each config is checked three ways: the driver over the GPU filter equals the same driver over the oracle filter bit
for bit at a size the oracle finishes in seconds; at full size the first observation (propagation, weights,
resampling) equals the oracle bit for bit and the rest is covered by properties (conservation laws, status bits,
finite estimates); and the GPU filter's estimates agree with the exact CTMC likelihood of the data set
(tests/golden/exact_ctmc.json, made by tests/golden/make_exact_ctmc.py from the reference's model definitions alone).
Reference code on these paths: src/hmm_examples.jl:117-121, :149-154, :188-199 (models), src/hmm_ibis.jl:50-116
(SMC^2 loop), src/hmm_mcmc.jl:166-211 (PMMH intent).
"""
import json
import math
import os
import warnings

import numpy as np
import pytest

from conftest import GOLDEN
from helpers import CATALOGUE_CASES, OracleBank, catalogue_case_exact

pytestmark = pytest.mark.gpu

EXACT = json.load(open(os.path.join(GOLDEN, "exact_ctmc.json")))
SEIR_X0, SEIR_THETA = [100, 0, 1, 0], [0.005, 0.2, 0.1]
LOTKA_X0, LOTKA_THETA = [70, 70], [0.5, 0.0025, 0.3]


@pytest.fixture(scope="module")
def seir_data(dp):
    return dp.get_observations(os.path.join(GOLDEN, "seir_synthetic.csv"))


@pytest.fixture(scope="module")
def lotka_data(dp):
    return dp.get_observations(os.path.join(GOLDEN, "lotka_synthetic.csv"))


def _seir_model(dp):
    m = dp.generate_model("SEIR", SEIR_X0)
    m.prior = dp.UniformProductPrior([0.0, 0.0, 0.0], [0.02, 0.6, 0.4])
    return m


# ---- exact likelihood: the GPU filter is unbiased on SEIR and Lotka-Volterra -------------------------------------
@pytest.mark.parametrize("which", [0, 1, 2, 3])
def test_gpu_filter_is_unbiased_against_the_exact_seir_likelihood(dp, seir_data, which):
    g = EXACT["SEIR"][which]
    pf = dp.ParticleFilter(dp.generate_model("SEIR", SEIR_X0), seir_data)
    B = 2000
    lls = pf.loglik(np.tile(np.array(g["theta"])[:, None], (1, B)), 2048, seed=4100 + which)
    assert np.all(np.isfinite(lls)) and np.all(pf.last_status == 0)
    ratio = np.exp(lls - g["ll"])
    se = ratio.std(ddof=1) / math.sqrt(B)
    assert abs(ratio.mean() - 1.0) < 4 * se and se < 0.004, (ratio.mean(), se)
    assert abs((g["ll"] - lls.mean()) - lls.var() / 2) < 4 * lls.std() / math.sqrt(B) + 0.002
    pf.close()


def test_gpu_filter_is_unbiased_against_the_exact_lotka_likelihood(dp, lotka_data):
    g = EXACT["LOTKA"][1]
    pf = dp.ParticleFilter(dp.generate_model("LOTKA", LOTKA_X0), lotka_data)
    B = 1000
    lls = pf.loglik(np.tile(np.array(g["theta"])[:, None], (1, B)), 4096, seed=4200)
    assert np.all(np.isfinite(lls)) and np.all(pf.last_status == 0)
    ratio = np.exp(lls - g["ll"])
    se = ratio.std(ddof=1) / math.sqrt(B)
    assert abs(ratio.mean() - 1.0) < 4 * se and se < 0.02, (ratio.mean(), se)
    pf.close()


@pytest.mark.parametrize("name,x0,fd,theta", [c for c in CATALOGUE_CASES if c[0] != "LOTKA"])
def test_gpu_filter_is_unbiased_against_the_exact_likelihood_of_every_catalogue_model(dp, orc, name, x0, fd, theta):
    """Every catalogue model, density and frequency dependent, through its compiled evaluator on the GPU: 4096 filters
    of 1024 particles against the exact likelihood from a generator built from the reference's rate function,
    transition matrix and observed component alone (src/hmm_examples.jl:103-204) -- no random number, no oracle RNG."""
    model, y, exact = catalogue_case_exact(dp, orc, name, x0, fd, theta)
    pf = dp.ParticleFilter(model, y)
    B = 4096
    lls = pf.loglik(np.tile(np.array(theta, dtype=float)[:, None], (1, B)), 1024, seed=4300)
    assert np.all(np.isfinite(lls)) and np.all(pf.last_status == 0)
    ratio = np.exp(lls - exact)
    se = ratio.std(ddof=1) / math.sqrt(B)
    assert abs(ratio.mean() - 1.0) < 4 * se and se < 0.003, (name, fd, ratio.mean(), se)
    pf.close()


@pytest.mark.parametrize("name,x0,theta,t,bound", [("SIS", [100, 1], [0.003, 0.1], 30.0, None), ("SIR", [100, 1, 0], [0.003, 0.1], 25.0, None),
                                                   ("SEIS", [100, 0, 1], [0.005, 0.2, 0.1], 20.0, None),
                                                   ("SEIR", [100, 0, 1, 0], [0.005, 0.2, 0.1], 30.0, None),
                                                   ("LOTKA", [70, 70], [0.5, 0.0025, 0.3], 3.0, 400)])
def test_gpu_propagation_samples_the_exact_transient_law(dp, make_obs, name, x0, theta, t, bound):
    """2^20 particles through two unweighted rows (no weights, no resampling: independent trajectories, gathered
    through the compaction schedule) against exp(Q t) applied to the initial condition: a chi-square test over every
    state with an expected count of 25 or more.  Pins the event loop -- waiting times, event choice, transitions, the
    stream per particle -- on the GPU without reference to the oracle's random numbers."""
    from oracle import exact_ctmc
    from test_exact_ctmc import chi_square_against_exact
    pf = dp.ParticleFilter(dp.generate_model(name, x0), make_obs([t / 2, t], [[0] * len(x0)] * 2, obs_ids=[0, 0]))
    P = 1 << 20
    pop = pf.alloc_population(1, P)
    pf.partial_log_likelihood(pop, np.array(theta, dtype=float)[:, None], 1, 2, seed=12)
    assert pf.last_status[0] == 0
    final = pop.download(0)
    states, probs, leak = exact_ctmc.transient_distribution(name, theta, x0, t, bound=bound)
    assert leak < 1e-9
    keys, cnt = np.unique(final, axis=0, return_counts=True)
    stat, df = chi_square_against_exact({tuple(k): c for k, c in zip(keys, cnt)}, states, probs, P)
    assert df > 30 and abs(stat - df) < 5.0 * math.sqrt(2.0 * df), (name, stat, df)
    pf.close()


def test_gpu_filter_is_unbiased_with_unknown_initial_time_and_an_unweighted_row(dp, make_obs):
    """t0_index and an obs_id = 0 row on the GPU against the exact likelihood (see tests/test_exact_ctmc.py)."""
    from test_exact_ctmc import T0_CASE, t0_case_exact, t0_case_model
    c = T0_CASE
    pf = dp.ParticleFilter(t0_case_model(dp), make_obs(c["times"], c["vals"], obs_ids=c["obs_ids"]))
    B = 4096
    lls = pf.loglik(np.tile(np.array(c["theta"])[:, None], (1, B)), 1024, seed=4400)
    ratio = np.exp(lls - t0_case_exact())
    se = ratio.std(ddof=1) / math.sqrt(B)
    assert abs(ratio.mean() - 1.0) < 4 * se and se < 0.004, (ratio.mean(), se)
    pf.close()


# ---- config 3: SEIR, 4-chain PMMH, 2^18 particles -----------------------------------------------------------------
def test_config3_pmmh_chains_equal_the_oracle_driven_chains(dp, orc, seir_data):
    """Same driver, same seeds, 512 particles: every sample of every chain identical."""
    model = _seir_model(dp)
    init = np.array([[0.005, 0.004, 0.006, 0.0045], [0.2, 0.25, 0.15, 0.2], [0.1, 0.1, 0.12, 0.08]])
    desc = dp.ModelDescriptor(model, seir_data)

    def evaluate_oracle(th, seed):
        return np.array([orc.pf_loglik(desc, 512, th[:, c], seed=seed, seg=c).ll for c in range(th.shape[1])])

    a = dp.run_pmcmc(evaluate_oracle, model.prior, init, steps=150, adapt_period=60, seed=5, verbose=False)
    b = dp.run_pmcmc_analysis(model, seir_data, initial_parameters=init, steps=150, adapt_period=60, np_=512, seed=5,
                              verbose=False)
    assert np.array_equal(a.samples.theta, b.samples.theta)
    assert b.status == 0 and len(np.unique(b.samples.theta[0, :, 0])) > 5  # the chains move


def test_config3_full_size_four_chains(dp, orc, seir_data):
    """4 chains x 2^18 particles, 400 iterations = 400 batched filter calls of 2^20 particles."""
    model = _seir_model(dp)
    init = np.array([[0.005, 0.004, 0.006, 0.0045], [0.2, 0.25, 0.15, 0.2], [0.1, 0.1, 0.12, 0.08]])
    out = dp.run_pmcmc_analysis(model, seir_data, initial_parameters=init, steps=400, adapt_period=200, np_=1 << 18, seed=6,
                                verbose=False)
    th = out.samples.theta
    assert th.shape == (3, 400, 4) and np.all(np.isfinite(th)) and out.status == 0
    assert all(model.prior.logpdf(th[:, i, c]) > -np.inf for i in range(0, 400, 37) for c in range(4))
    # with 2^18 particles the filter noise is tiny, so the chains behave like exact-likelihood MH: they stay where the
    # exact likelihood is high (the data were simulated at theta = (0.005, 0.2, 0.1))
    mu = th[:, 200:, :].reshape(3, -1).mean(axis=1)
    assert 0.003 < mu[0] < 0.008 and 0.1 < mu[1] < 0.4 and 0.05 < mu[2] < 0.2, mu
    # the filter estimate at this size, against the exact likelihood: 4 independent 2^18-particle estimates per theta
    pf = dp.ParticleFilter(model, seir_data)
    for g in EXACT["SEIR"]:
        lls = pf.loglik(np.tile(np.array(g["theta"])[:, None], (1, 4)), 1 << 18, seed=61)
        assert np.all(pf.last_status == 0) and np.all(np.abs(lls - g["ll"]) < 0.03), (lls, g["ll"])
    # first observation at full size against the oracle, all four segments: states, log-likelihood, event count
    pop = pf.alloc_population(4, 1 << 18)
    ll1 = pf.partial_log_likelihood(pop, init, 1, 1, seed=62)
    events = 0
    for c in range(4):
        host = np.zeros((1 << 18, 4), dtype=np.int64)
        ref = orc.pf_partial(pf.desc, host, init[:, c], 1, 1, seed=62, seg=c, nthreads=8)
        assert ll1[c] == ref.ll and np.array_equal(pop.download(c), host)
        events += ref.events
    assert pf.last_event_count() == events
    pf.close()


# ---- config 4: SEIR, SMC^2, one rank's share 64 theta-particles x 2^16 ------------------------------------------
def test_config4_smc2_on_seir_equals_the_oracle_bank(dp, orc, seir_data):
    model = _seir_model(dp)
    theta0 = model.prior.rand(np.random.default_rng(7), 48)
    a = dp.run_pibis(OracleBank(dp, orc, model, seir_data), model.prior, theta0, 0.3, True, 1.002, 128, seed=8,
                     rng=np.random.default_rng(9), verbose=False)
    pf = dp.ParticleFilter(model, seir_data)
    b = dp.run_pibis(pf, model.prior, theta0, 0.3, True, 1.002, 128, seed=8, rng=np.random.default_rng(9), verbose=False)
    assert a.n_resample >= 1 and a.n_resample == b.n_resample
    assert np.array_equal(a.bme, b.bme) and np.array_equal(a.theta, b.theta) and np.array_equal(a.weight, b.weight)
    pf.close()


def test_config4_full_size_share_of_one_rank(dp, orc, seir_data):
    """64 theta-particles x 2^16 state particles (what each of 8 GPUs holds of 512 x 2^16), global segment indices of
    rank 3 (seg_base = 192): SMC^2 runs, and the incremental filter it is made of keeps its invariants."""
    model = _seir_model(dp)
    N, P, base = 64, 1 << 16, 192
    rng = np.random.default_rng(10)
    theta0 = model.prior.rand(rng, N)
    pf = dp.ParticleFilter(model, seir_data)
    out = dp.run_pibis(pf, model.prior, theta0, 0.3, True, 1.002, P, seed=11, rng=rng, verbose=False)
    assert np.all(np.isfinite(out.bme)) and out.n_resample >= 1 and out.status & 1 == 0
    assert np.all(np.isfinite(out.mu)) and out.theta.shape == (3, N)
    # the increments one by one on device-resident populations, as hmm_ibis.jl:53-56 calls them
    pop = pf.alloc_population(N, P)
    total = np.zeros(N)
    for k in range(1, pf.n_obs + 1):
        total += pf.partial_log_likelihood(pop, theta0, k, k, seed=12, seg_base=base)
        assert np.all(pf.last_status & 1 == 0)
        for b in (0, 17, 63):
            x = pop.download(b)
            assert np.all(x >= 0) and np.all(x.sum(axis=1) == 101)        # S + E + I + R is conserved
            assert np.all(x[:, 0] <= 100) and np.all(x[:, 3] <= 101)
            if k < pf.n_obs and np.isfinite(total[b]):
                nj = pop.debug(dp.DBG_OFFSPRING_END, b)                    # offspring boundaries of this resampling
                assert np.all(np.diff(nj) >= 0) and nj[-1] == P and nj[0] >= 0
    whole = pf.loglik(theta0, P, seed=12, seg_base=base)                    # one call == the five increments
    assert np.array_equal(np.isfinite(whole), np.isfinite(total))
    fin = np.isfinite(whole)
    assert fin.sum() >= 8 and np.allclose(whole[fin], total[fin], rtol=0, atol=1e-9)
    # first observation of three segments against the oracle at full size (global segment index in the RNG key)
    pop2 = pf.alloc_population(N, P)
    ll1 = pf.partial_log_likelihood(pop2, theta0, 1, 1, seed=12, seg_base=base)
    for b in (0, 31, 63):
        host = np.zeros((P, 4), dtype=np.int64)
        ref = orc.pf_partial(pf.desc, host, theta0[:, b], 1, 1, seed=12, seg=base + b, nthreads=8)
        assert (ll1[b] == ref.ll or (math.isinf(ref.ll) and ll1[b] == ref.ll)) and np.array_equal(pop2.download(b), host)
    pf.close()


# ---- config 5: Lotka-Volterra, 2^20 particles, one chain per GPU --------------------------------------------------
def test_config5_lotka_full_size_filter_and_chain(dp, orc, lotka_data):
    model = dp.generate_model("LOTKA", LOTKA_X0)
    model.prior = dp.UniformProductPrior([0.0, 0.0, 0.0], [1.0, 0.005, 0.6])
    pf = dp.ParticleFilter(model, lotka_data)
    P = 1 << 20
    ll = pf.loglik(LOTKA_THETA, P, seed=13)
    assert pf.last_status[0] == 0 and abs(ll[0] - EXACT["LOTKA"][1]["ll"]) < 0.1, ll   # sd of the estimate ~0.02 at 2^20
    assert 3000 < pf.last_event_count() / P < 4500                                    # ~3600 events per particle
    # first observation at full size against the oracle
    pop = pf.alloc_population(1, P)
    ll1 = pf.partial_log_likelihood(pop, LOTKA_THETA, 1, 1, seed=13)
    host = np.zeros((P, 2), dtype=np.int64)
    ref = orc.pf_partial(pf.desc, host, LOTKA_THETA, 1, 1, seed=13, nthreads=8)
    assert ll1[0] == ref.ll and pf.last_event_count() == ref.events and np.array_equal(pop.download(0), host)
    pop.close()
    pf.close()
    # one rank's chain of config 5: 1 chain x 2^20 particles x 20 observations per iteration
    out = dp.run_pmcmc_analysis(model, lotka_data, initial_parameters=np.array(LOTKA_THETA)[:, None], steps=40,
                                adapt_period=20, np_=P, seed=14, verbose=False)
    assert out.status == 0 and np.all(np.isfinite(out.samples.theta))
    assert len(np.unique(out.samples.theta[0, :, 0])) > 2


def test_config5_lotka_event_cap_at_full_size(dp, orc, lotka_data):
    """An event cap that bites (the reference has none and would simply keep going): capped particles get weight zero,
    the status bit is raised, the wrapper warns -- and the capped filter still equals the oracle with the same cap."""
    model = dp.generate_model("LOTKA", LOTKA_X0)
    model.max_events_per_interval = 380   # around the predator peak (observations 6-8) ~430 +- 20 events happen per interval
    pf = dp.ParticleFilter(model, lotka_data)
    with pytest.warns(dp.EventCapWarning):
        small = pf.loglik(LOTKA_THETA, 3000, seed=15)
    ref = orc.pf_loglik(pf.desc, 3000, LOTKA_THETA, seed=15)
    assert pf.last_status[0] & dp.ST_EVENT_CAP and ref.status & dp.ST_EVENT_CAP
    assert small[0] == ref.ll and pf.last_event_count() == ref.events
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", dp.EventCapWarning)
        big = pf.loglik(LOTKA_THETA, 1 << 20, seed=15)
    assert pf.last_status[0] & dp.ST_EVENT_CAP
    assert pf.last_event_count() <= 380 * 20 * (1 << 20)
    assert np.isfinite(big[0]) and big[0] < EXACT["LOTKA"][1]["ll"] - 0.5   # zeroed weights: visibly below the true value
    pf.close()
    strict = dp.ParticleFilter(model, lotka_data, on_event_cap="raise")
    with pytest.raises(dp.DpompError):
        strict.loglik(LOTKA_THETA, 3000, seed=15)
    strict.close()
