"""RNG-independent pins beyond Pooley SIS: exact CTMC likelihoods of the SEIR and Lotka-Volterra data sets
(oracle/exact_ctmc.py, golden values in tests/golden/exact_ctmc.json made by tests/golden/make_exact_ctmc.py) and of
every other catalogue model on simulated data, the oracle's unbiasedness against them, and the SPEC <-> FAITHFUL
bound on every catalogue model."""
import json
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN
from helpers import CATALOGUE_CASES, catalogue_case_exact, synthetic_data, threaded
from oracle import exact_ctmc, exact_sis

EXACT = json.load(open(os.path.join(GOLDEN, "exact_ctmc.json")))
SEIR_X0, LOTKA_X0 = [100, 0, 1, 0], [70, 70]


@pytest.mark.parametrize("theta", [(0.003, 0.1), (0.0032, 0.1), (0.0025, 0.08), (0.004, 0.15)])
def test_generic_ctmc_oracle_reproduces_the_dense_sis_one(theta):
    """Sparse generator + uniformisation (any model) against the dense expm of oracle/exact_sis.py."""
    want = exact_sis.exact_sis_loglik(theta, [100, 1], exact_sis.POOLEY_TIMES, exact_sis.POOLEY_CASES)[0]
    got = exact_ctmc.exact_loglik("SIS", theta, [100, 1], exact_sis.POOLEY_TIMES, exact_sis.POOLEY_CASES)
    assert got["n_states"] == 102 and abs(got["ll"] - want) < 1e-9 and got["truncation"] < 1e-12


def test_seir_golden_value_is_reproducible(dp):
    y = dp.get_observations(os.path.join(GOLDEN, "seir_synthetic.csv"))
    g = EXACT["SEIR"][0]
    r = exact_ctmc.exact_loglik("SEIR", g["theta"], SEIR_X0, [o.time for o in y], [o.val[2] for o in y])
    assert r["n_states"] == g["n_states"] == 182002   # (S, E, I) with S <= 100, S + E + I <= 101, minus unreachable ones
    assert abs(r["ll"] - g["ll"]) < 1e-9 and r["leak"] == 0.0
    assert np.allclose(r["increments"], g["increments"], atol=1e-9)


def test_lotka_box_truncation_has_converged():
    a, b = EXACT["LOTKA"]
    assert a["bound"] < b["bound"] and abs(a["ll"] - b["ll"]) < 1e-5 and b["leak"] < 1e-10
    assert a["ll"] <= b["ll"] + 1e-9  # absorbed mass only ever lowers the value


@pytest.mark.parametrize("mode", ["spec", "faithful"])
@pytest.mark.parametrize("which", [0, 2])
def test_oracle_is_unbiased_against_the_exact_seir_likelihood(dp, orc, mode, which):
    """E[exp(ll_hat)] = p(y | theta) for SEIR (obs seq = 3, seir_rf): pins the oracle's SEIR rate function,
    transition matrix and observed component without any random number."""
    y = dp.get_observations(os.path.join(GOLDEN, "seir_synthetic.csv"))
    desc = dp.ModelDescriptor(dp.generate_model("SEIR", SEIR_X0), y)
    g = EXACT["SEIR"][which]
    m = orc.MODE_SPEC if mode == "spec" else orc.MODE_FAITHFUL
    R = 128
    lls = np.array(threaded(lambda s: orc.pf_loglik(desc, 1024, g["theta"], seed=s + 1, mode=m).ll, R))
    ratio = np.exp(lls - g["ll"])
    se = ratio.std(ddof=1) / math.sqrt(R)
    assert abs(ratio.mean() - 1.0) < 3.5 * se and se < 0.02, (ratio.mean(), se)
    assert abs((g["ll"] - lls.mean()) - lls.var() / 2) < 0.03


def test_oracle_is_unbiased_against_the_exact_lotka_likelihood(dp, orc):
    y = dp.get_observations(os.path.join(GOLDEN, "lotka_synthetic.csv"))
    desc = dp.ModelDescriptor(dp.generate_model("LOTKA", LOTKA_X0), y)
    g = EXACT["LOTKA"][1]
    R = 48
    lls = np.array(threaded(lambda s: orc.pf_loglik(desc, 2048, g["theta"], seed=s + 1).ll, R))
    ratio = np.exp(lls - g["ll"])
    se = ratio.std(ddof=1) / math.sqrt(R)
    assert abs(ratio.mean() - 1.0) < 3.5 * se, (ratio.mean(), se)


@pytest.mark.parametrize("name,x0,fd,theta", [c for c in CATALOGUE_CASES if c[0] != "LOTKA"])
def test_oracle_is_unbiased_against_the_exact_likelihood_of_every_catalogue_model(dp, orc, name, x0, fd, theta):
    """E[exp(ll_hat)] = p(y | theta) with the exact value from a generator built directly from the reference's rate
    function, transition matrix and observed component (src/hmm_examples.jl:103-204): pins every catalogue model --
    density and frequency dependent -- in the oracle without any random number.  (Lotka-Volterra: above.)"""
    model, y, exact = catalogue_case_exact(dp, orc, name, x0, fd, theta)
    desc = dp.ModelDescriptor(model, y)
    R = 128
    lls = np.array(threaded(lambda s: orc.pf_loglik(desc, 512, theta, seed=s + 1).ll, R))
    ratio = np.exp(lls - exact)
    se = ratio.std(ddof=1) / math.sqrt(R)
    assert abs(ratio.mean() - 1.0) < 4.0 * se and se < 0.03, (ratio.mean(), se)


def test_exact_likelihood_tells_the_catalogue_variants_apart(dp, orc):
    """The pin has teeth: on the same data the frequency-dependent rate function, a wrong observed component or the
    SEIS transition matrix in place of SEIR's move the exact value by far more than the test above tolerates."""
    from oracle import exact_ctmc
    model, y, exact = catalogue_case_exact(dp, orc, "SEI", [100, 0, 1], False, [0.005, 0.2])
    t, v2, v1 = [o.time for o in y], [o.val[2] for o in y], [o.val[1] for o in y]
    fd = exact_ctmc.exact_loglik("SEI_FD", [0.005, 0.2], [100, 0, 1], t, v2)["ll"]
    wrong_comp = exact_ctmc.exact_loglik("SEI", [0.005, 0.2], [100, 0, 1], t, v1)["ll"]
    assert abs(fd - exact) > 1.0 and abs(wrong_comp - exact) > 0.2
    _, y2, e2 = catalogue_case_exact(dp, orc, "SEIS", [100, 0, 1], False, [0.005, 0.2, 0.1])
    as_sei = exact_ctmc.exact_loglik("SEI", [0.005, 0.2], [100, 0, 1], [o.time for o in y2], [o.val[2] for o in y2])["ll"]
    assert abs(as_sei - e2) > 0.05


def chi_square_against_exact(counts_by_state, states, probs, n, min_expected=25.0):
    """(statistic, degrees of freedom) of observed state counts against the exact law; states with small expected
    counts are pooled into one cell."""
    expected = probs * n
    big = expected >= min_expected
    obs = np.array([counts_by_state.get(tuple(s), 0) for s in states], dtype=float)
    o = np.append(obs[big], obs[~big].sum() + (n - obs.sum()))
    e = np.append(expected[big], expected[~big].sum() + n * max(0.0, 1.0 - probs.sum()))
    keep = e > 0
    return float((((o - e) ** 2) / np.where(keep, e, 1.0))[keep].sum()), int(keep.sum()) - 1


@pytest.mark.parametrize("name,x0,theta,t", [("SIS", [100, 1], [0.003, 0.1], 30.0), ("SIR", [100, 1, 0], [0.003, 0.1], 25.0),
                                             ("SEIS", [100, 0, 1], [0.005, 0.2, 0.1], 20.0)])
def test_oracle_propagation_samples_the_exact_transient_law(dp, orc, make_obs, name, x0, theta, t):
    """The Gillespie propagation on its own: with unweighted rows (obs_id = 0: no weights, no resampling) the particles
    of a filter are independent trajectories, so their states at time t must follow exp(Q t) applied to the initial
    condition -- computed without random numbers from the reference's rate function and transition matrix."""
    y = make_obs([t / 2, t], [[0] * len(x0)] * 2, obs_ids=[0, 0])
    desc = dp.ModelDescriptor(dp.generate_model(name, x0), y)
    P = 40000
    r = orc.pf_loglik(desc, P, theta, seed=11, nthreads=8)
    states, probs, leak = exact_ctmc.transient_distribution(name, theta, x0, t)
    assert leak == 0.0 and abs(probs.sum() - 1.0) < 1e-9
    keys, cnt = np.unique(r.pop, axis=0, return_counts=True)
    stat, df = chi_square_against_exact({tuple(k): c for k, c in zip(keys, cnt)}, states, probs, P)
    assert df > 10 and abs(stat - df) < 5.0 * math.sqrt(2.0 * df), (stat, df)


T0_CASE = dict(times=[20.0, 30.0, 40.0, 60.0], vals=[[0, 18], [0, 0], [0, 65], [0, 70]], obs_ids=[1, 0, 1, 1],
               theta=[0.003, 0.1, 4.5])


def t0_case_model(dp):
    rf = dp.RateFunction([dp.RateTerm(1, ((1,), (2,))), dp.RateTerm(2, ((2,),))])
    return dp.generate_custom_model("SIS", rf, [100, 1], [[-1, 1], [1, -1]], t0_index=3)


def t0_case_exact():
    c = T0_CASE
    return exact_ctmc.exact_loglik("SIS", c["theta"][:2], [100, 1], c["times"], [v[1] for v in c["vals"]], t0=c["theta"][2],
                                   obs_ids=c["obs_ids"])["ll"]


def test_oracle_is_unbiased_with_unknown_initial_time_and_an_unweighted_row(dp, orc, make_obs):
    """t0_index (the trajectory starts at theta[3], src/hmm_particle_filter.jl:45-47) and an obs_id = 0 row (propagate
    only, :58) against the exact likelihood with the same start time and the same row skipped."""
    c = T0_CASE
    desc = dp.ModelDescriptor(t0_case_model(dp), make_obs(c["times"], c["vals"], obs_ids=c["obs_ids"]))
    exact = t0_case_exact()
    R = 160
    lls = np.array(threaded(lambda s: orc.pf_loglik(desc, 1024, c["theta"], seed=s + 1).ll, R))
    ratio = np.exp(lls - exact)
    se = ratio.std(ddof=1) / math.sqrt(R)
    assert abs(ratio.mean() - 1.0) < 4.0 * se and se < 0.03, (ratio.mean(), se)
    # the start time matters: the exact value for t0 = 0 is far away
    assert abs(exact_ctmc.exact_loglik("SIS", c["theta"][:2], [100, 1], c["times"], [v[1] for v in c["vals"]],
                                       obs_ids=c["obs_ids"])["ll"] - exact) > 0.1


@pytest.mark.parametrize("name,x0,fd,theta", CATALOGUE_CASES)
def test_spec_mode_tracks_faithful_mode_on_every_catalogue_model(dp, orc, name, x0, fd, theta):
    """SPEC (what the GPU computes) against FAITHFUL (the reference's Float64 left fold, libm log/exp, rsp_systematic
    verbatim) on all catalogue rate functions: same event counts, log-likelihood within 1e-11, ancestors equal except
    where a resampling point falls within rounding of a boundary."""
    model = dp.generate_model(name, x0, freq_dep=fd)
    desc = dp.ModelDescriptor(model, synthetic_data(dp, orc, model, theta))
    diff = 0
    for seed in (1, 2, 3):
        a = orc.pf_loglik(desc, 1024, theta, seed=seed, mode=orc.MODE_SPEC, trace=True)
        b = orc.pf_loglik(desc, 1024, theta, seed=seed, mode=orc.MODE_FAITHFUL, trace=True)
        assert a.events == b.events
        assert (math.isinf(a.ll) and a.ll == b.ll) or abs(a.ll - b.ll) < 1e-11, (a.ll, b.ll)
        diff += int((a.ancestors != b.ancestors).sum())
    assert diff <= 2


def test_spec_mode_tracks_faithful_mode_with_t0_index_and_unweighted_rows(dp, orc, make_obs):
    """Unknown initial time (t0_index) and an obs_id = 0 row (propagate only: no weight, no resampling)."""
    rf = dp.RateFunction([dp.RateTerm(1, ((1,), (2,))), dp.RateTerm(2, ((2,),))])
    model = dp.generate_custom_model("SIS", rf, [100, 1], [[-1, 1], [1, -1]], t0_index=3)
    y = make_obs([20.0, 30.0, 40.0, 60.0], [[0, 18], [0, 0], [0, 65], [0, 70]], obs_ids=[1, 0, 1, 1])
    desc = dp.ModelDescriptor(model, y)
    for seed in (1, 2, 3, 4):
        a = orc.pf_loglik(desc, 1024, [0.003, 0.1, 4.5], seed=seed, mode=orc.MODE_SPEC, trace=True)
        b = orc.pf_loglik(desc, 1024, [0.003, 0.1, 4.5], seed=seed, mode=orc.MODE_FAITHFUL, trace=True)
        assert a.events == b.events and abs(a.ll - b.ll) < 1e-11
        assert a.per_obs_ll[1] == 0.0 and b.per_obs_ll[1] == 0.0
        assert int((a.ancestors != b.ancestors).sum()) <= 1


def test_seir_evidence_table_is_reproducible_and_converged(dp):
    """tests/golden/seir_evidence.json (made by tests/golden/make_seir_evidence.py: exact SEIR likelihood on a tensor
    Gauss-Legendre grid over a uniform prior box): the two stored rules agree (quadrature converged to < 0.01 in
    -ln evidence, 0.4 % in the posterior mean), the box holds the posterior mass, and a stored node is reproduced by the
    forward algorithm today."""
    import json
    ev = json.load(open(os.path.join(GOLDEN, "seir_evidence.json")))
    fine, coarse = ev["rules"]
    assert fine["n"] == 12 and coarse["n"] == 9
    assert abs(fine["neg_log_evidence"] - coarse["neg_log_evidence"]) < 0.01 and ev["quadrature_error_estimate"] < 0.01
    assert np.allclose(fine["posterior_mean"], coarse["posterior_mean"], rtol=8e-3)
    assert max(fine["edge_share"]) < 0.02 and fine["max_truncation"] < 1e-12
    # weights of the rule reproduce the stored evidence from the stored log-likelihoods
    x, w = np.polynomial.legendre.leggauss(12)
    W = np.einsum("i,j,k->ijk", w / 2, w / 2, w / 2).reshape(-1)
    ll = np.array(fine["ll"])
    assert abs(-(ll.max() + np.log(np.sum(W * np.exp(ll - ll.max())))) - fine["neg_log_evidence"]) < 1e-12
    # one node, recomputed: the node of the largest likelihood
    i = int(ll.argmax())
    a = [fine["axes"][0][i // 144], fine["axes"][1][(i // 12) % 12], fine["axes"][2][i % 12]]
    y = dp.get_observations(os.path.join(GOLDEN, "seir_synthetic.csv"))
    r = exact_ctmc.exact_loglik("SEIR", a, [100, 0, 1, 0], [o.time for o in y], [o.val[2] for o in y])
    assert abs(r["ll"] - ll[i]) < 1e-9
