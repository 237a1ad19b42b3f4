"""Pins the oracle at distribution level: the exact CTMC forward-algorithm likelihood of the
Pooley SIS data and the PF's unbiasedness on the likelihood scale."""
import math

import numpy as np
import pytest

from helpers import threaded
from oracle import exact_sis

KNOWN = {  # exact log-likelihoods, re-derived by oracle/exact_sis.py
    (0.003, 0.1): -15.6873794556,
    (0.0032, 0.1): -16.0197707226,
    (0.0025, 0.08): -15.4530158001,
    (0.004, 0.15): -17.3813549614,
}
INCREMENTS_003_01 = [-4.496179, -2.859243, -2.816745, -2.772524, -2.742689]


@pytest.mark.parametrize("theta,want", list(KNOWN.items()))
def test_exact_forward_algorithm_known_answers(theta, want):
    ll, incs = exact_sis.exact_sis_loglik(theta, [100, 1], exact_sis.POOLEY_TIMES, exact_sis.POOLEY_CASES)
    assert abs(ll - want) < 5e-10
    if theta == (0.003, 0.1):
        assert np.allclose(incs, INCREMENTS_003_01, atol=1e-6)


def test_exact_agrees_with_uniformisation():
    """Independent evaluation of exp(Q t) by uniformisation (Poisson-weighted powers of the
    jump chain) instead of scipy's Pade expm."""
    theta = (0.003, 0.1)
    Q = exact_sis.sis_generator(*theta, 101)
    lam = max(-Q.diagonal()) * 1.05
    Pm = np.eye(102) + Q / lam
    alpha = np.zeros(102)
    alpha[1] = 1.0
    tmp1, tmp2 = math.log(1 / (math.sqrt(2 * math.pi) * 2.0)), 8.0
    total = 0.0
    for y in exact_sis.POOLEY_CASES:
        v, acc, term_log = alpha.copy(), np.zeros(102), -lam * 20.0
        n = 0
        while n < 4000:
            acc += math.exp(term_log) * v
            v = v @ Pm
            n += 1
            term_log += math.log(lam * 20.0 / n)
        alpha = acc * np.exp(tmp1 - (y - np.arange(102)) ** 2 / tmp2)
        c = alpha.sum()
        total += math.log(c)
        alpha /= c
    assert abs(total - KNOWN[theta]) < 1e-7


@pytest.mark.parametrize("mode", ["spec", "faithful"])
def test_pf_is_unbiased_on_the_likelihood_scale(dp, orc, pooley, mode):
    """E[exp(ll_hat)] = p(y | theta): log mean_r exp(ll_r) within 3.5 standard errors of exact, and
    mean(ll_r) below exact by about Var/2 (Jensen).  The CPU-runnable benchmark configuration: P = 1024, seeds 1..R."""
    desc = dp.ModelDescriptor(dp.generate_model("SIS", [100, 1]), pooley)
    theta = (0.003, 0.1)
    R = 160
    m = orc.MODE_SPEC if mode == "spec" else orc.MODE_FAITHFUL
    lls = np.array(threaded(lambda s: orc.pf_loglik(desc, 1024, theta, seed=s + 1, mode=m).ll, R))
    exact = KNOWN[theta]
    ratio = np.exp(lls - exact)
    se = ratio.std(ddof=1) / math.sqrt(R)
    assert abs(ratio.mean() - 1.0) < 3.5 * se, (ratio.mean(), se)
    assert lls.mean() < exact + 3 * lls.std() / math.sqrt(R)
    assert abs((exact - lls.mean()) - lls.var() / 2) < 0.03


def test_other_thetas_statistically(dp, orc, pooley):
    desc = dp.ModelDescriptor(dp.generate_model("SIS", [100, 1]), pooley)
    for theta, exact in KNOWN.items():
        lls = np.array(threaded(lambda s: orc.pf_loglik(desc, 2048, theta, seed=100 + s).ll, 48))
        ratio = np.exp(lls - exact)
        assert abs(ratio.mean() - 1.0) < 4 * ratio.std(ddof=1) / math.sqrt(len(lls))


def test_spec_mode_tracks_reference_faithful_mode(dp, orc, pooley):
    """SPEC (fixed-point weights, own log/exp -- what the GPU computes) against FAITHFUL (the
    reference's Float64 left fold, libm, rsp_systematic verbatim): same trajectories except on
    measure-zero ties; log-likelihood within 1e-11."""
    desc = dp.ModelDescriptor(dp.generate_model("SIS", [100, 1]), pooley)
    diff_slots = 0
    for seed in range(1, 21):
        a = orc.pf_loglik(desc, 1024, (0.003, 0.1), seed=seed, mode=orc.MODE_SPEC, trace=True)
        b = orc.pf_loglik(desc, 1024, (0.003, 0.1), seed=seed, mode=orc.MODE_FAITHFUL, trace=True)
        assert abs(a.ll - b.ll) < 1e-11
        assert a.events == b.events
        diff_slots += int((a.ancestors != b.ancestors).sum())
    assert diff_slots <= 2  # ancestors may differ only where u_i falls within rounding of a boundary


def test_events_per_particle_matches_the_exact_expectation(dp, orc, pooley):
    desc = dp.ModelDescriptor(dp.generate_model("SIS", [100, 1]), pooley)
    r = orc.pf_loglik(desc, 8192, (0.003, 0.1), seed=3)
    assert 1050 < r.events / 8192 < 1120  # ~1085 events per particle per PF call (exact expectation under the filtered law)


@pytest.mark.parametrize("rs_type", [2, 3])
def test_stratified_and_multinomial_resamplers(dp, orc, pooley, rs_type):
    """rsp_stratified / rsp_multinomial (hmm_pf_resample.jl:46-63, :5-20; both throw as shipped): documented
    intent in FAITHFUL Float64 == SPEC fixed point, and the filter stays unbiased on the likelihood scale."""
    desc = dp.ModelDescriptor(dp.generate_model("SIS", [100, 1]), pooley)
    theta = (0.003, 0.1)
    for seed in range(1, 9):
        a = orc.pf_loglik(desc, 1024, theta, seed=seed, rs_type=rs_type, trace=True)
        b = orc.pf_loglik(desc, 1024, theta, seed=seed, rs_type=rs_type, mode=orc.MODE_FAITHFUL, trace=True)
        assert abs(a.ll - b.ll) < 1e-11 and int((a.ancestors != b.ancestors).sum()) <= 1
        if rs_type == 2:
            assert np.all(np.diff(a.ancestors) >= 0)  # one point per stratum: ancestors stay sorted
    lls = np.array(threaded(lambda s: orc.pf_loglik(desc, 1024, theta, seed=200 + s, rs_type=rs_type).ll, 120))
    ratio = np.exp(lls - KNOWN[theta])
    assert abs(ratio.mean() - 1.0) < 3.5 * ratio.std(ddof=1) / math.sqrt(len(lls))


def test_ess_adaptive_resampling_in_the_oracle(dp, orc, pooley):
    """The dormant `essc` criterion made real (SPEC mode): a threshold above 1 resamples always and reproduces the
    default path bit for bit; a real threshold changes the path, carries weights, and stays unbiased."""
    desc = dp.ModelDescriptor(dp.generate_model("SIS", [100, 1]), pooley)
    theta = (0.003, 0.1)
    for seed in (1, 2, 3):
        a = orc.pf_loglik(desc, 1024, theta, seed=seed, trace=True)
        b = orc.pf_loglik(desc, 1024, theta, seed=seed, ess_frac=1.5, trace=True)
        assert a.ll == b.ll and np.array_equal(a.pop, b.pop)
    differs = sum(orc.pf_loglik(desc, 1024, theta, seed=s, ess_frac=0.3).ll != orc.pf_loglik(desc, 1024, theta, seed=s).ll for s in range(1, 11))
    assert differs >= 5
    lls = np.array(threaded(lambda s: orc.pf_loglik(desc, 1024, theta, seed=300 + s, ess_frac=0.3).ll, 140))
    ratio = np.exp(lls - KNOWN[theta])
    assert abs(ratio.mean() - 1.0) < 3.5 * ratio.std(ddof=1) / math.sqrt(len(lls))
    # incremental calls carry the state exactly like one call
    pop = np.zeros((700, 2), dtype=np.int64)
    carry, total = None, 0.0
    for k in range(1, 6):
        r = orc.pf_partial(desc, pop, theta, k, k, seed=9, ess_frac=0.3, carry=carry)
        carry, total = r.carry, total + r.ll
    full = orc.pf_loglik(desc, 700, theta, seed=9, ess_frac=0.3)
    assert abs(total - full.ll) < 1e-12 and np.array_equal(pop, full.pop)
