"""Models beyond the compiled-in evaluators' 8 state components / events / parameters: the reference's DPOMPModel has no
size limit (any closure, hmm_structs.jl:107-116); here such a model is compiled at create time into an evaluator of
exactly its size (rate table -> source, or user source; initial condition and transition rows as constants).

  * age-structured SEIR, 3 age groups: 12 state components, 9 events, 5 parameters -- a RATE TABLE;
  * SIS metapopulation over 10 patches: 20 state components, 20 events -- USER SOURCE (host twin for the oracle);
  * a table with four integer factors in one rate.
"""
import numpy as np
import pytest

AGE_THETA = [0.004, 0.003, 0.005, 0.25, 0.12]


def age_seir(dp):
    """S_a, E_a, I_a, R_a for a = 1..3; infection of group a: beta_a S_a (I_1 + I_2 + I_3); E -> I; I -> R."""
    terms, trans = [], []
    inf = (3, 7, 11)
    for a in range(3):
        s, e, i, r = 4 * a + 1, 4 * a + 2, 4 * a + 3, 4 * a + 4
        for theta, factors, frm, to in ((a + 1, ((s,), inf), s, e), (4, ((e,),), e, i), (5, ((i,),), i, r)):
            terms.append(dp.RateTerm(theta, factors))
            row = [0] * 12
            row[frm - 1], row[to - 1] = -1, 1
            trans.append(row)
    obs = dp.partial_gaussian_obs_model(2.0, seq=inf, y_seq=inf)
    return dp.generate_custom_model("SEIR-3-ages", dp.RateFunction(terms), [40, 0, 1, 0, 40, 0, 0, 0, 20, 0, 0, 0], trans,
                                    obs_model=obs, prior=dp.generate_weak_prior(5))


PATCH_SOURCE = """
__device__ void dpomp_user_rates(const double* theta, const int* x, double* r) {
    int itot = 0;
    for (int i = 0; i < 10; ++i) itot += x[2 * i + 1];
    for (int i = 0; i < 10; ++i) {
        const double S = (double)x[2 * i], I = (double)x[2 * i + 1];
        r[2 * i] = theta[0] * S * (I + theta[2] * (double)(itot - x[2 * i + 1]));   // infection in patch i (own + coupled)
        r[2 * i + 1] = theta[1] * I;                                              // recovery in patch i
    }
}
"""
PATCH_THETA = [0.02, 0.15, 0.05]


def patch_sis(dp):
    trans = []
    for i in range(10):
        a, b = [0] * 20, [0] * 20
        a[2 * i], a[2 * i + 1] = -1, 1
        b[2 * i], b[2 * i + 1] = 1, -1
        trans += [a, b]
    x0 = [12, 0] * 10
    x0[0], x0[1] = 10, 2
    seq = tuple(2 * i + 2 for i in range(10))
    obs = dp.partial_gaussian_obs_model(2.0, seq=seq, y_seq=seq)
    rf = dp.RateSource(PATCH_SOURCE, n_events=20, n_params=3)
    return dp.generate_custom_model("SIS-10-patches", rf, x0, trans, obs_model=obs, prior=dp.generate_weak_prior(3))


def four_factor(dp):
    terms = [dp.RateTerm(1, ((1,), (2,), (3,), (1, 2))), dp.RateTerm(2, ((2,),)), dp.RateTerm(3, ((3,),))]
    return dp.generate_custom_model("four-factors", dp.RateFunction(terms), [30, 2, 3], [[-1, 1, 0], [0, -1, 1], [1, 0, -1]],
                                    prior=dp.generate_weak_prior(3))


def _data(dp, orc, model, theta, T=4, tmax=12.0):
    from helpers import synthetic_data
    return synthetic_data(dp, orc, model, theta, T=T, tmax=tmax)


def test_descriptor_holds_models_of_up_to_32_components(dp, orc):
    model = age_seir(dp)
    y = _data(dp, orc, model, AGE_THETA)
    d = dp.ModelDescriptor(model, y)
    assert (d.n_states, d.n_events, d.n_params) == (12, 9, 5) and d.c.abi_version == 2
    assert d.c.rate[0].n_factors == 2 and d.c.rate[0].factor_mask[1] == (1 << 2) | (1 << 6) | (1 << 10)
    with pytest.raises(ValueError):
        dp.ModelDescriptor(dp.generate_custom_model("too big", dp.RateFunction([dp.RateTerm(1, ((1,),))] * 33), [1] * 33,
                                                    np.zeros((33, 33), dtype=int)), y)


def test_oracle_runs_big_models(dp, orc, tmp_path):
    model = age_seir(dp)
    y = _data(dp, orc, model, AGE_THETA)
    d = dp.ModelDescriptor(model, y)
    r = orc.pf_loglik(d, 300, AGE_THETA, seed=5)
    assert np.isfinite(r.ll) and r.events > 300
    assert np.all(r.pop.sum(axis=1) == 101) and np.all(r.pop[:, 0:4].sum(axis=1) == 41)   # every age group keeps its size
    x = [35, 2, 4, 0, 38, 1, 1, 0, 20, 0, 0, 0]
    want = [AGE_THETA[0] * 35 * 5, AGE_THETA[3] * 2, AGE_THETA[4] * 4, AGE_THETA[1] * 38 * 5, AGE_THETA[3] * 1, AGE_THETA[4] * 1,
            AGE_THETA[2] * 20 * 5, 0.0, 0.0]
    assert np.allclose(orc.rates(d, AGE_THETA, x), want, rtol=1e-15)
    orc.set_user_rates_from_source(PATCH_SOURCE, str(tmp_path))
    pm = patch_sis(dp)
    yd = _data(dp, orc, pm, PATCH_THETA)
    r = orc.pf_loglik(dp.ModelDescriptor(pm, yd), 200, PATCH_THETA, seed=5)
    assert np.isfinite(r.ll) and np.all(r.pop.sum(axis=1) == 120)


def test_generated_evaluators_of_big_models_compile_without_a_gpu(dp, orc, tmp_path):
    """hipRTC runs without a device: the evaluator generated from a 12-component table (constants for the initial
    condition and the transition rows, rates with any number of factors) and a 20-event user source both compile."""
    model = age_seir(dp)
    assert "error" not in dp.check_rate_source(model, _data(dp, orc, model, AGE_THETA)).lower()
    ff = four_factor(dp)
    assert "error" not in dp.check_rate_source(ff, _data(dp, orc, ff, [0.0005, 0.3, 0.2])).lower()
    orc.set_user_rates_from_source(PATCH_SOURCE, str(tmp_path))
    pm = patch_sis(dp)
    assert "error" not in dp.check_rate_source(pm, _data(dp, orc, pm, PATCH_THETA)).lower()


def _check_filter(dp, orc, model, y, theta, pf):
    th2 = np.stack([np.array(theta), np.array(theta) * 1.2], axis=1)
    for P in (3000, 200, 64):   # multi-kernel path (one particle per lane) / whole-filter kernel
        pop = pf.alloc_population(2, P)
        ll = pf.partial_log_likelihood(pop, th2, 1, len(y), seed=21, seg_base=4)
        events = pf.last_event_count()
        pop2 = pf.alloc_population(2, P)
        inc = sum(pf.partial_log_likelihood(pop2, th2, k, k, seed=21, seg_base=4) for k in range(1, len(y) + 1))
        want = 0
        for b in range(2):
            ref = orc.pf_loglik(pf.desc, P, th2[:, b], seed=21, seg=4 + b)
            want += ref.events
            assert ll[b] == ref.ll and abs(inc[b] - ref.ll) < 1e-12, (P, b, ll[b], ref.ll)
            assert np.array_equal(pop.download(b), ref.pop) and np.array_equal(pop2.download(b), ref.pop)
        assert events == want
        pop.close()
        pop2.close()


def _check_simulator(dp, orc, pf, theta):
    counts, times, types, states = pf.simulate_events(theta, n_sims=6, seed=9, seg=2)
    for i in range(6):
        n_ref, t_ref, e_ref, s_ref = orc.simulate_events(pf.desc, theta, seed=9, seg=2, p=i)
        assert counts[i] == n_ref == len(t_ref) and np.array_equal(times[i], t_ref) and np.array_equal(types[i], e_ref)
        assert np.array_equal(states[:, :, i], s_ref)


@pytest.mark.gpu
def test_age_structured_seir_table_matches_oracle(dp, orc):
    model = age_seir(dp)
    y = _data(dp, orc, model, AGE_THETA)
    pf = dp.ParticleFilter(model, y)
    assert pf.evaluator == dp._abi.EVALUATOR_TABLE_COMPILED
    _check_filter(dp, orc, model, y, AGE_THETA, pf)
    _check_simulator(dp, orc, pf, AGE_THETA)
    with pytest.raises(dp.DpompError):
        pf.set_propagate_variant(32)      # the compaction schedules are built for models of up to 8 components
    pf.close()
    with pytest.raises(dp.DpompError) as ei:   # no interpreter for a model of this size
        dp.ParticleFilter(model, y, jit=False)
    assert "compiled at create time" in str(ei.value)


@pytest.mark.gpu
def test_twenty_event_user_source_matches_oracle(dp, orc, tmp_path):
    orc.set_user_rates_from_source(PATCH_SOURCE, str(tmp_path))
    model = patch_sis(dp)
    y = _data(dp, orc, model, PATCH_THETA)
    pf = dp.ParticleFilter(model, y)
    assert pf.evaluator == dp._abi.EVALUATOR_SOURCE
    _check_filter(dp, orc, model, y, PATCH_THETA, pf)
    _check_simulator(dp, orc, pf, PATCH_THETA)
    pf.close()


@pytest.mark.gpu
def test_rate_with_four_factors_matches_oracle(dp, orc):
    model = four_factor(dp)
    theta = [0.0005, 0.3, 0.2]
    y = _data(dp, orc, model, theta)
    pf = dp.ParticleFilter(model, y)
    assert pf.evaluator == dp._abi.EVALUATOR_TABLE_COMPILED
    _check_filter(dp, orc, model, y, theta, pf)
    pf.close()
