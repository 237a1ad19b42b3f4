"""Parity of the HIP path (through the C ABI) with the CPU oracle's SPEC mode.

Bar: integer work (states, offspring boundaries, event counts) bit-exact; the log-likelihood,
being a fixed sequence of IEEE operations on both sides, is also compared for equality, with a
1e-12 absolute tolerance stated as the fallback contract.
"""
import math
import os

import numpy as np
import pytest

from oracle import exact_sis

pytestmark = pytest.mark.gpu

LL_TOL = 1e-12


def _assert_ll(got, ref):
    if math.isinf(ref):
        assert got == ref
    else:
        assert abs(got - ref) <= LL_TOL, (got, ref)


@pytest.fixture(scope="module")
def sis(dp, pooley):
    pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), pooley)
    yield pf
    pf.close()


def test_device_numerics_match_oracle_bitwise(sis, orc):
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.random(4096), 1.0 - rng.random(512) * 1e-6, rng.random(512) * 1e-12, [2.0 ** -53, 1 - 2.0 ** -53]])
    u, lg, ex = sis.selftest_numerics(x, seed=77)
    ref_u = np.array([orc.uniforms(77, 0, i, 0, orc.DOMAIN_PROPAGATE, 1)[0] for i in range(len(x))])
    assert np.array_equal(u, ref_u)
    assert np.array_equal(lg, orc.dlog(x))
    assert np.array_equal(ex, orc.dexp(-37.0 * x))


# 0 = default: whole-filter kernel for P <= 256; 33 = compaction kernel with small filters laid end to end over the workgroups
@pytest.mark.parametrize("rpl", [0, 1, 2, 4, 16, 32, 33])
@pytest.mark.parametrize("P", [1, 63, 64, 65, 200, 256, 257, 511, 512, 513, 1000, 1024, 1025, 4097])
def test_sis_full_filter_matches_oracle(sis, orc, P, rpl):
    """Every propagation-kernel variant (one particle per lane; LDS queue with 2 / 4 per lane; workgroup compaction,
    also with filters laid end to end)."""
    theta = [0.003, 0.1]
    sis.set_propagate_variant(rpl)
    pop = sis.alloc_population(1, P)
    ll = sis.partial_log_likelihood(pop, theta, 1, sis.n_obs, seed=11)
    ref = orc.pf_loglik(sis.desc, P, theta, seed=11)
    assert np.array_equal(pop.download(0), ref.pop)
    assert sis.last_event_count() == ref.events
    assert sis.last_status[0] == ref.status
    _assert_ll(ll[0], ref.ll)
    assert ll[0] == ref.ll  # in practice exact
    sis.set_propagate_variant(0)


def test_step_by_step_states_logw_and_offspring(sis, orc, dp):
    """Observation by observation: propagated-and-resampled states, log-weights and offspring
    boundaries all equal the oracle's."""
    P, theta, seed = 3000, [0.0032, 0.09], 5
    pop = sis.alloc_population(1, P)
    ref_pop = np.zeros((P, 2), dtype=np.int64)
    for k in range(1, sis.n_obs + 1):
        ll = sis.partial_log_likelihood(pop, theta, k, k, seed=seed)
        r = orc.pf_partial(sis.desc, ref_pop, theta, k, k, seed=seed, trace=True)
        assert np.array_equal(pop.download(0), ref_pop), f"states differ after observation {k}"
        assert np.array_equal(pop.debug(dp.DBG_LOGW), r.logw)
        if k < sis.n_obs:
            assert np.array_equal(pop.debug(dp.DBG_OFFSPRING_END).astype(np.int64), r.offspring_end)
        assert ll[0] == r.ll


def test_incremental_equals_full(sis):
    P, theta = 2048, [0.003, 0.1]
    pop = sis.alloc_population(1, P)
    full = sis.partial_log_likelihood(pop, theta, 1, sis.n_obs, seed=3)[0]
    final = pop.download(0).copy()
    parts = [sis.partial_log_likelihood(pop, theta, k, k, seed=3)[0] for k in range(1, sis.n_obs + 1)]
    assert np.array_equal(pop.download(0), final)
    assert abs(sum(parts) - full) < 1e-12


def test_batched_segments_are_independent_and_partition_invariant(sis, orc):
    """B filters in one launch == B single launches with seg_base: the RNG is keyed by the global
    segment index, so sharding segments over GPUs cannot change a result."""
    P = 1500
    thetas = np.array([[0.003, 0.0025, 0.004, 0.0032], [0.1, 0.08, 0.15, 0.1]])
    batch = sis.loglik(thetas, P, seed=9, seg_base=10)
    for b in range(4):
        single = sis.loglik(thetas[:, b], P, seed=9, seg_base=10 + b)[0]
        ref = orc.pf_loglik(sis.desc, P, thetas[:, b], seed=9, seg=10 + b).ll
        assert batch[b] == single == ref


def test_determinism_same_seed_same_bits(sis):
    a = sis.loglik([0.003, 0.1], 10000, seed=123)
    b = sis.loglik([0.003, 0.1], 10000, seed=123)
    c = sis.loglik([0.003, 0.1], 10000, seed=124)
    assert a[0] == b[0] and a[0] != c[0]


CASES = [
    ("SI", [100, 1], False, [0.003]),
    ("SI", [100, 1], True, [0.3]),
    ("SIR", [100, 1, 0], False, [0.003, 0.1]),
    ("SIR", [100, 1, 0], True, [0.3, 0.1]),
    ("SIS", [100, 1], True, [0.3, 0.1]),
    ("SEI", [100, 0, 1], False, [0.005, 0.2]),
    ("SEI", [100, 0, 1], True, [0.5, 0.2]),
    ("SEIR", [100, 0, 1, 0], False, [0.005, 0.2, 0.1]),
    ("SEIR", [100, 0, 1, 0], True, [0.5, 0.2, 0.1]),
    ("SEIS", [100, 0, 1], False, [0.005, 0.2, 0.1]),
    ("SEIS", [100, 0, 1], True, [0.5, 0.2, 0.1]),
    ("LOTKA", [70, 70], False, [0.5, 0.0025, 0.3]),
    ("ROSSMAC", [100, 10, 200, 20], False, [0.02, 0.05, 0.04, 0.1]),
]


def _synthetic_data(dp, orc, model, theta, T=4, tmax=20.0, seed=99):
    """Observations simulated with the ORACLE's forward simulator (the default flow of the
    reference: gillespie_sim with dmy_obs_fn, y.val = state)."""
    times = [tmax * (i + 1) / T for i in range(T)]
    blank = [dp.Observation(t, 1, 1.0, [0] * model.n_states) for t in times]
    d = dp.ModelDescriptor(model, blank)
    traj = orc.simulate(d, 1, theta, seed=seed)  # [T][S][1]
    return [dp.Observation(t, 1, 1.0, [int(v) for v in traj[i, :, 0]]) for i, t in enumerate(times)]


@pytest.mark.parametrize("name,x0,fd,theta", CASES)
@pytest.mark.parametrize("generic,rpl", [(False, 0), (True, 0), (False, 1), (True, 1), (False, 4), (True, 2), (False, 16), (True, 32),
                                         (False, 33), (True, 33)])
def test_catalogue_models_match_oracle(dp, orc, name, x0, fd, theta, generic, rpl):
    """Every catalogue model, compiled built-in evaluator and table-driven generic evaluator,
    equals the oracle (which only knows the generic table)."""
    model = dp.generate_model(name, x0, freq_dep=fd)
    y = _synthetic_data(dp, orc, model, theta)
    pf = dp.ParticleFilter(model, y, force_generic=generic)
    pf.set_propagate_variant(rpl)
    P = 200 if rpl in (0, 33) else 777   # 200: the reference's default npf, served by the whole-filter kernel
    pop = pf.alloc_population(1, P)
    ll = pf.partial_log_likelihood(pop, theta, 1, len(y), seed=42)
    ref = orc.pf_loglik(pf.desc, P, theta, seed=42)
    assert np.array_equal(pop.download(0), ref.pop)
    assert pf.last_event_count() == ref.events
    assert ll[0] == ref.ll
    pf.close()


def test_custom_model_with_t0_index_and_unweighted_observation(dp, orc, make_obs):
    """A hand-built model (the reference's custom_sim testset, test/runtests.jl:74-100, extended with
    an unknown initial time and an obs_id = 0 row that must neither weight nor resample)."""
    rf = dp.RateFunction([dp.RateTerm(1, ((1,), (2,))), dp.RateTerm(2, ((2,),))])
    model = dp.generate_custom_model("SIS", rf, [100, 1], [[-1, 1], [1, -1]], t0_index=3)
    y = make_obs([20.0, 30.0, 40.0, 60.0], [[0, 18], [0, 0], [0, 65], [0, 70]], obs_ids=[1, 0, 1, 1])
    pf = dp.ParticleFilter(model, y)
    theta = [0.003, 0.1, 4.5]
    P = 2000
    pop = pf.alloc_population(1, P)
    ll = pf.partial_log_likelihood(pop, theta, 1, 4, seed=8)
    ref = orc.pf_loglik(pf.desc, P, theta, seed=8, trace=True)
    assert ref.per_obs_ll[1] == 0.0
    assert np.array_equal(pop.download(0), ref.pop)
    assert ll[0] == ref.ll
    pf.close()


def test_event_cap_flags_and_zeroes_the_weight(dp, orc, pooley):
    model = dp.generate_model("SIS", [100, 1])
    model.max_events_per_interval = 150  # intervals 2..5 need ~270 events per particle
    pf = dp.ParticleFilter(model, pooley)
    pf.set_propagate_variant(4)
    ll = pf.loglik([0.003, 0.1], 512, seed=2)
    ref = orc.pf_loglik(pf.desc, 512, [0.003, 0.1], seed=2)
    assert pf.last_status[0] & dp.ST_EVENT_CAP and ref.status & dp.ST_EVENT_CAP
    assert (ll[0] == ref.ll) or (math.isinf(ll[0]) and math.isinf(ref.ll))
    assert pf.last_event_count() == ref.events
    pf.close()


def test_impossible_data_gives_minus_inf_not_an_error(dp, orc, make_obs):
    model = dp.generate_model("SIS", [100, 1])
    y = make_obs([20.0, 40.0], [[0, 5000], [0, 5000]])  # far outside the reachable states: exp underflows
    pf = dp.ParticleFilter(model, y)
    ll = pf.loglik([0.003, 0.1], 256, seed=1)
    ref = orc.pf_loglik(pf.desc, 256, [0.003, 0.1], seed=1)
    # max-shifted weights cannot underflow: the filter stays finite where the reference returns -Inf
    assert ll[0] == ref.ll and math.isfinite(ll[0])
    pf.close()


def test_population_gather_upload_download(sis):
    P = 1024
    pop = sis.alloc_population(3, P)
    thetas = np.array([[0.003, 0.0025, 0.004], [0.1, 0.08, 0.15]])
    sis.partial_log_likelihood(pop, thetas, 1, 2, seed=4)
    before = [pop.download(b).copy() for b in range(3)]
    pop.gather([2, 0, 0])
    assert np.array_equal(pop.download(0), before[2])
    assert np.array_equal(pop.download(1), before[0])
    assert np.array_equal(pop.download(2), before[0])
    pop.upload(1, before[1])
    assert np.array_equal(pop.download(1), before[1])


def test_call_sequence_errors(dp, sis):
    pop = sis.alloc_population(1, 64)
    with pytest.raises(dp.DpompError):
        sis.partial_log_likelihood(pop, [0.003, 0.1], 2, 3)  # never initialised
    with pytest.raises(dp.DpompError):
        sis.partial_log_likelihood(pop, [0.003, 0.1], 0, 3)
    with pytest.raises(ValueError):
        sis.loglik([0.003], 64)


def test_simulate_matches_oracle(dp, orc, pooley):
    model = dp.generate_model("SEIR", [100, 0, 1, 0])
    blank = [dp.Observation(t, 1, 1.0, [0, 0, 0, 0]) for t in (20.0, 40.0, 60.0, 80.0, 100.0)]
    pf = dp.ParticleFilter(model, blank)
    theta = [0.005, 0.2, 0.1]
    got = pf.simulate(theta, n_sims=300, seed=6)
    ref = orc.simulate(pf.desc, 300, theta, seed=6)
    assert np.array_equal(got[0], ref)
    pf.close()


# ---- full size (the headline benchmark: Pooley SIS, 2^20 particles): properties that need no oracle ----
def test_full_size_pooley_2pow20(sis, dp):
    P = 1 << 20
    theta = [0.003, 0.1]
    pop = sis.alloc_population(1, P)
    lls = []
    for k in range(1, sis.n_obs + 1):
        lls.append(sis.partial_log_likelihood(pop, theta, k, k, seed=2026)[0])
        if k < sis.n_obs:
            N = pop.debug(dp.DBG_OFFSPRING_END).astype(np.int64)
            assert N[-1] == P and np.all(np.diff(N) >= 0) and N[0] >= 0  # counts sum to P, ancestors sorted
            states = pop.download(0)
            assert np.all(states.sum(axis=1) == 101)  # closed population: S + I conserved
    exact, incs = exact_sis.exact_sis_loglik(theta, [100, 1], exact_sis.POOLEY_TIMES, exact_sis.POOLEY_CASES)
    # Monte-Carlo tolerance: at P = 2^20 the per-observation sd is ~4e-3/sqrt(P/1024) ~ 1e-3 (sd of ll at
    # P = 1024 is 0.12 over 5 observations); 0.02 is > 5 sd of the total.
    assert abs(sum(lls) - exact) < 0.02, (sum(lls), exact)
    for got, want in zip(lls, incs):
        assert abs(got - want) < 0.02
    # the same call twice gives the same bits; one call over 1..T equals the increments
    full = sis.partial_log_likelihood(pop, theta, 1, sis.n_obs, seed=2026)[0]
    assert abs(full - sum(lls)) < 1e-12


@pytest.mark.parametrize("pattern", ["low_weight_run", "alternating", "single_survivor"])
def test_resampling_with_crafted_weights(dp, orc, make_obs, pattern):
    """Uploaded populations with engineered weights: a long run of zero-weight parents (the output
    window of one workgroup then spans more parent tiles than it stages in LDS -> fallback search),
    alternating zero/non-zero parents, and a single surviving particle."""
    model = dp.generate_model("SIS", [100, 1])
    y = make_obs([1.0, 1.0 + 1e-9, 2.0], [[0, 50], [0, 50], [0, 50]])
    pf = dp.ParticleFilter(model, y)
    P = 20000
    I = np.full(P, 50, dtype=np.int64)
    if pattern == "low_weight_run":
        I[1:9000] = 0          # > 8 tiles of parents with weight exp(-2500/8) -> 0 in fixed point
        I[9000:9100] = 47      # a few light ones
    elif pattern == "alternating":
        I[::2] = 0
        I[1::6] = 52
    else:
        I[:] = 0
        I[12345] = 50
    start = np.stack([101 - I, I], axis=1)
    pop = pf.alloc_population(1, P)
    pop.upload(0, start)
    theta = [1e-7, 1e-7]  # essentially no events in the 1e-9 interval
    ll = pf.partial_log_likelihood(pop, theta, 2, 2, seed=31)
    ref_pop = start.copy()
    ref = orc.pf_partial(pf.desc, ref_pop, theta, 2, 2, seed=31, trace=True)
    assert np.array_equal(pop.debug(dp.DBG_OFFSPRING_END).astype(np.int64), ref.offspring_end)
    assert np.array_equal(pop.download(0), ref_pop)
    assert ll[0] == ref.ll
    if pattern == "single_survivor":
        assert np.all(pop.download(0)[:, 1] == 50)
    pf.close()


def test_many_segments_small_filters(dp, orc, pooley):
    """SMC^2 shape: many segments, few particles each (B = 300, P = 200, the reference's default npf)."""
    pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), pooley)
    rng = np.random.default_rng(1)
    B = 300
    theta = np.stack([rng.uniform(0.002, 0.004, B), rng.uniform(0.05, 0.2, B)])
    got = pf.loglik(theta, 200, seed=5)
    for b in rng.integers(0, B, 12):
        assert got[b] == orc.pf_loglik(pf.desc, 200, theta[:, b], seed=5, seg=int(b)).ll
    pf.close()


@pytest.mark.parametrize("rs_type", [2, 3])
@pytest.mark.parametrize("P", [1, 777, 1024, 5000])
def test_stratified_and_multinomial_resamplers_match_oracle(dp, orc, pooley, rs_type, P):
    """rs_type 2 (stratified, Kitagawa) and 3 (multinomial): the reference's rsp_stratified /
    rsp_multinomial throw as shipped; device == oracle SPEC on their documented intent."""
    pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), pooley, rs_type=rs_type)
    pop = pf.alloc_population(2, P)
    theta = np.array([[0.003, 0.0028], [0.1, 0.09]])
    ll = pf.partial_log_likelihood(pop, theta, 1, pf.n_obs, seed=17, seg_base=3)
    for b in range(2):
        ref = orc.pf_loglik(pf.desc, P, theta[:, b], seed=17, seg=3 + b, rs_type=rs_type)
        assert np.array_equal(pop.download(b), ref.pop)
        assert ll[b] == ref.ll
    pf.close()


def test_resampler_choice_through_the_closure(dp, orc, pooley):
    f = dp.get_particle_filter_lpdf(dp.generate_model("SIS", [100, 1]), pooley, np=512, rs_type=2)
    got = f([0.003, 0.1], seed=5)
    assert got == orc.pf_loglik(f.filter.desc, 512, [0.003, 0.1], seed=5, rs_type=2).ll
    with pytest.raises(dp.DpompError):
        dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), pooley, rs_type=7)


GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name,x0,theta,csv", [("SEIR", [100, 0, 1, 0], [0.005, 0.2, 0.1], "seir_synthetic.csv"),
                                               ("LOTKA", [70, 70], [0.5, 0.0025, 0.3], "lotka_synthetic.csv")])
def test_baseline_config_shapes_at_scale(dp, orc, name, x0, theta, csv):
    """The SEIR / Lotka-Volterra benchmark models on their committed synthetic data: exact parity at P = 512, then a large batched run
    (4 x 2^16) whose estimates must agree with each other within Monte-Carlo scatter and raise no status flag."""
    y = dp.get_observations(os.path.join(GOLDEN_DIR, csv))
    pf = dp.ParticleFilter(dp.generate_model(name, x0), y)
    small = pf.loglik(theta, 512, seed=6)[0]
    assert small == orc.pf_loglik(pf.desc, 512, theta, seed=6).ll
    big = pf.loglik(np.tile(np.array(theta)[:, None], (1, 4)), 1 << 16, seed=7)
    assert np.all(np.isfinite(big)) and np.all(pf.last_status == 0)
    assert big.max() - big.min() < 0.25, big   # four independent 65536-particle estimates of the same quantity
    assert abs(big.mean() - small) < 4.0        # and the 512-particle estimate is in the same place (its own sd is ~1)
    pf.close()


def test_single_observation_no_resampling(dp, orc, make_obs):
    """T = 1: propagate, weight, log mean weight -- and never resample (obs_i < length(obs_data) is false)."""
    pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), make_obs([30.0], [[0, 40]]))
    pop = pf.alloc_population(1, 1500)
    ll = pf.partial_log_likelihood(pop, [0.003, 0.1], 1, 1, seed=2)
    ref = orc.pf_loglik(pf.desc, 1500, [0.003, 0.1], seed=2)
    assert ll[0] == ref.ll and np.array_equal(pop.download(0), ref.pop)
    pf.close()


def test_zero_length_interval_and_zero_rates(dp, orc, make_obs):
    """Two observations at the same time (nothing can happen in between) and a parameter vector with all
    rates zero (total rate 0: particles idle without drawing, hmm_particle_filter.jl:22)."""
    y = make_obs([10.0, 10.0, 25.0], [[0, 3], [0, 4], [0, 9]])
    pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), y)
    for theta in ([0.003, 0.1], [0.0, 0.0]):
        pop = pf.alloc_population(1, 640)
        ll = pf.partial_log_likelihood(pop, theta, 1, 3, seed=5)
        ref = orc.pf_loglik(pf.desc, 640, theta, seed=5)
        assert ll[0] == ref.ll and np.array_equal(pop.download(0), ref.pop)
        assert pf.last_event_count() == ref.events
    assert pf.last_event_count() == 0   # theta = 0: no events at all, every particle still at [100, 1]
    assert np.all(pop.download(0) == [100, 1])
    pf.close()


def test_maximum_model_size_through_the_generic_evaluator(dp, orc, make_obs):
    """S = 8 states, E = 8 events, K = 8 parameters (the ABI's limits): a ring of compartments with every rate
    form the table offers (plain, product, frequency-dependent, ratio)."""
    T = dp.RateTerm
    terms = [T(1, ((1,), (2,))), T(2, ((2,),)), T(3, ((3,), (4,)), tuple(range(1, 9))), T(4, ((4,),)),
             T(5, ((5,), (6,)), (5, 6), "ratio"), T(6, ((6, 7),)), T(7, ((7,),)), T(8, ((8,), (1,)), (1, 2, 3))]
    trans = np.zeros((8, 8), dtype=int)
    for e in range(8):
        trans[e, e] = -1
        trans[e, (e + 1) % 8] = 1
    x0 = [40, 30, 20, 25, 35, 15, 10, 30]
    model = dp.generate_custom_model("RING8", dp.RateFunction(terms), x0, trans,
                                     obs_model=dp.partial_gaussian_obs_model(3.0, seq=(2, 5, 8), y_seq=(1, 3)))
    y = make_obs([2.0, 4.0, 6.0], [[40, 0, 30], [45, 0, 28], [50, 0, 25]])
    theta = [0.002, 0.1, 0.5, 0.15, 0.3, 0.05, 0.2, 0.004]
    pf = dp.ParticleFilter(model, y)
    pop = pf.alloc_population(2, 900)
    th2 = np.stack([theta, [t * 1.1 for t in theta]], axis=1)
    ll = pf.partial_log_likelihood(pop, th2, 1, 3, seed=21)
    for b in range(2):
        ref = orc.pf_loglik(pf.desc, 900, th2[:, b], seed=21, seg=b)
        assert ll[b] == ref.ll and np.array_equal(pop.download(b), ref.pop)
    # host evaluation of the same table agrees with the oracle's interpreter
    assert np.array_equal(model.rate_function(np.zeros(8), theta, x0), orc.rates(pf.desc, theta, x0))
    pf.close()


@pytest.mark.parametrize("rs_type", [1, 2])
def test_three_million_particles_unfused_scan_path(dp, orc, make_obs, rs_type):
    """P = 3 * 2^20 (3072 tiles > 2048): the tile-sum scan runs as its own kernel and k_offsets<false> consumes its
    prefixes.  Low rates keep the oracle fast (~0.3 events per particle and interval); parity stays bit-exact."""
    y = make_obs([1.0, 2.0, 3.0], [[0, 1], [0, 2], [0, 1]])
    pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), y, rs_type=rs_type)
    P = 3 * (1 << 20) + 17
    theta = [0.003, 0.2]
    pop = pf.alloc_population(1, P)
    ll = pf.partial_log_likelihood(pop, theta, 1, 3, seed=4)
    ref = orc.pf_loglik(pf.desc, P, theta, seed=4, rs_type=rs_type, nthreads=8)
    assert ll[0] == ref.ll
    assert np.array_equal(pop.download(0), ref.pop)
    assert pf.last_event_count() == ref.events
    pf.close()


@pytest.mark.parametrize("rs_type", [1, 2, 3])
@pytest.mark.parametrize("frac", [0.3, 0.6, 1.5])
def test_ess_adaptive_resampling_matches_oracle(dp, orc, pooley, rs_type, frac):
    """ESS-adaptive mode: segments of one batch take different resample/keep decisions, weights are carried on
    the device; result and final populations equal the oracle's, for all three resamplers."""
    pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), pooley, rs_type=rs_type, essc=frac, ess_adaptive=True)
    P = 2500
    theta = np.array([[0.003, 0.0028, 0.0034, 0.002], [0.1, 0.09, 0.12, 0.2]])
    pop = pf.alloc_population(4, P)
    ll = pf.partial_log_likelihood(pop, theta, 1, pf.n_obs, seed=23, seg_base=1)
    kept = 0
    for b in range(4):
        ref = orc.pf_loglik(pf.desc, P, theta[:, b], seed=23, seg=1 + b, rs_type=rs_type, ess_frac=frac)
        plain = orc.pf_loglik(pf.desc, P, theta[:, b], seed=23, seg=1 + b, rs_type=rs_type)
        assert ll[b] == ref.ll and np.array_equal(pop.download(b), ref.pop)
        kept += ref.ll != plain.ll
    if frac == 1.5:
        assert kept == 0          # threshold above P: identical to the default path
    if frac == 0.3:
        assert kept >= 1          # at least one segment really skipped a resampling
    pf.close()


def test_ess_adaptive_state_survives_incremental_calls(dp, orc, pooley):
    pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), pooley, essc=0.3, ess_adaptive=True)
    P, theta = 1800, [0.003, 0.1]
    pop = pf.alloc_population(1, P)
    parts = [pf.partial_log_likelihood(pop, theta, k, k, seed=31)[0] for k in range(1, pf.n_obs + 1)]
    ref = orc.pf_loglik(pf.desc, P, theta, seed=31, ess_frac=0.3)
    assert abs(sum(parts) - ref.ll) < 1e-12 and np.array_equal(pop.download(0), ref.pop)
    full = pf.loglik(theta, P, seed=31)[0]
    assert full == ref.ll
    pf.set_ess_threshold(0.0)   # back to the reference's behaviour
    assert pf.loglik(theta, P, seed=31)[0] == orc.pf_loglik(pf.desc, P, theta, seed=31).ll
    pf.close()


@pytest.mark.parametrize("rs_type,ess", [(1, 0.0), (2, 0.0), (3, 0.0), (1, 0.5)])
@pytest.mark.parametrize("theta", [(0.003, 0.1), (0.0025, 0.08)])
def test_unbiasedness_against_the_exact_likelihood_at_scale(dp, pooley, rs_type, ess, theta):
    """2000 independent 2048-particle filters in one batched call: the mean of exp(ll_hat - exact) must be 1 within
    4 standard errors (~1 %), and mean(ll_hat) sits below exact by about Var/2 (Jensen) -- a far tighter
    distribution-level pin than the CPU suite can afford, for every resampler and the ESS-adaptive mode."""
    exact = exact_sis.exact_sis_loglik(theta, [100, 1], exact_sis.POOLEY_TIMES, exact_sis.POOLEY_CASES)[0]
    pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), pooley, rs_type=rs_type, essc=ess, ess_adaptive=ess > 0)
    B = 2000
    th = np.tile(np.array(theta)[:, None], (1, B))
    lls = pf.loglik(th, 2048, seed=20261003 + rs_type)
    assert np.all(np.isfinite(lls)) and np.all(pf.last_status == 0)
    ratio = np.exp(lls - exact)
    se = ratio.std(ddof=1) / math.sqrt(B)
    assert abs(ratio.mean() - 1.0) < 4 * se, (ratio.mean(), se)
    assert se < 0.004
    assert abs((exact - lls.mean()) - lls.var() / 2) < 4 * lls.std() / math.sqrt(B) + 0.002
    pf.close()


def test_t0_index_differs_per_segment(dp, orc, make_obs):
    """Unknown initial time (DPOMPModel.t0_index): every segment of a batch starts its first interval at its own
    theta[t0_index] (hmm_particle_filter.jl:45)."""
    rf = dp.RateFunction([dp.RateTerm(1, ((1,), (2,))), dp.RateTerm(2, ((2,),))])
    model = dp.generate_custom_model("SIS", rf, [100, 1], [[-1, 1], [1, -1]], t0_index=3, prior=dp.generate_weak_prior(3))
    y = make_obs([20.0, 40.0, 60.0], [[0, 18], [0, 65], [0, 70]])
    pf = dp.ParticleFilter(model, y)
    theta = np.array([[0.003, 0.003, 0.0031], [0.1, 0.1, 0.11], [0.0, 7.5, 15.0]])
    pop = pf.alloc_population(3, 1200)
    ll = pf.partial_log_likelihood(pop, theta, 1, 3, seed=14)
    for b in range(3):
        ref = orc.pf_loglik(pf.desc, 1200, theta[:, b], seed=14, seg=b)
        assert ll[b] == ref.ll and np.array_equal(pop.download(b), ref.pop)
    assert ll[0] != ll[1]
    pf.close()


def test_batched_state_only_simulation(dp, orc):
    model = dp.generate_model("LOTKA", [70, 70])
    blank = dp.generate_observations(10.0, 5, 2)
    pf = dp.ParticleFilter(model, blank)
    theta = np.array([[0.5, 0.45], [0.0025, 0.003], [0.3, 0.35]])
    got = pf.simulate(theta, n_sims=257, seed=8, seg_base=4)       # [B, T, S, n]
    for b in range(2):
        assert np.array_equal(got[b], orc.simulate(pf.desc, 257, theta[:, b], seed=8, seg=4 + b))
    pf.close()


def test_small_filter_kernel_incremental_unweighted_and_batched(dp, orc, make_obs):
    """The whole-filter-per-workgroup path (P <= 256): incremental calls resuming from stored populations, an
    obs_id = 0 observation in the middle, t0_index, many segments; equal to the oracle and to the multi-kernel path."""
    rf = dp.RateFunction([dp.RateTerm(1, ((1,), (2,))), dp.RateTerm(2, ((2,),))])
    model = dp.generate_custom_model("SIS", rf, [100, 1], [[-1, 1], [1, -1]], t0_index=3, prior=dp.generate_weak_prior(3))
    y = make_obs([20.0, 30.0, 40.0, 60.0, 80.0], [[0, 18], [0, 0], [0, 65], [0, 70], [0, 66]], obs_ids=[1, 0, 1, 1, 1])
    pf = dp.ParticleFilter(model, y)
    B, P = 37, 200
    rng = np.random.default_rng(3)
    theta = np.stack([rng.uniform(0.0025, 0.0035, B), rng.uniform(0.08, 0.12, B), rng.uniform(0.0, 10.0, B)])
    pop = pf.alloc_population(B, P)
    a = pf.partial_log_likelihood(pop, theta, 1, 2, seed=6, seg_base=2)
    b = pf.partial_log_likelihood(pop, theta, 3, 5, seed=6, seg_base=2)
    assert "k_filter_small" in pf.kernel_stats()
    pf.set_propagate_variant(1)                       # the multi-kernel path on a second population
    pop2 = pf.alloc_population(B, P)
    full = pf.partial_log_likelihood(pop2, theta, 1, 5, seed=6, seg_base=2)
    for s in rng.integers(0, B, 6):
        ref = orc.pf_loglik(pf.desc, P, theta[:, s], seed=6, seg=2 + int(s))
        assert abs((a[s] + b[s]) - ref.ll) < 1e-12 and full[s] == ref.ll
        assert np.array_equal(pop.download(int(s)), ref.pop) and np.array_equal(pop2.download(int(s)), ref.pop)
    pf.close()


@pytest.mark.parametrize("rpl", [1, 32])
def test_lane_utilisation_statistic_matches_oracle_event_counts(sis, orc, rpl):
    """dpomp_get_lane_utilisation: sum of per-particle events / (64 * sum over waves of the wave's maximum), from
    either schedule, against the same figure computed from the oracle's per-particle event counts."""
    theta, P = [0.003, 0.1], 4097
    sis.set_propagate_variant(rpl)
    sis.loglik(np.array(theta)[:, None], P, seed=3)
    ref = orc.pf_loglik(sis.desc, P, theta, seed=3, trace=True)
    want = []
    for k in range(sis.n_obs):
        n = np.zeros(((P + 63) // 64) * 64, dtype=np.int64)
        n[:P] = ref.particle_events[k]
        want.append(n.sum() / (64.0 * n.reshape(-1, 64).max(axis=1).sum()))
    assert np.allclose(sis.lane_utilisation(), want, rtol=1e-12)
    sis.set_propagate_variant(0)


def test_automatic_schedule_reads_lane_utilisation_and_keeps_results(dp, pooley):
    """Chip-filling launch (2^19 particles = 8192 waves): the automatic schedule runs compaction unless the interval
    measured a lane utilisation >= 0.9 last time.  Pooley: 0.30 in the first interval (most epidemics die out at
    once), 0.82-0.87 afterwards.  Results do not depend on the schedule."""
    pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), pooley)
    theta = np.array([[0.003], [0.1]])
    assert np.all(pf.lane_utilisation() == -1.0)
    a = pf.loglik(theta, 1 << 19, seed=21)
    u = pf.lane_utilisation()
    assert u[0] < 0.4 and np.all(u[1:] > 0.75) and np.all(u[1:] < 0.9), u
    b = pf.loglik(theta, 1 << 19, seed=21)
    pf.set_propagate_variant(1)
    c = pf.loglik(theta, 1 << 19, seed=21)
    assert a[0] == b[0] == c[0]
    assert np.array_equal(pf.lane_utilisation(), u)
    pf.close()


def test_head_start_of_the_last_workgroups_is_scheduling_only(dp, pooley, monkeypatch):
    """From its second call on a handle the compaction kernel runs the workgroups that are dispatched last at raised
    wave priority for part of their event chain (from the chain length the previous call measured).  That may only
    move time: log-likelihood, event count and final population equal the first call's (no head start: nothing
    measured yet), those of a handle with the head start switched off, and the oracle's first observation."""
    theta = np.array([[0.003, 0.0032], [0.1, 0.1]])
    P = 1 << 18

    def run(pf, n):
        out = []
        pop = pf.alloc_population(2, P)
        for _ in range(n):
            ll = pf.partial_log_likelihood(pop, theta, 1, pf.n_obs, seed=77, seg_base=5)
            out.append((ll.copy(), pf.last_event_count(), pop.download(0).copy(), pop.download(1).copy()))
        return out

    pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), pooley)
    with_hs = run(pf, 3)
    pf.close()
    monkeypatch.setenv("DPOMP_HEAD_START", "0")
    pf_off = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), pooley)
    without = run(pf_off, 2)
    pf_off.close()
    for r in with_hs[1:] + without:
        assert np.array_equal(r[0], with_hs[0][0]) and r[1] == with_hs[0][1]
        assert np.array_equal(r[2], with_hs[0][2]) and np.array_equal(r[3], with_hs[0][3])


def test_event_loop_division_is_the_ieee_quotient(sis):
    """The event loop divides log(u) by the rate total with the compiler's Newton-Markstein sequence minus its range
    handling (dpomp_numerics.h: div_in_range).  Inside its range it must return the correctly rounded quotient --
    numpy's -- for every operand pair, including totals at the edges of the range and quotients that are hard to
    round; outside it reports itself out of range."""
    rng = np.random.default_rng(11)
    n = 1 << 20
    u = (2 * rng.integers(0, 1 << 52, n, dtype=np.uint64) + 1).astype(np.float64) * 2.0 ** -53
    a = np.log(u)                                            # numerators as the loop sees them: -37 .. -2^-53
    b = np.exp(rng.uniform(np.log(1e-12), np.log(1e12), n))  # rate totals over 24 decades
    b[: n // 8] = rng.integers(1, 1 << 20, n // 8) * 0.003   # and lattice-like totals (theta * S * I)
    edge = np.array([2.0 ** -200, np.nextafter(2.0 ** 200, 0), 1.0, 3.0, 2.0 ** -199 * 1.5, 2.0 ** 199 * 1.999])
    a = np.concatenate([a, [-1.0, -36.7, -2.0 ** -53, -1.0 / 3.0, -0.1, -1e-10]])
    b = np.concatenate([b, edge])
    fast, plain = sis.selftest_division(a, b)
    assert np.array_equal(plain, a / b)
    assert np.array_equal(fast, a / b)
    out = np.array([0.0, -1.0, 2.0 ** -201, 2.0 ** 200, np.inf, np.nan, 5e-324])
    fast, plain = sis.selftest_division(-np.ones(out.size), out)
    assert np.all(np.isnan(fast))
    with np.errstate(divide="ignore", invalid="ignore"):
        want = -1.0 / out
    assert np.array_equal(plain, want, equal_nan=True)


def test_sampled_profiling_brackets_every_nth_call(dp, pooley):
    """dpomp_set_profiling(h, n): kernels of every n-th filter call are timed; statistics are resolved lazily when
    they are read and survive switching profiling off."""
    pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), pooley)
    theta = np.array([[0.003], [0.1]])
    pf.loglik(theta, 4096, seed=1)
    assert all(n == 0 for n, _ in pf.kernel_stats().values())
    pf.set_profiling(2)
    for s in range(4):
        pf.loglik(theta, 4096, seed=s)
    st = pf.kernel_stats()
    assert st["k_propagate"][0] == 2 * pf.n_obs and st["k_wsum"][0] == 2 * pf.n_obs      # calls 0 and 2 of the four
    # the resampling gather runs inside the next propagation of the call; only a call that ENDS on a resampling
    # observation needs the separate pass
    assert st["k_offsets"][0] == 2 * (pf.n_obs - 1) and st["k_gather"][0] == 0 and st["k_propagate"][1] > 0.0
    pf.set_profiling(1)
    pop = pf.alloc_population(1, 4096)
    pf.partial_log_likelihood(pop, theta, 1, 3, seed=2)
    assert pf.kernel_stats()["k_gather"][0] == 1
    pf.set_profiling(2)
    n0 = pf.kernel_stats()["k_propagate"][0]
    pf.set_profiling(False)
    pf.loglik(theta, 4096, seed=9)
    assert pf.kernel_stats()["k_propagate"][0] == n0
    pf.set_profiling(True)
    pf.loglik(theta, 4096, seed=9)
    assert pf.kernel_stats()["k_propagate"][0] == n0 + pf.n_obs
    with pytest.raises(dp.DpompError):
        pf._check(pf._lib.dpomp_get_lane_utilisation(pf._h, pf.n_obs + 1, __import__("ctypes").byref(__import__("ctypes").c_double())))
    pf.close()


@pytest.mark.parametrize("name,x0,fd,theta", [c for c in CASES if c[0] in ("SEIR", "LOTKA", "ROSSMAC", "SIR")])
def test_catalogue_models_at_chip_filling_size_all_schedules_agree(dp, orc, name, x0, fd, theta):
    """2^19 particles (8192 waves: the automatic schedule runs workgroup compaction): compaction, static and the LDS
    queue return the same log-likelihood, populations and event count, for 3- and 4-component states and 2 to 6 event
    types; the first observation (propagation, weights, resampling) is also checked against the oracle at full size."""
    model = dp.generate_model(name, x0, freq_dep=fd)
    y = _synthetic_data(dp, orc, model, theta)
    pf = dp.ParticleFilter(model, y)
    P = 1 << 19
    th = np.array(theta)[:, None]
    out = {}
    for variant in (0, 1, 2):
        pf.set_propagate_variant(variant)
        pop = pf.alloc_population(1, P)
        ll = pf.partial_log_likelihood(pop, th, 1, len(y), seed=5)
        out[variant] = (ll[0], pop.download(0), pf.last_event_count(), int(pf.last_status[0]))
        pop.close()
    for variant in (1, 2):
        assert out[variant][0] == out[0][0] and out[variant][2] == out[0][2] and out[variant][3] == out[0][3]
        assert np.array_equal(out[variant][1], out[0][1])
    assert np.isfinite(out[0][0]) and out[0][3] == 0
    # and the oracle at the same size for the first observation (propagate + weights + resampling)
    pf.set_propagate_variant(0)
    pop = pf.alloc_population(1, P)
    ll1 = pf.partial_log_likelihood(pop, th, 1, 1, seed=5)
    host = np.zeros((P, model.n_states), dtype=np.int64)
    ref = orc.pf_partial(pf.desc, host, theta, 1, 1, seed=5, nthreads=8)
    assert ll1[0] == ref.ll and pf.last_event_count() == ref.events
    assert np.array_equal(pop.download(0), host)
    pf.close()


@pytest.mark.parametrize("tiles_per_wg", [2, 4, 7])
@pytest.mark.parametrize("rs_type,ess,B,P", [(1, 0.0, 3, 6000), (2, 0.0, 2, 9000), (3, 0.0, 2, 5000), (1, 0.5, 3, 6000), (1, 0.0, 2, 1025)])
def test_offspring_boundaries_with_several_tiles_per_workgroup(dp, orc, pooley, monkeypatch, tiles_per_wg, rs_type, ess, B, P):
    """On chip-filling launches a workgroup of the offspring-boundary kernel walks several consecutive tiles of a filter
    (prefix carried from tile to tile, the boundary before a tile taken from the tile before it).  Forced here at sizes
    the oracle finishes quickly, with tile counts that are not a multiple of the group (6 and 9 tiles by 2, 4 and 7),
    all three resamplers and the ESS-adaptive mode."""
    monkeypatch.setenv("DPOMP_TILES_PER_WG", str(tiles_per_wg))
    pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), pooley, rs_type=rs_type, essc=ess, ess_adaptive=ess > 0)
    theta = np.array([[0.003, 0.0025, 0.0035][:B], [0.1, 0.08, 0.12][:B]])
    pop = pf.alloc_population(B, P)
    ll = pf.partial_log_likelihood(pop, theta, 1, len(pooley), seed=5, seg_base=2)
    for b in range(B):
        ref = orc.pf_loglik(pf.desc, P, theta[:, b], seed=5, seg=2 + b, rs_type=rs_type, ess_frac=ess)
        assert ll[b] == ref.ll and pf.last_status[b] == ref.status, (b, ll[b], ref.ll)
        assert np.array_equal(pop.download(b), ref.pop)
    pf.close()


@pytest.mark.parametrize("unfused", [False, True])
def test_chip_filling_launch_takes_four_tiles_per_workgroup_by_default(dp, orc, make_obs, unfused):
    """8192 tiles or more in a launch: the default picks 4 tiles per workgroup -- 9 filters x 2^20 particles (fused tile
    prefix) and 3 filters x (3 x 2^20 + 17) (more than 2048 tiles per filter: prefixes from the scan kernel).  Low
    rates keep the oracle fast; parity stays bit-exact."""
    y = make_obs([1.0, 2.0, 3.0], [[0, 1], [0, 2], [0, 1]])
    pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), y)
    B, P = (3, 3 * (1 << 20) + 17) if unfused else (9, 1 << 20)
    theta = np.stack([np.linspace(0.0028, 0.0033, B), np.linspace(0.18, 0.22, B)])
    pop = pf.alloc_population(B, P)
    ll = pf.partial_log_likelihood(pop, theta, 1, 3, seed=4, seg_base=1)
    for b in (0, B - 1):
        ref = orc.pf_loglik(pf.desc, P, theta[:, b], seed=4, seg=1 + b, nthreads=8)
        assert ll[b] == ref.ll
        assert np.array_equal(pop.download(b), ref.pop)
    pf.close()


@pytest.mark.parametrize("rs_type,ess,P", [(2, 0.0, 5 * (1 << 20)), (3, 0.0, 4 * (1 << 20) + 4097), (1, 0.5, 3 * (1 << 20)),
                                           (2, 0.5, 3 * (1 << 20) + 5)])
def test_many_tiles_other_resamplers_and_ess_sums(dp, orc, pooley, rs_type, ess, P):
    """Filters of more than 4096 tiles (4 Mi particles: the stratified / multinomial gather searches the tile ends in
    L2 instead of LDS) and of more than 2048 tiles in the ESS mode (separate scan kernel, parallel 128-bit sum of the
    squared weights): the first observation -- propagation, weights, ESS decision, resampling -- against the oracle."""
    pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), pooley, rs_type=rs_type, essc=ess, ess_adaptive=ess > 0)
    theta = [0.003, 0.1]
    pop = pf.alloc_population(1, P)
    ll = pf.partial_log_likelihood(pop, theta, 1, 1, seed=77)
    host = np.zeros((P, 2), dtype=np.int64)
    ref = orc.pf_partial(pf.desc, host, theta, 1, 1, seed=77, rs_type=rs_type, nthreads=8, ess_frac=ess)
    assert ll[0] == ref.ll and pf.last_event_count() == ref.events and pf.last_status[0] == ref.status
    assert np.array_equal(pop.download(0), host)
    pf.close()


@pytest.mark.parametrize("rpl", [0, 32, 33])
@pytest.mark.parametrize("B,P", [(7, 200), (33, 100), (5, 64), (3, 511), (9, 300), (4, 1500), (2100, 200)])
def test_many_filters_per_workgroup_match_oracle(dp, orc, pooley, B, P, rpl):
    """Filters with DIFFERENT parameters share workgroups when they are laid end to end (33; the default for thousands
    of small filters, 2100 x 200): every filter must still equal the oracle run of its own parameters -- populations,
    event counts, log-likelihoods -- in one call and one observation per call (the SMC^2 increments)."""
    rng = np.random.default_rng(B * 1000 + P)
    theta = np.stack([rng.uniform(0.0015, 0.0045, B), rng.uniform(0.05, 0.2, B)])
    pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), pooley)
    pf.set_propagate_variant(rpl)
    pop = pf.alloc_population(B, P)
    ll = pf.partial_log_likelihood(pop, theta, 1, pf.n_obs, seed=77, seg_base=5)
    events = pf.last_event_count()
    pop2 = pf.alloc_population(B, P)
    inc = np.zeros(B)
    for k in range(1, pf.n_obs + 1):
        inc += pf.partial_log_likelihood(pop2, theta, k, k, seed=77, seg_base=5)
    want_events = 0
    check = range(B) if B <= 64 else list(range(0, B, 97)) + [B - 1]
    for b in check:
        ref = orc.pf_loglik(pf.desc, P, theta[:, b], seed=77, seg=5 + b)
        want_events += ref.events
        assert ll[b] == ref.ll, (b, ll[b], ref.ll)
        assert abs(inc[b] - ref.ll) < 1e-12
        assert np.array_equal(pop.download(b), ref.pop), b
        assert np.array_equal(pop2.download(b), ref.pop), b
    if B <= 64:
        assert events == want_events
    pf.close()
