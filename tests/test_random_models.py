"""Randomised models through the rate table: host evaluation == oracle interpreter (CPU), and the device's generic
evaluator == oracle on whole filters (GPU).  Models are physically sensible by construction: an event only removes
an individual from a compartment that appears in its rate (so rates vanish before a count can go negative), and
ratio/frequency-dependent denominators use the whole population of closed models (never zero)."""
import numpy as np
import pytest


def random_model(dp, rng):
    S = int(rng.integers(2, 9))
    E = int(rng.integers(1, 9))
    K = int(rng.integers(1, 9))
    closed = bool(rng.random() < 0.6)
    terms, trans = [], np.zeros((E, S), dtype=int)
    for e in range(E):
        src = int(rng.integers(1, S + 1))
        other = int(rng.integers(1, S + 1))
        kind = rng.integers(0, 4) if closed else rng.integers(0, 2)
        theta = int(rng.integers(1, K + 1)) if e >= K else e + 1       # every parameter up to min(E,K) is used
        if kind == 0:                                                   # theta * x_src
            terms.append(dp.RateTerm(theta, ((src,),)))
        elif kind == 1:                                                 # theta * x_src * (x_other + ...)
            grp = tuple(sorted({other, int(rng.integers(1, S + 1))}))
            terms.append(dp.RateTerm(theta, ((src,), grp)))
        elif kind == 2:                                                 # frequency dependent
            terms.append(dp.RateTerm(theta, ((src,), (other,)), tuple(range(1, S + 1))))
        else:                                                           # ratio form
            terms.append(dp.RateTerm(theta, ((src,), (other,)), tuple(range(1, S + 1)), "ratio"))
        trans[e, src - 1] -= 1
        if closed or rng.random() < 0.7:
            dst = int(rng.integers(1, S + 1))
            trans[e, dst - 1] += 1                                      # src == dst gives a null move: allowed
    K_used = max(t.theta for t in terms)
    x0 = [int(v) for v in rng.integers(5, 40, S)]
    seq = tuple(sorted({int(rng.integers(1, S + 1)), int(rng.integers(1, S + 1))}))
    model = dp.generate_custom_model("RND", dp.RateFunction(terms), x0, trans,
                                     obs_model=dp.partial_gaussian_obs_model(float(rng.uniform(1.5, 4.0)), seq=seq, y_seq=(1,)),
                                     prior=dp.generate_weak_prior(K_used))
    theta = rng.uniform(0.002, 0.05, K_used)
    return model, theta


@pytest.mark.parametrize("seed", range(25))
def test_host_table_equals_oracle_interpreter(dp, orc, seed):
    rng = np.random.default_rng(1000 + seed)
    model, theta = random_model(dp, rng)
    desc = dp.ModelDescriptor(model, [dp.Observation(1.0, 1, 1.0, [10])])
    for _ in range(40):
        x = [int(v) for v in rng.integers(0, 60, model.n_states)]
        if sum(x) == 0:
            continue
        host = model.rate_function(np.zeros(model.n_events), theta, x)
        ref = orc.rates(desc, theta, x)
        assert np.array_equal(host, ref, equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(16))
def test_random_models_filter_matches_oracle(dp, orc, seed):
    rng = np.random.default_rng(2000 + seed)
    model, theta = random_model(dp, rng)
    model.max_events_per_interval = 5000
    y = [dp.Observation(t, 1, 1.0, [int(rng.integers(5, 60))]) for t in (1.0, 2.5, 4.0)]
    pf = dp.ParticleFilter(model, y)
    pf.set_propagate_variant(int(rng.choice([0, 1, 2, 4, 16, 32, 33])))
    P = int(rng.integers(100, 1400)) if seed % 2 else int(rng.integers(2, 257))
    pop = pf.alloc_population(2, P)
    th2 = np.stack([theta, theta * 1.3], axis=1)
    ll = pf.partial_log_likelihood(pop, th2, 1, 3, seed=seed, seg_base=5)
    for b in range(2):
        ref = orc.pf_loglik(pf.desc, P, th2[:, b], seed=seed, seg=5 + b)
        assert np.array_equal(pop.download(b), ref.pop)
        assert (ll[b] == ref.ll) or (np.isinf(ll[b]) and np.isinf(ref.ll))
        assert pf.last_status[b] == ref.status
    pf.close()
