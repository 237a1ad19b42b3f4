"""Forward simulator (gillespie_sim, src/hmm_sim.jl:86-102) and the simulation / observation file formats
(src/hmm_utils.jl:19-72).  The reference pins `x.population[end][1] == 45` at Julia seed 1
(test/runtests.jl:22-26), which depends on Julia's RNG stream; here the GPU trajectory must equal the
oracle's event for event."""
import os

import numpy as np
import pytest


def test_generate_observations(dp):
    y = dp.generate_observations(100.0, 5, 2)
    assert [o.time for o in y] == [20.0, 40.0, 60.0, 80.0, 100.0]
    assert all(o.obs_id == 1 and o.prop == 1.0 and o.val == [0, 0] for o in y)


def test_save_to_file_round_trip(dp, tmp_path):
    ev = [dp.Event(0.5, 1), dp.Event(1.25, 2), dp.Event(3.0, 1)]
    pop = [[99, 2], [100, 1], [99, 2]]
    obs = [dp.Observation(2.0, 1, 1.0, [100, 1]), dp.Observation(4.0, 0, 1.0, [99, 2])]
    res = dp.SimResults("SIS", dp.Particle(np.array([0.003, 0.1]), [100, 1], [99, 2], ev, 0.0), pop, obs)
    prefix = str(tmp_path / "out") + os.sep
    dp.save_to_file(res, prefix)
    sim = open(prefix + "sim.csv").read().splitlines()
    assert sim[0] == "time, event,1,2" and sim[1] == "0.5,1,99,2" and len(sim) == 4
    back = dp.get_observations(prefix + "obs.csv", type_col=2, val_seq=[3, 4])
    assert [(o.time, o.obs_id, o.val) for o in back] == [(2.0, 1, [100, 1]), (4.0, 0, [99, 2])]


def test_oracle_event_record_is_consistent(dp, orc):
    model = dp.generate_model("SIS", [100, 1])
    desc = dp.ModelDescriptor(model, dp.generate_observations(100.0, 5, 2))
    n, times, types, states = orc.simulate_events(desc, [0.003, 0.1], seed=8)
    assert n == len(times) and np.all(np.diff(times) > 0) and set(np.unique(types)) <= {1, 2}
    final = np.array([100, 1]) + np.asarray(model.m_transition)[types - 1].sum(axis=0)
    assert np.array_equal(final, states[-1])
    # and it is the same trajectory the state-only simulator walks
    assert np.array_equal(orc.simulate(desc, 1, [0.003, 0.1], seed=8)[:, :, 0], states)


@pytest.mark.gpu
@pytest.mark.parametrize("name,x0,theta", [("SIS", [100, 1], [0.003, 0.1]), ("SEIR", [100, 0, 1, 0], [0.005, 0.2, 0.1]),
                                           ("LOTKA", [70, 70], [0.5, 0.0025, 0.3])])
def test_gillespie_sim_matches_oracle_event_for_event(dp, orc, name, x0, theta):
    model = dp.generate_model(name, x0)
    tmax = 20.0 if name == "LOTKA" else 100.0
    res = dp.gillespie_sim(model, theta, tmax=tmax, num_obs=5, n_sims=3, seed=12, verbose=False)
    desc = dp.ModelDescriptor(model, dp.generate_observations(tmax, 5, len(x0)))
    for i, r in enumerate(res):
        n, times, types, states = orc.simulate_events(desc, theta, seed=12, p=i)
        assert len(r.particle.trajectory) == n
        assert [e.time for e in r.particle.trajectory] == list(times)
        assert [e.event_type for e in r.particle.trajectory] == list(types)
        assert r.particle.final_condition == list(states[-1])
        assert [o.val for o in r.observations] == [list(s) for s in states]   # dmy_obs_fn: y.val .= population
        if n:
            assert r.population[-1] == r.particle.final_condition
    single = dp.gillespie_sim(model, theta, tmax=tmax, seed=12, verbose=False)
    assert [e.time for e in single.particle.trajectory] == [e.time for e in res[0].particle.trajectory]


@pytest.mark.gpu
def test_simulation_truncates_at_max_events(dp, orc):
    model = dp.generate_model("SIS", [100, 1])
    pf = dp.ParticleFilter(model, dp.generate_observations(100.0, 5, 2))
    counts, times, types, states = pf.simulate_events([0.003, 0.1], n_sims=64, seed=3, max_events=300)
    ref = [orc.simulate_events(pf.desc, [0.003, 0.1], seed=3, p=i, max_events=300) for i in range(64)]
    assert list(counts) == [r[0] for r in ref] and (counts == -1).any() and (counts >= 0).any()
    for i in range(64):
        assert np.array_equal(times[i], ref[i][1]) and np.array_equal(types[i], ref[i][2])
    pf.close()


@pytest.mark.gpu
def test_simulator_samples_the_exact_transient_law(dp):
    """30 000 forward simulations (the recording kernel behind gillespie_sim): the states at both observation times
    against exp(Q t) applied to the initial condition (chi-square over every state with an expected count >= 25) and
    the number of recorded events against the event count the states imply (SIS: infections - recoveries = I - 1)."""
    from oracle import exact_ctmc
    from test_exact_ctmc import chi_square_against_exact
    theta, x0, n = [0.003, 0.1], [100, 1], 30000
    pf = dp.ParticleFilter(dp.generate_model("SIS", x0), dp.generate_observations(30.0, 2, 2))
    counts, times, types, states = pf.simulate_events(theta, n_sims=n, seed=5, max_events=2000)
    assert (counts >= 0).all()
    for k, t in enumerate((15.0, 30.0)):
        st, pr, leak = exact_ctmc.transient_distribution("SIS", theta, x0, t)
        keys, cnt = np.unique(states[k].T, axis=0, return_counts=True)
        stat, df = chi_square_against_exact({tuple(a): c for a, c in zip(keys, cnt)}, st, pr, n)
        assert df > 20 and abs(stat - df) < 5.0 * np.sqrt(2.0 * df), (t, stat, df)
    for i in range(0, n, 997):
        net = int((types[i] == 1).sum()) - int((types[i] == 2).sum())
        assert net == states[1, 1, i] - 1 and np.all(np.diff(times[i]) >= 0) and (len(times[i]) == 0 or times[i][-1] <= 30.0)
    pf.close()

