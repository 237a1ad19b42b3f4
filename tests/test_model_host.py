"""Host logic: the model catalogue, the descriptor flattening and the data reader, checked against
an independent transcription of the reference's rate functions (src/hmm_examples.jl:103-168)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Independent restatement of the reference formulas (1-based Julia indices -> p[i-1]).
REF_RATES = {
    ("SI", False): lambda t, p: [t[0] * p[0] * p[1]],
    ("SIR", False): lambda t, p: [t[0] * p[0] * p[1], t[1] * p[1]],
    ("SIS", False): lambda t, p: [t[0] * p[0] * p[1], t[1] * p[1]],
    ("SEI", False): lambda t, p: [t[0] * p[0] * p[2], t[1] * p[1]],
    ("SEIR", False): lambda t, p: [t[0] * p[0] * p[2], t[1] * p[1], t[2] * p[2]],
    ("SEIS", False): lambda t, p: [t[0] * p[0] * p[2], t[1] * p[1], t[2] * p[2]],
    ("SI", True): lambda t, p: [t[0] * p[0] * p[1] / sum(p)],
    ("SIR", True): lambda t, p: [t[0] * p[0] * p[1] / sum(p), t[1] * p[1]],
    ("SIS", True): lambda t, p: [t[0] * p[0] * p[1] / sum(p), t[1] * p[1]],
    ("SEI", True): lambda t, p: [t[0] * p[0] * p[2] / sum(p), t[1] * p[1]],
    ("SEIR", True): lambda t, p: [t[0] * p[0] * p[2] / sum(p), t[1] * p[1], t[2] * p[2]],
    ("SEIS", True): lambda t, p: [t[0] * p[0] * p[2] / sum(p), t[1] * p[1], t[2] * p[2]],
    ("LOTKA", False): lambda t, p: [t[0] * p[1], t[1] * p[0] * p[1], t[2] * p[0]],
    ("ROSSMAC", False): lambda t, p: [t[0] * (p[2] + p[3]), t[0] * p[2], t[0] * p[3], t[1] * (p[0] * p[3] / (p[0] + p[1])),
                                      t[2] * (p[1] * p[2] / (p[2] + p[3])), t[3] * p[1]],
}
REF_TRANSITIONS = {  # src/hmm_examples.jl:171-204
    "SI": [[-1, 1]], "SIR": [[-1, 1, 0], [0, -1, 1]], "SIS": [[-1, 1], [1, -1]], "SEI": [[-1, 1, 0], [0, -1, 1]],
    "SEIR": [[-1, 1, 0, 0], [0, -1, 1, 0], [0, 0, -1, 1]], "SEIS": [[-1, 1, 0], [0, -1, 1], [1, 0, -1]],
    "LOTKA": [[0, 1], [1, -1], [-1, 0]],
    "ROSSMAC": [[0, 0, 1, 0], [0, 0, -1, 0], [0, 0, 0, -1], [-1, 1, 0, 0], [0, 0, -1, 1], [1, -1, 0, 0]],
}
OBS_SEQ = {"SI": 2, "SIR": 2, "SIS": 2, "SEI": 3, "SEIR": 3, "SEIS": 3, "LOTKA": 2, "ROSSMAC": 2}


@pytest.mark.parametrize("name,fd", list(REF_RATES))
def test_catalogue_rates_on_host_and_in_the_oracle(dp, orc, name, fd):
    S = len(REF_TRANSITIONS[name][0])
    model = dp.generate_model(name, [50] * S, freq_dep=fd)
    assert np.array_equal(model.m_transition, REF_TRANSITIONS[name])
    assert model.obs_model.seq == (OBS_SEQ[name],) and model.obs_model.sigma == 2.0 and model.t0_index == 0
    K = model.n_params
    obs = [dp.Observation(1.0, 1, 1.0, [0] * S)]
    desc = dp.ModelDescriptor(model, obs)
    rng = np.random.default_rng(hash(name) % 1000)
    for _ in range(200):
        theta = rng.random(K)
        pop = [int(v) for v in rng.integers(1, 400, S)]
        want = REF_RATES[(name, fd)](list(theta), pop)
        host = model.rate_function(np.zeros(model.n_events), theta, pop)
        assert np.array_equal(host, want)                       # Python host evaluation
        assert np.array_equal(orc.rates(desc, theta, pop), want)  # the C table interpreter: bit-equal


def test_model_names_follow_the_reference(dp, capsys):
    assert dp.generate_model("LOTKA", [70, 70]).model_name == "PN"
    assert dp.generate_model("ROSSMAC", [1, 1, 1, 1]).model_name == "SIAB"
    assert dp.generate_model("SEIRS", [1, 1, 1]) is None  # documented but not accepted by the reference either
    assert "not recognised" in capsys.readouterr().out
    m = dp.generate_model("SIS", [100, 1])
    assert len(m.prior) == 2 and m.prior.logpdf([0.5, 0.5]) == 0.0 and m.prior.logpdf([1.5, 0.5]) == -np.inf


def test_choose_event_strict_greater_and_fallthrough(orc):
    cum = [0.3, 0.3, 1.0]  # event 2 has zero rate
    assert orc.choose_event(cum, 0.1) == 0
    assert orc.choose_event(cum, 0.3) == 2   # cum[0] > etc is false at equality; zero-rate event never chosen
    assert orc.choose_event(cum, 0.999999) == 2
    assert orc.choose_event([2.0], 0.5) == 0


def test_gaussian_obs_model(dp, orc):
    om = dp.partial_gaussian_obs_model(2.0, seq=2)
    y = dp.Observation(20.0, 1, 1.0, [0, 18])
    assert om(y, [83, 18], None) == pytest.approx(-1.612085713764618, abs=1e-15)  # tmp1 of src/hmm_examples.jl:61 for sigma = 2
    assert om(y, [81, 20], None) == pytest.approx(-1.612085713764618 - 4 / 8)
    model = dp.generate_model("SIS", [100, 1])
    desc = dp.ModelDescriptor(model, [y])
    for I in (0, 5, 18, 77):
        assert orc.obs_loglik(desc, 0, [101 - I, I]) == om(y, [101 - I, I], None)
    om2 = dp.partial_gaussian_obs_model(1.5, seq=(3, 4), y_seq=(1,))
    assert om2(dp.Observation(1.0, 1, 1.0, [10]), [0, 0, 4, 3], None) == pytest.approx(om2.tmp1 - 9 / 4.5)


def test_get_observations_reads_the_reference_format(dp):
    y = dp.get_observations(os.path.join(GOLDEN, "pooley.csv"))
    assert [o.time for o in y] == [20.0, 40.0, 60.0, 80.0, 100.0]
    assert [o.val for o in y] == [[0, 18], [0, 65], [0, 70], [0, 66], [0, 67]]
    assert all(o.obs_id == 1 and o.prop == 1.0 for o in y)
    arr = np.array([[3.0, 2, 7, 9], [1.0, 0, 4, 5]])
    y2 = dp.get_observations(arr, type_col=2, val_seq=[3, 4])
    assert [o.time for o in y2] == [1.0, 3.0] and y2[0].obs_id == 0 and y2[1].val == [7, 9]


def test_descriptor_flattening(dp, pooley):
    m = dp.generate_model("ROSSMAC", [100, 10, 200, 20])
    y = [dp.Observation(5.0, 1, 1.0, [0, 9, 0, 0])]
    d = dp.ModelDescriptor(m, y).c
    assert (d.n_states, d.n_events, d.n_params, d.n_obs, d.n_obs_values) == (4, 6, 4, 1, 4)
    assert d.rate[0].n_factors == 1 and d.rate[0].factor_mask[0] == 0b1100 and d.rate[0].form == 0
    assert d.rate[3].form == 1 and d.rate[3].factor_mask[0] == 0b0001 and d.rate[3].factor_mask[1] == 0b1000 and d.rate[3].denom_mask == 0b0011
    assert [d.transition[3][s] for s in range(4)] == [-1, 1, 0, 0]
    assert d.obs.state_mask == 0b10 and d.obs.value_mask == 0b10 and d.obs.sigma == 2.0
    with pytest.raises(ValueError):
        dp.generate_model("SIS", [100, 1, 0])
    with pytest.raises(ValueError):
        dp.ModelDescriptor(dp.generate_model("SIS", [100, 1]), [])
    with pytest.raises(IndexError):
        dp.ModelDescriptor(dp.generate_model("SEIR", [100, 0, 1, 0]), pooley)  # y.val[3] does not exist in pooley.csv


def test_dmy_obs_fn(dp):
    y = dp.Observation(1.0, 1, 1.0, [0, 0])
    dp.dmy_obs_fn(y, [4, 5], None)
    assert y.val == [4, 5]
