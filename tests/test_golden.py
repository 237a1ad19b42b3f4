"""Committed golden vectors (tests/golden/pf_golden.json, made by tests/golden/make_golden.py):
the oracle must reproduce them on CPU, and the HIP path must reproduce them on the GPU."""
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load():
    return json.load(open(os.path.join(GOLDEN, "pf_golden.json")))


def _model(dp, name):
    if name.startswith("pooley_sis"):
        return dp.generate_model("SIS", [100, 1])
    if name.startswith("seir"):
        return dp.generate_model("SEIR", [100, 0, 1, 0])
    return dp.generate_model("LOTKA", [70, 70])


def _obs(dp, c):
    return [dp.Observation(t, i, 1.0, v) for t, i, v in c["obs"]]


CASES = [c["name"] for c in _load()["cases"]]


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(dp, orc, name):
    c = next(x for x in _load()["cases"] if x["name"] == name)
    desc = dp.ModelDescriptor(_model(dp, name), _obs(dp, c))
    r = orc.pf_loglik(desc, c["P"], c["theta"], seed=c["seed"], trace=True)
    assert r.ll == float.fromhex(c["ll"])
    assert [float(v) for v in r.per_obs_ll] == [float.fromhex(v) for v in c["per_obs_ll"]]
    assert r.events == c["events"] and r.status == c["status"]
    assert r.states.tolist() == c["states_after_propagation"]
    assert r.pop[:64].tolist() == c["final_population_head"]
    assert int((r.pop * (np.arange(c["P"])[:, None] % 251 + 1)).sum()) == c["final_population_checksum"]
    assert r.offspring_end[:64].tolist() == c["last_offspring_end_head"]
    assert r.ancestors[:64].tolist() == c["last_ancestors_head"]


def test_exact_values_in_golden_match_the_forward_algorithm():
    from oracle import exact_sis
    g = _load()["exact_pooley_sis"]
    for key, v in g.items():
        theta = tuple(float(x) for x in key.split(","))
        ll, incs = exact_sis.exact_sis_loglik(theta, [100, 1], exact_sis.POOLEY_TIMES, exact_sis.POOLEY_CASES)
        assert abs(ll - v["ll"]) < 1e-10 and np.allclose(incs, v["increments"], atol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_reproduces_golden(dp, name):
    c = next(x for x in _load()["cases"] if x["name"] == name)
    pf = dp.ParticleFilter(_model(dp, name), _obs(dp, c))
    P, T = c["P"], len(c["obs"])
    pop = pf.alloc_population(1, P)
    total = 0.0
    ref_pop = None
    for k in range(1, T + 1):
        # propagate-only view of step k: run the increment, compare its log mean weight
        ll = pf.partial_log_likelihood(pop, c["theta"], k, k, seed=c["seed"])[0]
        assert ll == float.fromhex(c["per_obs_ll"][k - 1])
        total += ll
    final = pop.download(0)
    assert final[:64].tolist() == c["final_population_head"]
    assert int((final * (np.arange(P)[:, None] % 251 + 1)).sum()) == c["final_population_checksum"]
    full = pf.loglik(c["theta"], P, seed=c["seed"])[0]
    assert full == float.fromhex(c["ll"])
    assert pf.last_event_count() == c["events"]
    pf.close()
