#!/usr/bin/env python3
"""Generates tests/golden/pf_golden.json from the CPU oracle (SPEC mode) and the exact CTMC
forward algorithm.  The reference itself (Julia) cannot run in the build container, so these
vectors pin the *spec*: any later change to the oracle or to the kernels that moves a bit shows
up against them.  Re-run after a deliberate spec change:  python tests/golden/make_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import dpomp_amd as dp  # noqa: E402
from oracle import exact_sis, oracle as orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def case(name, model, obs, theta, P, seed, nkeep=64):
    desc = dp.ModelDescriptor(model, obs)
    r = orc.pf_loglik(desc, P, theta, seed=seed, trace=True, nkeep=nkeep)
    return {
        "name": name, "theta": list(theta), "P": P, "seed": seed,
        "obs": [[o.time, o.obs_id, o.val] for o in obs],
        "ll": r.ll.hex(), "per_obs_ll": [float(v).hex() for v in r.per_obs_ll], "events": int(r.events),
        "status": int(r.status),
        "states_after_propagation": r.states.tolist(),        # [T][nkeep][S], before resampling
        "final_population_head": r.pop[:nkeep].tolist(),
        "final_population_checksum": int((r.pop * (np.arange(P)[:, None] % 251 + 1)).sum()),
        "last_offspring_end_head": r.offspring_end[:nkeep].tolist(),
        "last_ancestors_head": r.ancestors[:nkeep].tolist(),
    }


def simulated_obs(model, theta, times, seed):
    blank = [dp.Observation(t, 1, 1.0, [0] * model.n_states) for t in times]
    traj = orc.simulate(dp.ModelDescriptor(model, blank), 1, theta, seed=seed)
    return [dp.Observation(t, 1, 1.0, [int(v) for v in traj[i, :, 0]]) for i, t in enumerate(times)]


def main():
    pooley = dp.get_observations(os.path.join(HERE, "pooley.csv"))
    out = {"cases": [], "exact_pooley_sis": {}}
    sis = dp.generate_model("SIS", [100, 1])
    out["cases"].append(case("pooley_sis_p1024", sis, pooley, (0.003, 0.1), 1024, 1))
    out["cases"].append(case("pooley_sis_p1000_ragged", sis, pooley, (0.0032, 0.09), 1000, 20261003))
    seir = dp.generate_model("SEIR", [100, 0, 1, 0])
    th = (0.005, 0.2, 0.1)
    y = simulated_obs(seir, th, [20.0, 40.0, 60.0, 80.0, 100.0], 20261003)  # the SEIR benchmark data recipe (make_datasets.py)
    out["cases"].append(case("seir_synthetic_p1024", seir, y, th, 1024, 3))
    lv = dp.generate_model("LOTKA", [70, 70])
    th = (0.5, 0.0025, 0.3)
    y = simulated_obs(lv, th, [float(t) for t in range(1, 9)], 20261003)
    out["cases"].append(case("lotka_synthetic_p512", lv, y, th, 512, 4))
    for theta in [(0.003, 0.1), (0.0032, 0.1), (0.0025, 0.08), (0.004, 0.15)]:
        ll, incs = exact_sis.exact_sis_loglik(theta, [100, 1], exact_sis.POOLEY_TIMES, exact_sis.POOLEY_CASES)
        out["exact_pooley_sis"][f"{theta[0]},{theta[1]}"] = {"ll": ll, "increments": incs}
    # exact evidence under the prior of test/runtests.jl:29 (100 x 100 quadrature, ~5 minutes): kept from the
    # previous file unless --evidence is given
    key = "exact_pooley_sis_evidence_u001_u05"
    path = os.path.join(HERE, "pf_golden.json")
    if "--evidence" in sys.argv or not os.path.exists(path) or key not in json.load(open(path)):
        out[key] = exact_sis.exact_evidence((0, 0), (0.01, 0.5), n=100)
    else:
        out[key] = json.load(open(path))[key]
    with open(os.path.join(HERE, "pf_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote pf_golden.json:", [c["name"] for c in out["cases"]])


if __name__ == "__main__":
    main()
