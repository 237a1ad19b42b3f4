#!/usr/bin/env python3
"""Exact log-likelihoods (oracle/exact_ctmc.py: sparse generator + uniformisation forward algorithm) of the committed
synthetic SEIR and Lotka-Volterra data sets -> tests/golden/exact_ctmc.json.

These are known answers independent of any random number generator, of the C oracle and of the GPU code: they follow
from the reference's model definitions alone (seir_rf / lotka_rf, m_transition, obs seq; src/hmm_examples.jl).
SEIR has 182 002 reachable states and is exact up to the uniformisation truncation (< 1e-12 per interval).
Lotka-Volterra is unbounded: the value is computed on a box 0..bound in both components with the mass leaving the box
absorbed; two bounds are stored to show the value has converged (leak = largest absorbed mass of any interval).
Run time: ~1 min SEIR (5 s per theta), ~3 min Lotka-Volterra.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import dpomp_amd as dp  # noqa: E402
from oracle import exact_ctmc as ex  # noqa: E402

SEIR_THETAS = [(0.005, 0.2, 0.1), (0.004, 0.25, 0.1), (0.006, 0.15, 0.12), (0.0045, 0.2, 0.08)]
LOTKA_THETA = (0.5, 0.0025, 0.3)

if __name__ == "__main__":
    out = {"SEIR": [], "LOTKA": []}
    y = dp.get_observations(os.path.join(HERE, "seir_synthetic.csv"))
    for th in SEIR_THETAS:
        r = ex.exact_loglik("SEIR", th, [100, 0, 1, 0], [o.time for o in y], [o.val[2] for o in y])
        out["SEIR"].append({"theta": list(th), **r})
        print("SEIR", th, r["ll"], flush=True)
    y = dp.get_observations(os.path.join(HERE, "lotka_synthetic.csv"))
    for bound in (480, 560):
        r = ex.exact_loglik("LOTKA", LOTKA_THETA, [70, 70], [o.time for o in y], [o.val[1] for o in y], bound=bound)
        out["LOTKA"].append({"theta": list(LOTKA_THETA), "bound": bound, **r})
        print("LOTKA", bound, r["ll"], r["leak"], flush=True)
    with open(os.path.join(HERE, "exact_ctmc.json"), "w") as f:
        json.dump(out, f, indent=1)
