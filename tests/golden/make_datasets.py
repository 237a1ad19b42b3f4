#!/usr/bin/env python3
"""Synthetic data sets for the SEIR and Lotka-Volterra benchmark configurations (the reference ships data only for
Pooley SIS).

Recipe: simulate with the framework's own simulator (here: the CPU oracle's, which the
GPU simulator equals event for event), observation = state (the reference's default dmy_obs_fn), written in the
reference's CSV dialect (header, ", " separators).  Seed policy: first seed >= 20261003 whose trajectory is a real
outbreak (SEIR: final R >= 50; an extinct epidemic makes a degenerate benchmark).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import dpomp_amd as dp  # noqa: E402
from oracle import oracle as orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def simulate(model, theta, tmax, num_obs, accept):
    blank = dp.generate_observations(tmax, num_obs, model.n_states)
    desc = dp.ModelDescriptor(model, blank)
    seed = 20261003
    while True:
        n, times, types, states = orc.simulate_events(desc, theta, seed=seed)
        if accept(states):
            return seed, n, [o.time for o in blank], states
        seed += 1


def write(path, times, states):
    S = states.shape[1]
    with open(path, "w") as f:
        f.write("time, " + ", ".join(f"val{i + 1}" for i in range(S)) + "\n")
        for t, row in zip(times, states):
            f.write(f"{t}, " + ", ".join(str(int(v)) for v in row) + "\n")


if __name__ == "__main__":
    seir = dp.generate_model("SEIR", [100, 0, 1, 0])
    seed, n, times, states = simulate(seir, [0.005, 0.2, 0.1], 100.0, 5, lambda s: s[-1, 3] >= 50)
    write(os.path.join(HERE, "seir_synthetic.csv"), times, states)
    print("SEIR theta=(0.005,0.2,0.1) seed", seed, "events", n, states.tolist())
    lv = dp.generate_model("LOTKA", [70, 70])
    seed, n, times, states = simulate(lv, [0.5, 0.0025, 0.3], 20.0, 20, lambda s: s[:, 0].min() > 0 and s[:, 1].min() > 0)
    write(os.path.join(HERE, "lotka_synthetic.csv"), times, states)
    print("LOTKA theta=(0.5,0.0025,0.3) seed", seed, "events", n, states[:5].tolist(), "...")
