#!/usr/bin/env python3
"""Runs the single-GPU shapes of the SEIR and Lotka-Volterra benchmark configurations (synthetic data simulated with
this framework's own simulator, seed 20261003: tests/golden/make_datasets.py) and prints timings.  Not part of the test suite."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import dpomp_amd as dp  # noqa: E402


def synth(model, theta, tmax, num_obs):
    """The committed synthetic data sets (tests/golden/make_datasets.py)."""
    name = "seir_synthetic.csv" if model.n_states == 4 else "lotka_synthetic.csv"
    return dp.get_observations("tests/golden/" + name), -1


def timed(fn, reps=3):
    fn()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t) / reps, out


res = {}
# config 3: SEIR, 2^18 particles, 4 chains (one batched call = one PMMH iteration of all chains)
seir = dp.generate_model("SEIR", [100, 0, 1, 0])
th3 = [0.005, 0.2, 0.1]
y3, nev3 = synth(seir, th3, 100.0, 5)
pf = dp.ParticleFilter(seir, y3)
theta4 = np.tile(np.array(th3)[:, None], (1, 4)) * (1 + 0.05 * np.arange(4))
dt, ll = timed(lambda: pf.loglik(theta4, 1 << 18, seed=1))
res["config3_seir_4x2^18"] = {"obs": [o.val for o in y3], "events_in_truth": nev3, "s_per_batched_call": dt, "ll": ll.tolist(),
                              "particle_steps_per_s": 4 * (1 << 18) * 5 / dt, "events": pf.last_event_count(), "status": pf.last_status.tolist()}
# config 4 shape on one GPU: 64 theta-particles x 2^16 state particles (one rank's share of 512)
rng = np.random.default_rng(1)
th64 = np.stack([rng.uniform(0.003, 0.008, 64), rng.uniform(0.1, 0.3, 64), rng.uniform(0.05, 0.2, 64)])
dt, ll = timed(lambda: pf.loglik(th64, 1 << 16, seed=2))
res["config4_share_seir_64x2^16"] = {"s_per_batched_call": dt, "particle_steps_per_s": 64 * (1 << 16) * 5 / dt,
                                     "ll_range": [float(ll.min()), float(ll.max())], "events": pf.last_event_count()}
pf.close()
# config 5: Lotka-Volterra, 2^20 particles, 20 unit-spaced observations, event cap on
lv = dp.generate_model("LOTKA", [70, 70])
lv.max_events_per_interval = 20000
th5 = [0.5, 0.0025, 0.3]
y5, nev5 = synth(lv, th5, 20.0, 20)
pf = dp.ParticleFilter(lv, y5)
dt, ll = timed(lambda: pf.loglik(th5, 1 << 20, seed=3), reps=2)
res["config5_lotka_2^20"] = {"obs": [o.val for o in y5], "events_in_truth": nev5, "s_per_call": dt, "ll": ll.tolist(),
                             "particle_steps_per_s": (1 << 20) * 20 / dt, "events": pf.last_event_count(), "status": pf.last_status.tolist()}
pf.close()
print(json.dumps(res, indent=1))
