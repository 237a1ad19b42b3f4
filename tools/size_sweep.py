"""Throughput of the filter against the number of particles (how much of k_propagate's time is ramp-up / tail).

    python tools/size_sweep.py            # on the GPU box
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpomp_amd as dp  # noqa: E402


G = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
CASES = {   # data set, initial condition, theta (tests/golden/make_datasets.py)
    "SIS": ("pooley.csv", [100, 1], [0.003, 0.1]),
    "SEIR": ("seir_synthetic.csv", [100, 0, 1, 0], [0.005, 0.2, 0.1]),
    "LOTKA": ("lotka_synthetic.csv", [70, 70], [0.5, 0.0025, 0.3]),
}


def main():
    name = os.environ.get("MODEL", "SIS")
    fname, ic, th = CASES[name]
    y = dp.get_observations(os.path.join(G, fname))
    model = dp.generate_model(name, ic)
    pf = dp.ParticleFilter(model, y)
    rows = []
    variants = [int(v) for v in os.environ.get("VARIANTS", "0").split(",")]
    theta = np.array(th, dtype=float).reshape(-1, 1)
    T = len(y)
    for lg in [int(v) for v in os.environ.get("LOG2P", "17,18,19,20,21,22").split(",")]:
        P = 1 << lg
        samples = {v: [] for v in variants}
        ev = 0
        for rep in range(7):            # variants interleaved, so clock drift hits them alike; median reported
            for variant in variants:
                pf.set_propagate_variant(variant)
                pf.loglik(theta, P, seed=1)
                t0 = time.perf_counter()
                n = 3
                for s in range(n):
                    pf.loglik(theta, P, seed=2 + s)
                samples[variant].append((time.perf_counter() - t0) / n)
                ev = pf.last_event_count()
        for variant in variants:
            dt = float(np.median(samples[variant]))
            rows.append({"variant": variant, "log2_P": lg, "ms": round(dt * 1e3, 4), "model": name, "particle_steps_per_s": P * T / dt,
                         "events_per_s": ev / dt, "spread_ms": round((max(samples[variant]) - min(samples[variant])) * 1e3, 4)})
            print(json.dumps(rows[-1]), flush=True)
        print(json.dumps({"log2_P": lg, "lane_utilisation": [round(float(u), 3) for u in pf.lane_utilisation()]}), flush=True)
    pf.close()


if __name__ == "__main__":
    main()
