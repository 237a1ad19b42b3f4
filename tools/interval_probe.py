"""Per-interval cost of the filter for each propagate variant (static / workgroup compaction).

    MODEL=SIS python tools/interval_probe.py        # on the GPU box
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
    P = 1 << int(os.environ.get("LOG2P", "20"))
    fname, ic, theta = CASES[name]
    y = dp.get_observations(os.path.join(G, fname))
    model = dp.generate_model(name, ic)
    pf = dp.ParticleFilter(model, y)
    th = np.array(theta, dtype=float).reshape(-1, 1)
    for variant in [int(v) for v in os.environ.get("VARIANTS", "1,32").split(",")]:
        pf.set_propagate_variant(variant)
        pop = pf.alloc_population(1, P)
        times = np.zeros(pf.n_obs)
        reps = 4
        for rep in range(reps + 1):
            for k in range(1, pf.n_obs + 1):
                t0 = time.perf_counter()
                pf.partial_log_likelihood(pop, th, k, k, seed=5 + rep)
                if rep:
                    times[k - 1] += time.perf_counter() - t0
        print(json.dumps({"model": name, "variant": variant, "log2_P": int(np.log2(P)),
                          "interval_ms": [round(t / reps * 1e3, 3) for t in times], "total_ms": round(times.sum() / reps * 1e3, 3)}),
              flush=True)
        pop.close()
    pf.close()


if __name__ == "__main__":
    main()
