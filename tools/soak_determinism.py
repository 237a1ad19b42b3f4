"""Soak test: the same filter call repeated many times must return the same bits every time (log-likelihood, event
count and a hash of the final population), for every schedule -- a race in the workgroup compaction or in the atomics
would show up as a rare mismatch.

    python tools/soak_determinism.py      # on the GPU box, ~1 minute
"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpomp_amd as dp  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
CASES = {"SIS": ("pooley.csv", [100, 1], [0.003, 0.1]), "SEIR": ("seir_synthetic.csv", [100, 0, 1, 0], [0.005, 0.2, 0.1]),
         "LOTKA": ("lotka_synthetic.csv", [70, 70], [0.5, 0.0025, 0.3])}


def main():
    reps = int(os.environ.get("REPS", "400"))
    for name, (fname, ic, theta) in CASES.items():
        y = dp.get_observations(os.path.join(G, fname))
        pf = dp.ParticleFilter(dp.generate_model(name, ic), y)
        th = np.array(theta, dtype=float).reshape(-1, 1)
        for variant, P in ((0, 1 << 20), (32, 1 << 17), (2, 1 << 17)):
            pf.set_propagate_variant(variant)
            pop = pf.alloc_population(1, P)
            seen, hashes = set(), set()
            n = reps if P < (1 << 20) or name != "LOTKA" else reps // 4
            for r in range(n):
                ll = pf.partial_log_likelihood(pop, th, 1, pf.n_obs, seed=12345)
                seen.add((float(ll[0]), pf.last_event_count()))
                if r % 50 == 0:
                    hashes.add(hashlib.sha1(pop.download(0).tobytes()).hexdigest())
            print(json.dumps({"model": name, "variant": variant, "particles": P, "calls": n, "distinct_results": len(seen),
                              "distinct_population_hashes": len(hashes)}), flush=True)
            assert len(seen) == 1 and len(hashes) == 1
            pop.close()
        pf.close()


if __name__ == "__main__":
    main()
