#!/usr/bin/env python3
"""Where many 200-particle filters should leave the whole-filter kernel (one workgroup per filter, everything on
chip) for the multi-kernel path with filters laid end to end over the compaction kernel: ms per call for both.
    python tools/small_filter_crossover.py [P]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import dpomp_amd as dp  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 200
y = dp.get_observations("tests/golden/pooley.csv")
for B in (64, 250, 500, 750, 1000, 1500, 2000, 3000, 4000):
    theta = np.tile(np.array([[0.003], [0.1]]), (1, B))
    row = []
    for variant in (0, 33):
        pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), y)
        pf.set_propagate_variant(variant)
        for i in range(10):
            pf.loglik(theta, P, seed=i)
        t0 = time.perf_counter()
        for i in range(40):
            pf.loglik(theta, P, seed=20 + i)
        row.append((time.perf_counter() - t0) / 40 * 1e3)
        pf.close()
    print(f"B={B:5d} P={P}: default {row[0]:.3f} ms   end-to-end layout forced {row[1]:.3f} ms   waves {B * ((P + 63) // 64)}", flush=True)
