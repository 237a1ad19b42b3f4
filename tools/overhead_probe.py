#!/usr/bin/env python3
"""Fixed cost of a filter call: a model in which nothing happens (all rates zero), so a call is only its launches,
copies and synchronisation.  Prints microseconds per call for the whole-filter kernel (P <= 256) and for the
multi-kernel path, and for a real 1 x 200 filter (the reference's default size)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import dpomp_amd as dp  # noqa: E402

y = dp.get_observations("tests/golden/pooley.csv")
pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), y)


def per_call(theta, B, P, n=300):
    th = np.tile(np.array(theta)[:, None], (1, B))
    for i in range(20):
        pf.loglik(th, P, seed=i)
    t = time.perf_counter()
    for i in range(n):
        pf.loglik(th, P, seed=i)
    return (time.perf_counter() - t) / n * 1e6


for label, theta, B, P in [("no events, whole-filter kernel 1 x 200", [0.0, 0.0], 1, 200),
                           ("no events, multi-kernel path   1 x 1024", [0.0, 0.0], 1, 1024),
                           ("no events, multi-kernel path   1 x 2^20", [0.0, 0.0], 1, 1 << 20),
                           ("real filter, whole-filter kernel 1 x 200", [0.003, 0.1], 1, 200),
                           ("real filter, 64 x 200", [0.003, 0.1], 64, 200),
                           ("real filter, multi-kernel 1 x 4096", [0.003, 0.1], 1, 4096)]:
    print(f"{label:45s} {per_call(theta, B, P):9.1f} us/call")
