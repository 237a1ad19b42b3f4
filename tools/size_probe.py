#!/usr/bin/env python3
"""Milliseconds per 2^20 particles of a Pooley SIS filter at sizes from 2^17 to 2^22 particles (per-size steady state).
DPOMP_HEAD_START=0 switches the head start of the last workgroups off (compaction kernel) for comparison.

    python tools/size_probe.py ; DPOMP_HEAD_START=0 python tools/size_probe.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import dpomp_amd as dp  # noqa: E402

pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), dp.get_observations("tests/golden/pooley.csv"))
th = np.array([[0.003], [0.1]])
out = []
sizes = (1 << 17, 1 << 18, 1 << 19, 3 << 18, 1 << 20, 3 << 19, 1 << 21, 1 << 22)
for P in sizes:
    n = max(10, int(60 * (1 << 20) / P))
    for i in range(max(8, n // 3)):
        pf.loglik(th, P, seed=i)
    t = time.perf_counter()
    for i in range(n):
        pf.loglik(th, P, seed=i)
    out.append((time.perf_counter() - t) / n * 1e3 / (P / (1 << 20)))
print("head start", os.environ.get("DPOMP_HEAD_START", "1"), " ".join(f"{P / (1 << 20):g}:{v:.4f}" for P, v in zip(sizes, out)), flush=True)
