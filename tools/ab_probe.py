#!/usr/bin/env python3
"""A/B timing of experimental builds of the library on one GPU box: for every library path given (``-`` = the in-tree
build) a child process loads it through DPOMP_HIP_LIB and prints milliseconds per filter call for Pooley SIS 2^20,
SEIR 4 x 2^18, SEIR 64 x 2^16, SIS 4000 x 200 and Lotka-Volterra 2^20 x 20 observations.

    python tools/ab_probe.py - build_variants/libX.so -
"""
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, ".")


def one():
    import dpomp_amd as dp
    out = []
    cases = [("SIS", [100, 1], "tests/golden/pooley.csv", [0.003, 0.1], 1, 1 << 20),
             ("SEIR", [100, 0, 1, 0], "tests/golden/seir_synthetic.csv", [0.005, 0.2, 0.1], 4, 1 << 18),
             ("SEIR", [100, 0, 1, 0], "tests/golden/seir_synthetic.csv", [0.005, 0.2, 0.1], 64, 1 << 16),
             ("SIS", [100, 1], "tests/golden/pooley.csv", [0.003, 0.1], 4000, 200),
             ("LOTKA", [70, 70], "tests/golden/lotka_synthetic.csv", [0.5, 0.0025, 0.3], 1, 1 << 20)]
    for name, x0, f, th, B, P in cases:
        pf = dp.ParticleFilter(dp.generate_model(name, x0), dp.get_observations(f))
        t = np.tile(np.array(th)[:, None], (1, B))
        n = 100 if name != "LOTKA" else 25
        for i in range(25 if name != "LOTKA" else 6):
            pf.loglik(t, P, seed=i)
        t0 = time.perf_counter()
        for i in range(n):
            ll = pf.loglik(t, P, seed=i)
        out.append((time.perf_counter() - t0) / n * 1e3)
        pf.close()
    print(f"{os.environ.get('DPOMP_HIP_LIB', '-'):40s}", " ".join(f"{v:8.4f}" for v in out), f"{ll[0]:.6f}", flush=True)


if __name__ == "__main__":
    if sys.argv[1:] == ["one"]:
        one()
    else:
        for lib in sys.argv[1:] or ["-"]:
            env = dict(os.environ)
            env.pop("DPOMP_HIP_LIB", None)
            if lib != "-":
                env["DPOMP_HIP_LIB"] = os.path.abspath(lib)
            r = subprocess.run([sys.executable, __file__, "one"], env=env, timeout=170)
            if r.returncode != 0:
                sys.exit(r.returncode)
