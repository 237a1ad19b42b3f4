import os, sys, time, json
import numpy as np
sys.path.insert(0, os.getcwd())
import dpomp_amd as dp
y = dp.get_observations("tests/golden/pooley.csv")
pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), y)
pf.set_profiling(True)
for theta in ([0.0, 0.0], [1e-5, 1e-3]):
    th = np.array(theta).reshape(-1, 1)
    for v in (1, 32, 2):
        pf.set_propagate_variant(v)
        for rep in range(3):
            pf.loglik(th, 1 << 20, seed=rep)
        s0 = pf.kernel_stats()
        for rep in range(5):
            pf.loglik(th, 1 << 20, seed=rep)
        s1 = pf.kernel_stats()
        out = {k: round((s1[k][1] - s0[k][1]) / max(1, s1[k][0] - s0[k][0]) * 1e3, 2) for k in s1 if s1[k][0] > s0[k][0]}
        print(json.dumps({"theta": theta, "variant": v, "avg_us": out, "events": pf.last_event_count()}), flush=True)
