#!/usr/bin/env python3
"""The whole SMC^2 benchmark configuration (SEIR, 512 theta-particles x 2^16 state particles -- the shape that is
meant for 8 GPUs) on ONE GPU, end to end through run_smc2_analysis: wall time, evidence, posterior means, and the
exact likelihood at the posterior mean for orientation.  JSON on stdout.

    python tools/smc2_full_config.py [n_theta] [log2_particles]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpomp_amd as dp  # noqa: E402

n_theta = int(sys.argv[1]) if len(sys.argv) > 1 else 512
log2p = int(sys.argv[2]) if len(sys.argv) > 2 else 16
y = dp.get_observations(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "seir_synthetic.csv"))
model = dp.generate_model("SEIR", [100, 0, 1, 0])
model.prior = dp.UniformProductPrior([0.0, 0.0, 0.0], [0.02, 0.6, 0.4])
out = {}
for rep in range(2):  # the first run pays the allocations and the clock ramp
    t = time.perf_counter()
    res = dp.run_smc2_analysis(model, y, np_=n_theta, npf=1 << log2p, seed=3 + rep, verbose=False)
    out[f"run{rep}"] = {"seconds": round(time.perf_counter() - t, 3), "minus_ln_evidence": [float(v) for v in np.atleast_1d(res.bme)],
                        "posterior_mean": [float(v) for v in res.mu], "status": int(getattr(res, "status", 0))}
out["shape"] = {"theta_particles": n_theta, "state_particles": 1 << log2p, "observations": len(y),
                "particle_steps_per_increment_pass": n_theta * (1 << log2p)}
print(json.dumps(out, indent=1))
