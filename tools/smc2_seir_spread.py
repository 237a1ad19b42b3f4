#!/usr/bin/env python3
"""SMC^2 on the synthetic SEIR data under the uniform prior box whose exact evidence is committed
(tests/golden/seir_evidence.json): -ln evidence and posterior means over seeds, as the reference's driver runs
(no proposal-density correction, src/hmm_ibis.jl:104) and with the correction.
    python tools/smc2_seir_spread.py [n_theta] [log2_particles] [n_seeds]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dpomp_amd as dp  # noqa: E402

n_theta = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
log2p = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n_seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 12
exact = json.load(open(os.path.join(ROOT, "tests", "golden", "seir_evidence.json")))
y = dp.get_observations(os.path.join(ROOT, "tests", "golden", "seir_synthetic.csv"))
model = dp.generate_model("SEIR", [100, 0, 1, 0])
model.prior = dp.UniformProductPrior(exact["prior"]["lower"], exact["prior"]["upper"])
out = {"exact_neg_log_evidence": exact["neg_log_evidence"], "exact_posterior_mean": exact["posterior_mean"],
       "theta_particles": n_theta, "state_particles": 1 << log2p}
for corr in (False, True):
    runs = [dp.run_smc2_analysis(model, y, np_=n_theta, npf=1 << log2p, seed=s, verbose=False, proposal_correction=corr)
            for s in range(1, n_seeds + 1)]
    bme = np.array([r.bme for r in runs])
    mu = np.array([r.mu for r in runs])
    out["corrected" if corr else "as_reference"] = {
        "bme0_mean": float(bme[:, 0].mean()), "bme0_sd": float(bme[:, 0].std(ddof=1)), "bme1_mean": float(bme[:, 1].mean()),
        "bme1_sd": float(bme[:, 1].std(ddof=1)), "mu_mean": mu.mean(0).tolist(), "mu_sd": mu.std(0, ddof=1).tolist(),
        "n_resample": [int(r.n_resample) for r in runs], "acceptance": [round(float(r.acceptance_rate), 3) for r in runs]}
print(json.dumps(out, indent=1))
