"""E[exp(ll_hat)] = p(y | theta): the bootstrap filter's likelihood estimate is unbiased on the likelihood scale.
Runs N independent filters on the GPU and compares log(mean(exp(ll_hat))) with the exact CTMC likelihood of the Pooley
SIS data (oracle/exact_sis.py); with N ~ 10^6 this resolves a relative bias of ~1e-3 in the whole pipeline (streams,
event loop, weights, resampling).

    python tools/unbiasedness_check.py        # on the GPU box
"""
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpomp_amd as dp  # noqa: E402

EXACT = {(0.003, 0.1): -15.6873794556, (0.0032, 0.1): -16.0197707226, (0.0025, 0.08): -15.4530158001,
         (0.004, 0.15): -17.3813549614}   # exact values, re-derived by oracle/exact_sis.py (tests/test_oracle_exact.py)


def main():
    y = dp.get_observations(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "pooley.csv"))
    pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), y)
    for (theta, exact), (P, N, rs, ess) in zip(list(EXACT.items()) * 2,
                                                [(200, 1 << 20, 1, 0.0), (200, 1 << 20, 1, 0.0), (1024, 1 << 17, 1, 0.0),
                                                 (1024, 1 << 17, 1, 0.0), (200, 1 << 19, 2, 0.0), (200, 1 << 19, 3, 0.0),
                                                 (200, 1 << 19, 1, 0.5), (64, 1 << 20, 1, 0.0)]):
        pf.set_resampler(rs)
        pf.set_ess_threshold(ess)
        B = 4096
        lls = []
        for c in range(N // B):
            th = np.tile(np.array(theta)[:, None], (1, B))
            lls.append(pf.loglik(th, P, seed=77, seg_base=c * B))
        ll = np.concatenate(lls)
        m = ll.max()
        w = np.exp(ll - m)
        est = m + math.log(w.mean())
        se = w.std(ddof=1) / math.sqrt(len(w)) / w.mean()     # relative standard error of mean(exp(ll))
        print(json.dumps({"theta": theta, "particles": P, "filters": len(ll), "resampler": rs, "ess_threshold": ess,
                          "log_mean_exp": est, "exact": exact, "difference": est - exact, "standard_error": se,
                          "z": (est - exact) / se, "mean_ll": float(ll.mean())}), flush=True)
    pf.close()


if __name__ == "__main__":
    main()
