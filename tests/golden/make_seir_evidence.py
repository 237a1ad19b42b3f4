#!/usr/bin/env python3
"""Exact evidence and posterior mean of the committed synthetic SEIR data set under a uniform prior on a box
-> tests/golden/seir_evidence.json.

    p(y) = (1/V) * integral over the box of p(y | theta) d theta

with p(y | theta) from the exact CTMC forward algorithm (oracle/exact_ctmc.py: 182 002 reachable states, sparse
generator from the reference's seir_rf / m_transition, src/hmm_examples.jl:117-121, :189; uniformisation) and the
integral by a tensor Gauss-Legendre rule.  The likelihood is smooth in theta, so the rule converges geometrically in
the number of nodes per axis; two rules (n and a smaller one) are stored so that their difference shows the
quadrature error.  No random number is involved: this is the known answer the SMC^2 driver (run_pibis,
src/hmm_ibis.jl:12-135) is tested against on a model other than Pooley SIS.

Slow: every node is one forward recursion over 182 002 states (5-12 s); the whole table is ~2500 of them.
Usage: python tests/golden/make_seir_evidence.py [--n 12] [--n-check 9] [--procs 6] [--box lo1,hi1,lo2,hi2,lo3,hi3]
"""
import argparse
import json
import math
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

X0 = [100, 0, 1, 0]
DEFAULT_BOX = (0.002, 0.010, 0.05, 0.50, 0.03, 0.25)


def _data():
    import dpomp_amd as dp
    y = dp.get_observations(os.path.join(HERE, "seir_synthetic.csv"))
    return [o.time for o in y], [o.val[2] for o in y]


_TIMES, _VALS = None, None


def _ll(theta):
    global _TIMES, _VALS
    from oracle import exact_ctmc as ex
    if _TIMES is None:
        _TIMES, _VALS = _data()
    r = ex.exact_loglik("SEIR", tuple(theta), X0, _TIMES, _VALS)
    return r["ll"], r["truncation"]


def rule(n, box, pool):
    x, w = np.polynomial.legendre.leggauss(n)
    w = w / 2.0  # weights of the uniform density on the interval
    axes = [0.5 * (box[2 * d] + box[2 * d + 1]) + 0.5 * (box[2 * d + 1] - box[2 * d]) * x for d in range(3)]
    nodes = [(axes[0][i], axes[1][j], axes[2][k]) for i in range(n) for j in range(n) for k in range(n)]
    wts = np.array([w[i] * w[j] * w[k] for i in range(n) for j in range(n) for k in range(n)])
    t0 = time.time()
    res = []
    for c, r in enumerate(pool.imap(_ll, nodes, chunksize=4)):
        res.append(r)
        if (c + 1) % 50 == 0:
            print(f"  n={n}: {c + 1}/{len(nodes)} nodes, {time.time() - t0:.0f} s", flush=True)
    ll = np.array([r[0] for r in res])
    m = ll.max()
    lw = wts * np.exp(ll - m)
    Z = lw.sum()
    th = np.array(nodes)
    mean = (lw[:, None] * th).sum(0) / Z
    second = (lw[:, None] * th * th).sum(0) / Z
    return {"n": n, "log_evidence": m + math.log(Z), "neg_log_evidence": -(m + math.log(Z)),
            "posterior_mean": mean.tolist(), "posterior_sd": np.sqrt(np.maximum(second - mean ** 2, 0)).tolist(),
            "max_ll": float(m), "argmax_node": th[int(ll.argmax())].tolist(),
            "max_truncation": float(max(r[1] for r in res)),
            # share of the integral carried by the outermost node layer of each axis: small = the box holds the mass
            "edge_share": [float(lw[(np.isclose(th[:, d], axes[d][0])) | (np.isclose(th[:, d], axes[d][-1]))].sum() / Z) for d in range(3)],
            "axes": [a.tolist() for a in axes], "ll": ll.tolist()}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=12)
    ap.add_argument("--n-check", type=int, default=9)
    ap.add_argument("--procs", type=int, default=6)
    ap.add_argument("--box", type=str, default=",".join(str(v) for v in DEFAULT_BOX))
    ap.add_argument("--out", type=str, default=os.path.join(HERE, "seir_evidence.json"))
    args = ap.parse_args()
    box = tuple(float(v) for v in args.box.split(","))
    out = {"model": "SEIR", "x0": X0, "data": "tests/golden/seir_synthetic.csv (observed component 3, sigma 2)",
           "prior": {"kind": "uniform box", "lower": [box[0], box[2], box[4]], "upper": [box[1], box[3], box[5]]},
           "rules": []}
    with Pool(args.procs) as pool:
        for n in [args.n] + ([args.n_check] if args.n_check > 0 else []):
            r = rule(n, box, pool)
            print(f"n={n}: -ln evidence = {r['neg_log_evidence']:.6f}, posterior mean = {r['posterior_mean']}, "
                  f"sd = {r['posterior_sd']}, edge share = {r['edge_share']}", flush=True)
            out["rules"].append(r)
            with open(args.out, "w") as f:
                json.dump(out, f, indent=1)
    if len(out["rules"]) == 2:
        a, b = out["rules"]
        out["quadrature_error_estimate"] = abs(a["neg_log_evidence"] - b["neg_log_evidence"])
        out["neg_log_evidence"] = a["neg_log_evidence"]
        out["posterior_mean"] = a["posterior_mean"]
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)
