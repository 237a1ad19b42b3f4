"""One-off fuzz campaign: many random rate-table models (tests/test_random_models.py's generator) through the GPU
filter with random schedules and sizes, each compared bit for bit with the oracle.

    python tools/fuzz_random_models.py --models 400 --jit 0 --budget-seconds 600 [--out result.json]   # on the GPU box

--jit 1 compiles every table at create time (about 0.7 s per distinct model: budget accordingly), --jit 0 runs the
device interpreter.  The run stops cleanly at the budget, at the first mismatch, or when the models are done, and ALWAYS
writes its JSON (how far it got, why it stopped).
"""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dpomp_amd as dp  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (checker)
from test_random_models import random_model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", type=int, default=int(os.environ.get("N", "400")))
    ap.add_argument("--jit", type=int, choices=[0, 1], default=0, help="1 = compile rate tables at create time, 0 = interpret them")
    ap.add_argument("--budget-seconds", type=float, default=600.0)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    n_models = args.models
    bad, inf, capped = [], 0, 0
    t_start, done, stopped = time.time(), 0, "all models done"
    for seed in range(n_models):
        if time.time() - t_start > args.budget_seconds:
            stopped = f"time budget of {args.budget_seconds:.0f} s reached"
            break
        if bad:
            stopped = "first mismatch"
            break
        rng = np.random.default_rng(50000 + seed)
        model, theta = random_model(dp, rng)
        model.max_events_per_interval = 5000
        warnings.simplefilter("ignore", dp.EventCapWarning)
        y = [dp.Observation(t, int(rng.random() < 0.85), 1.0, [int(rng.integers(5, 60))]) for t in (1.0, 2.5, 4.0, 4.0 + float(rng.random()))]
        pf = dp.ParticleFilter(model, y, jit=bool(args.jit))
        variant = int(rng.choice([0, 1, 2, 4, 16, 32, 33]))
        pf.set_propagate_variant(variant)
        P = int(rng.integers(2, 3000))
        rs = int(rng.choice([1, 1, 2, 3]))
        pf.set_resampler(rs)
        pop = pf.alloc_population(2, P)
        th2 = np.stack([theta, theta * 1.3], axis=1)
        ll = pf.partial_log_likelihood(pop, th2, 1, len(y), seed=seed, seg_base=3)
        for b in range(2):
            ref = orc.pf_loglik(pf.desc, P, th2[:, b], seed=seed, seg=3 + b, rs_type=rs)
            same = (np.array_equal(pop.download(b), ref.pop) and pf.last_status[b] == ref.status
                    and ((ll[b] == ref.ll) or (np.isinf(ll[b]) and np.isinf(ref.ll))))
            if not same:
                bad.append({"seed": seed, "b": b, "variant": variant, "P": P, "rs": rs, "gpu": float(ll[b]), "oracle": ref.ll})
            inf += int(np.isinf(ref.ll))
            capped += int(ref.status & 1)
        pf.close()
        done = seed + 1
        if seed % 100 == 99:
            print(f"# {seed + 1} models, {len(bad)} mismatches", file=sys.stderr, flush=True)
    text = json.dumps({"models_requested": n_models, "models": done, "filters": 2 * done, "jit": args.jit, "stopped": stopped,
                       "seconds": round(time.time() - t_start, 1), "mismatches": bad, "minus_inf_results": inf,
                       "event_cap_hits": capped})
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")
    print(text)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
