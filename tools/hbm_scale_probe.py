#!/usr/bin/env python3
"""The memory-bound stage (weight sum, offspring boundaries, resampling gather) at working sets that do and do not fit
the 256 MiB Infinity Cache: per-kernel durations (HIP events carrying each kernel's own begin and end) against the
bytes they move.  Two byte counts per kernel: the Int64 accounting (state as the reference holds it) and the bytes the
kernels really move (int32 state in HBM).  Usage: python tools/hbm_scale_probe.py [reps] [config ...]  -> JSON."""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
import dpomp_amd as dp  # noqa: E402

POOLEY = [(20.0, [0, 18]), (40.0, [0, 65]), (60.0, [0, 70]), (80.0, [0, 66]), (100.0, [0, 67])]
HBM_PEAK, HBM_MEASURED_COPY = 8000.0, 6290.0   # GB/s: spec, and the measured float4-copy rate of this chip


def moved_bytes(P, S, B):
    """HBM bytes each kernel has to move per launch with int32 state (what is in memory):
    k_wsum reads logw and writes the fixed-point weights; k_offsets reads those and writes N_j (int32); k_gather reads
    N_j, gathers and writes the state."""
    n = P * B
    return {"k_wsum": (8 + 8) * n, "k_offsets": (8 + 4) * n, "k_gather": (4 + 4 * S + 4 * S) * n,
            "k_propagate": (4 * S + 4 * S + 8) * n}


def int64_bytes(P, S, B):
    n = P * B
    return {"k_wsum": 8 * n, "k_offsets": 16 * n, "k_gather": (2 * S + 1) * 8 * n, "k_propagate": (2 * S + 1) * 8 * n}


def run(name, model, y, theta, P, reps, incremental=False):
    """incremental: the filter as one call per observation on a device-resident population -- the way SMC^2 drives it
    (src/hmm_ibis.jl:53-56); every call then ends on a resampling observation and runs the separate gather pass
    (k_gather), which a whole-filter call folds into the next propagation."""
    pf = dp.ParticleFilter(model, y)
    B, S = theta.shape[1], model.n_states
    pop = pf.alloc_population(B, P) if incremental else None

    def one(seed):
        if incremental:
            for k in range(1, len(y) + 1):
                pf.partial_log_likelihood(pop, theta, k, k, seed=seed)
        else:
            pf.loglik(theta, P, seed=seed)
    for i in range(2):
        one(i)
    pf.set_profiling(1)
    for i in range(reps):
        one(10 + i)
    stats = pf.kernel_stats()
    pf.set_profiling(0)
    pf.close()
    mv, i64 = moved_bytes(P, S, B), int64_bytes(P, S, B)
    working_set = B * ((P + 1023) // 1024 * 1024) * (2 * 4 * S + 8 + 8 + 4)
    out = {"incremental_calls": incremental, "filters": B, "particles_per_filter": P, "state_components": S, "working_set_MiB": round(working_set / 2 ** 20, 1),
           "fits_infinity_cache": working_set < 256 * 2 ** 20, "kernels": {}}
    for k, (n, ms) in stats.items():
        if not n or k not in mv:
            continue
        us = ms / n * 1e3
        out["kernels"][k] = {"avg_us": round(us, 2), "launches": n, "moved_bytes": mv[k], "moved_GBs": round(mv[k] / us / 1e3, 1),
                             "frac_of_8TBs": round(mv[k] / us / 1e3 / HBM_PEAK, 3),
                             "frac_of_measured_copy_6.29TBs": round(mv[k] / us / 1e3 / HBM_MEASURED_COPY, 3),
                             "int64_accounting_bytes": i64[k], "int64_accounting_GBs": round(i64[k] / us / 1e3, 1)}
    return out


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    which = sys.argv[2:]
    sis = dp.generate_model("SIS", [100, 1])
    seir = dp.generate_model("SEIR", [100, 0, 1, 0])
    y_sis = [dp.Observation(t, 1, 1.0, v) for t, v in POOLEY]
    y_seir = dp.get_observations("tests/golden/seir_synthetic.csv")
    rng = np.random.default_rng(1)
    th64 = np.stack([rng.uniform(0.003, 0.008, 64), rng.uniform(0.1, 0.3, 64), rng.uniform(0.05, 0.2, 64)])
    configs = {
        "sis_1x2^20": (sis, y_sis, np.tile(np.array([[0.003], [0.1]]), (1, 1)), 1 << 20),
        "seir_64x2^16_config4_share": (seir, y_seir, th64, 1 << 16),
        "sis_16x2^20": (sis, y_sis, np.tile(np.array([[0.003], [0.1]]), (1, 16)), 1 << 20),
        "sis_1x2^24": (sis, y_sis, np.tile(np.array([[0.003], [0.1]]), (1, 1)), 1 << 24),
        "seir_64x2^18": (seir, y_seir, th64, 1 << 18),
    }
    res = {}
    for k, v in configs.items():
        if not which or k in which:
            res[k] = run(k, *v, reps)
        if k + "+incremental" in which or (not which and k in ("seir_64x2^16_config4_share", "seir_64x2^18")):
            res[k + "+incremental"] = run(k, *v, reps, incremental=True)
    print(json.dumps(res, indent=1))
