#!/usr/bin/env python3
"""Where a filter call's time goes on the single-GPU shapes of the benchmark configurations (Pooley SIS 2^20, SEIR 4 x 2^18,
SEIR 64 x 2^16, Lotka-Volterra 2^20, SIS 4000 x 200): wall time per call without
profiling, then per-kernel average durations (HIP events carrying each kernel's own begin and end) and their sum, and
the propagation kernel's event rate against the VALU-issue ceiling of the model's event loop (valu_frac; ceilings from
the ISA of the build: tools/isa_loop_count.py -> profiles/*/valu_ceilings.json).
Usage: python tools/config_profile.py [reps]   -> JSON on stdout."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import dpomp_amd as dp  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def ceilings():
    """{family: (VALU instructions per event, events/s ceiling)} of the compaction kernel, latest committed profile."""
    out = {}
    root = "profiles"
    for d in sorted(os.listdir(root)):
        f = os.path.join(root, d, "valu_ceilings.json")
        if os.path.exists(f):
            for label, v in json.load(open(f))["kernels"].items():
                if "k_propagate_c" in label and "ceiling_events_per_s" in v:
                    out[label.split(" ")[0]] = (v["valu"], v["ceiling_events_per_s"])
    return out


CEIL = ceilings()
FAMILY = {"config2": "sis", "config3": "seir", "config4": "seir", "config5": "lotka", "smc2": "sis"}
POOLEY = [(20.0, [0, 18]), (40.0, [0, 65]), (60.0, [0, 70]), (80.0, [0, 66]), (100.0, [0, 67])]
rng = np.random.default_rng(1)
th64 = np.stack([rng.uniform(0.003, 0.008, 64), rng.uniform(0.1, 0.3, 64), rng.uniform(0.05, 0.2, 64)])
seir_y = dp.get_observations("tests/golden/seir_synthetic.csv")
lv_y = dp.get_observations("tests/golden/lotka_synthetic.csv")
CONFIGS = [
    ("config2_sis_1x2^20", dp.generate_model("SIS", [100, 1]), [dp.Observation(t, 1, 1.0, v) for t, v in POOLEY], np.array([[0.003], [0.1]]), 1 << 20),
    ("config3_seir_4x2^18", dp.generate_model("SEIR", [100, 0, 1, 0]), seir_y, np.tile(np.array([0.005, 0.2, 0.1])[:, None], (1, 4)) * (1 + 0.05 * np.arange(4)), 1 << 18),
    ("config4_share_seir_64x2^16", dp.generate_model("SEIR", [100, 0, 1, 0]), seir_y, th64, 1 << 16),
    ("config5_lotka_1x2^20", dp.generate_model("LOTKA", [70, 70]), lv_y, np.array([[0.5], [0.0025], [0.3]]), 1 << 20),
    ("smc2_default_sis_4000x200", dp.generate_model("SIS", [100, 1]), [dp.Observation(t, 1, 1.0, v) for t, v in POOLEY],
     np.tile(np.array([[0.003], [0.1]]), (1, 4000)), 200),
]
out = {}
for name, model, y, theta, P in CONFIGS:
    pf = dp.ParticleFilter(model, y)
    for i in range(3):
        pf.loglik(theta, P, seed=i)
    t0 = time.perf_counter()
    for i in range(reps):
        pf.loglik(theta, P, seed=10 + i)
    wall = (time.perf_counter() - t0) / reps
    pf.set_profiling(1)
    for i in range(reps):
        pf.loglik(theta, P, seed=10 + i)
    stats = pf.kernel_stats()
    pf.set_profiling(0)
    ks = {k: {"launches_per_call": n / reps, "avg_us": round(ms / n * 1e3, 2), "us_per_call": round(ms / reps * 1e3, 1)}
          for k, (n, ms) in stats.items() if n}
    B, T = theta.shape[1], len(y)
    out[name] = {"ms_per_call": round(wall * 1e3, 4), "particle_steps_per_s": B * P * T / wall,
                 "kernel_us_per_call": round(sum(v["us_per_call"] for v in ks.values()), 1), "kernels": ks,
                 "events_per_call": pf.last_event_count(), "lane_utilisation": [round(float(u), 3) for u in pf.lane_utilisation()]}
    fam = FAMILY[name.split("_")[0]]
    if fam in CEIL and "k_propagate" in ks:
        rate = pf.last_event_count() / (ks["k_propagate"]["us_per_call"] * 1e-6)
        out[name]["propagate"] = {"events_per_s": float(f"{rate:.4g}"), "valu_instructions_per_event": CEIL[fam][0],
                                  "valu_ceiling_events_per_s": float(f"{CEIL[fam][1]:.4g}"), "valu_frac": round(rate / CEIL[fam][1], 3)}
    pf.close()
print(json.dumps(out, indent=1))
