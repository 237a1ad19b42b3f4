#!/usr/bin/env python3
"""Lane utilisation of propagation schedules, modelled on the oracle's per-particle event counts (CPU only):
the compaction kernel as built (256 particles per workgroup, live-lane counts after 32 / 64 / 12 events, packing into
ceil(live / 64) waves), and a two-launch variant that stops every particle after K events and packs the survivors of
the whole launch densely into a second launch (a device-wide re-pack).  Cost = wave-level event iterations issued.

    python tools/schedule_model.py [log2 particles]      -> JSON
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dpomp_amd as dp  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def compaction_wave_steps(n):
    """Wave-level event iterations the compaction kernel issues for one workgroup whose particles execute n[i] events."""
    rem = np.array(n, dtype=np.int64)
    slots = np.full(256, -1, dtype=np.int64)
    slots[:len(rem)] = np.arange(len(rem))
    n_start = len(rem)
    steps, chunk = 0, 32
    while True:
        live_any = False
        for w in range(4):
            idx = slots[64 * w:64 * w + 64]
            idx = idx[idx >= 0]
            if idx.size and rem[idx].max() > 0:
                steps += int(min(chunk, rem[idx].max()))
                live_any = True
        if not live_any:
            break
        rem = np.maximum(rem - chunk, 0)
        live = np.nonzero(rem > 0)[0]
        total = live.size
        if total == 0:
            break
        chunk = (1 << 30 if total <= 64 else 12) if total < n_start else 64
        waves_now = sum(1 for w in range(4) if np.any((slots[64 * w:64 * w + 64] >= 0) & (rem[np.maximum(slots[64 * w:64 * w + 64], 0)] > 0)))
        if (total + 63) // 64 < waves_now:
            slots[:] = -1
            slots[:total] = live
    return steps


def launch_steps(counts):
    return sum(compaction_wave_steps(counts[i:i + 256]) for i in range(0, len(counts), 256))


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    P = 1 << lg
    G = os.path.join(ROOT, "tests", "golden")
    out = {}
    for name, x0, theta, csv in (("SIS", [100, 1], [0.003, 0.1], "pooley.csv"), ("SEIR", [100, 0, 1, 0], [0.005, 0.2, 0.1], "seir_synthetic.csv")):
        y = dp.get_observations(os.path.join(G, csv))
        desc = dp.ModelDescriptor(dp.generate_model(name, x0), y)
        ref = orc.pf_loglik(desc, P, theta, seed=5, trace=True, nthreads=8)
        rows = []
        for k in range(len(y)):
            n = ref.particle_events[k].astype(np.int64)
            ev = int(n.sum())
            static = sum(int(n[i:i + 64].max()) for i in range(0, P, 64))
            now = launch_steps(n)
            row = {"interval": k + 1, "events": ev, "static": round(ev / (64 * static), 3), "compaction_as_built": round(ev / (64 * now), 3)}
            for K in (16, 32, 64):
                first = sum(int(np.minimum(n[i:i + 64], K).max()) for i in range(0, P, 64))  # no count before 32 events anyway
                surv = n[n > K] - K
                second = launch_steps(surv) if surv.size else 0
                row[f"repack_after_{K}"] = round(ev / (64 * (first + second)), 3)
                row[f"survivors_after_{K}"] = round(surv.size / P, 3)
            rows.append(row)
        out[name] = rows
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
