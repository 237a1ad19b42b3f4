#!/usr/bin/env python3
"""Per observation interval: duration of the propagation launch (rocprofv3 kernel trace), events executed, lane
utilisation of a one-particle-per-lane schedule, and the achieved fraction of the model's VALU-issue ceiling
(profiles/r03/valu_ceilings.json from tools/isa_loop_count.py) -- for the single-GPU shapes of the benchmark
configurations.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/itrace -- python3 tools/interval_trace.py run gpurun_out/itrace_plan.json
    python3 tools/interval_trace.py parse gpurun_out/itrace gpurun_out/itrace_plan.json > profiles/r03/interval_trace.json
"""
import csv
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WARM, MEASURED = 4, 6
POOLEY = [(20.0, [0, 18]), (40.0, [0, 65]), (60.0, [0, 70]), (80.0, [0, 66]), (100.0, [0, 67])]


def configs(dp):
    rng = np.random.default_rng(1)
    th64 = np.stack([rng.uniform(0.003, 0.008, 64), rng.uniform(0.1, 0.3, 64), rng.uniform(0.05, 0.2, 64)])
    seir_y = dp.get_observations(os.path.join(ROOT, "tests/golden/seir_synthetic.csv"))
    lv_y = dp.get_observations(os.path.join(ROOT, "tests/golden/lotka_synthetic.csv"))
    sis_y = [dp.Observation(t, 1, 1.0, v) for t, v in POOLEY]
    return [
        ("config2_sis_1x2^20", "sis", dp.generate_model("SIS", [100, 1]), sis_y, np.array([[0.003], [0.1]]), 1 << 20),
        ("config3_seir_4x2^18", "seir", dp.generate_model("SEIR", [100, 0, 1, 0]), seir_y,
         np.tile(np.array([0.005, 0.2, 0.1])[:, None], (1, 4)) * (1 + 0.05 * np.arange(4)), 1 << 18),
        ("config4_share_seir_64x2^16", "seir", dp.generate_model("SEIR", [100, 0, 1, 0]), seir_y, th64, 1 << 16),
        ("config5_lotka_1x2^20", "lotka", dp.generate_model("LOTKA", [70, 70]), lv_y, np.array([[0.5], [0.0025], [0.3]]), 1 << 20),
        ("smc2_default_sis_4000x200", "sis", dp.generate_model("SIS", [100, 1]), sis_y, np.tile(np.array([[0.003], [0.1]]), (1, 4000)), 200),
    ]


def run(plan_file):
    import dpomp_amd as dp
    plan = []
    for name, family, model, y, theta, P in configs(dp):
        pf = dp.ParticleFilter(model, y)
        T = len(y)
        for i in range(WARM + MEASURED):
            pf.loglik(theta, P, seed=10 + i)
        util = [float(u) for u in pf.lane_utilisation()]
        pop = pf.alloc_population(theta.shape[1], P)
        events = []
        for k in range(1, T + 1):  # one interval per call: the events of that interval (same seed as the last full call)
            pf.partial_log_likelihood(pop, theta, k, k, seed=10 + WARM + MEASURED - 1)
            events.append(int(pf.last_event_count()))
        pop.close()
        pf.close()
        plan.append({"name": name, "family": family, "T": T, "B": int(theta.shape[1]), "P": int(P), "events": events, "lane_utilisation": util})
    json.dump({"warm": WARM, "measured": MEASURED, "configs": plan}, open(plan_file, "w"))


def parse(raw, plan_file):
    plan = json.load(open(plan_file))
    f = max(glob.glob(os.path.join(raw, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if "k_propagate" in r["Kernel_Name"] or "k_filter_small" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = []  # per propagation: [main launch us, kernel, tail launch us (spilled contexts, if the interval spilled)]
    for r in rows:
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        name = r["Kernel_Name"].split("(")[0].replace("void dpomp::", "")
        if name.startswith("k_propagate_c") and name.rstrip().endswith(", 2>"):
            dur[-1][2] += us
        else:
            dur.append([us, name, 0.0])
    ceil_file = os.path.join(ROOT, "profiles", "r03", "valu_ceilings.json")
    ceilings = {}
    if os.path.exists(ceil_file):
        for label, v in json.load(open(ceil_file))["kernels"].items():
            if "k_propagate_c" in label:
                ceilings[label.split(" ")[0]] = v["ceiling_events_per_s"]
    pos, out = 0, {}
    for c in plan["configs"]:
        T = c["T"]
        pos += plan["warm"] * T
        per = np.zeros(T)
        tail = np.zeros(T)
        kern = [None] * T
        for i in range(plan["measured"]):
            for k in range(T):
                per[k] += dur[pos][0] + dur[pos][2]
                tail[k] += dur[pos][2]
                kern[k] = dur[pos][1]
                pos += 1
        per /= plan["measured"]
        tail /= plan["measured"]
        pos += T  # the single-interval calls
        ceiling = ceilings.get(c["family"])
        ev = np.array(c["events"], dtype=float)
        rate = ev / (per * 1e-6)
        out[c["name"]] = {
            "B": c["B"], "P": c["P"],
            "propagate_us_per_interval": [round(float(v), 1) for v in per],
            "of_which_tail_launch_us": [round(float(v), 1) for v in tail],
            "kernel_per_interval": kern,
            "events_per_interval": c["events"],
            "events_per_s_per_interval": [float(f"{v:.4g}") for v in rate],
            "lane_utilisation_static_schedule": [round(u, 3) for u in c["lane_utilisation"]],
            "propagate_us_per_call": round(float(per.sum()), 1),
            "events_per_s": float(f"{ev.sum() / (per.sum() * 1e-6):.4g}"),
            "valu_ceiling_events_per_s": ceiling,
            "valu_frac": round(float(ev.sum() / (per.sum() * 1e-6) / ceiling), 3) if ceiling else None,
            "valu_frac_per_interval": [round(float(v / ceiling), 3) for v in rate] if ceiling else None,
        }
    assert pos == len(dur), (pos, len(dur))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        parse(sys.argv[2], sys.argv[3])
