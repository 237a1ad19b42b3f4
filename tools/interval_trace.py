#!/usr/bin/env python3
"""Per observation interval: duration of the propagation launch (rocprofv3 kernel trace), events executed, lane
utilisation of a one-particle-per-lane schedule, and the achieved fraction of the model's VALU-issue ceiling
(profiles/r03/valu_ceilings.json from tools/isa_loop_count.py) -- for the single-GPU shapes of the benchmark
configurations.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/itrace -- python3 tools/interval_trace.py run gpurun_out/itrace_plan.json
    python3 tools/interval_trace.py parse gpurun_out/itrace gpurun_out/itrace_plan.json > profiles/r03/interval_trace.json

SQ counters per interval (each --pmc pass a run of its own, directories <raw>/sq1, sq2, sq3 as tools/collect_profiles_r03.sh
makes them):
    python3 tools/interval_trace.py parse_pmc <raw> <plan.json> > profiles/r03/interval_sq_counters.json
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


def parse_pmc(raw, plan_file):
    """Per configuration and interval: SQ counters of the propagation launch, averaged over the measured calls, with
    the derived figures of tools/summarise_profiles.py (VALU busy fraction of the chip, lane utilisation of the issued
    VALU instructions, VALU instructions per event and wave)."""
    plan = json.load(open(plan_file))
    per_pass = {}
    for sub in sorted(os.listdir(raw)):
        files = glob.glob(os.path.join(raw, sub, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        rows = {}
        for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
            if "k_propagate" not in r["Kernel_Name"] and "k_filter_small" not in r["Kernel_Name"]:
                continue
            rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        per_pass[sub] = [rows[k] for k in sorted(rows)]
    out = {}
    for sub, disp in per_pass.items():
        pos = 0
        for c in plan["configs"]:
            T = c["T"]
            pos += plan["warm"] * T
            acc = [dict() for _ in range(T)]
            for i in range(plan["measured"]):
                for k in range(T):
                    for name, v in disp[pos].items():
                        acc[k][name] = acc[k].get(name, 0.0) + v / plan["measured"]
                    pos += 1
            pos += T
            o = out.setdefault(c["name"], {"events_per_interval": c["events"], "intervals": [dict() for _ in range(T)]})
            for k in range(T):
                o["intervals"][k].update(acc[k])
        assert pos == len(disp), (sub, pos, len(disp))
    for name, o in out.items():
        for k, v in enumerate(o["intervals"]):
            ev = o["events_per_interval"][k]
            if v.get("SQ_ACTIVE_INST_VALU") and v.get("SQ_THREAD_CYCLES_VALU"):
                v["lane_utilisation_issued"] = round(v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]), 3)
            if v.get("SQ_ACTIVE_INST_VALU") and v.get("GRBM_GUI_ACTIVE"):
                v["valu_busy_frac"] = round(v["SQ_ACTIVE_INST_VALU"] * 4.0 / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 3)
            if v.get("SQ_INSTS_VALU") and ev:
                v["valu_instructions_per_event_lane"] = round(v["SQ_THREAD_CYCLES_VALU"] / 4.0 / ev, 1) if v.get("SQ_THREAD_CYCLES_VALU") else None
                v["wave_valu_instructions_per_64_events"] = round(v["SQ_INSTS_VALU"] / (ev / 64.0), 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    elif sys.argv[1] == "parse_pmc":
        parse_pmc(sys.argv[2], sys.argv[3])
    else:
        parse(sys.argv[2], sys.argv[3])
