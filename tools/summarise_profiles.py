#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/collect_profiles.sh into the small files kept under profiles/:

    kernel_stats.csv        the --stats per-kernel summary of the trace pass
    pmc_hbm_traffic.json    FETCH_SIZE / WRITE_SIZE per launch and kernel; hbm_bytes_per_launch_corrected doubles the
                            FETCH_SIZE figure as MI355X_MICROARCH.md prescribes for gfx950 (counter units: KiB)
    pmc_sq_counters.json    SQ counters per launch and kernel, with derived VALU busy fraction, lane utilisation, clock
    by_launch_shape.json    duration and HBM bytes per kernel and launch shape (grid in workgroups, x by y)

    python tools/summarise_profiles.py <raw dir> <out dir>
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def short(name):
    name = name.replace("void ", "")
    depth, out = 0, []
    for ch in name:          # drop the argument list, keep template arguments
        if ch == "(" and depth == 0:
            break
        depth += ch == "<"
        depth -= ch == ">"
        out.append(ch)
    return "".join(out).strip()


def counters(raw, sub):
    f = newest(os.path.join(raw, sub, "**", "*counter_collection.csv"))
    acc = defaultdict(lambda: defaultdict(list))
    if not f:
        return acc
    for r in csv.DictReader(open(f)):
        if "dpomp" not in r["Kernel_Name"]:
            continue
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def durations(raw, sub):
    f = newest(os.path.join(raw, sub, "**", "*kernel_trace.csv"))
    acc = defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if "dpomp" in r["Kernel_Name"]:
                acc[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return acc


def by_grid(raw):
    """Per kernel AND launch shape (a probe that runs several configurations launches the same kernel with different
    grids): average duration from the trace pass, FETCH_SIZE / WRITE_SIZE from the counter passes."""
    acc = defaultdict(lambda: {"us": [], "fetch": [], "write": []})

    def key(r):  # (kernel, grid of the launch in workgroups: "tiles x filters" for the weight kernels and the gather)
        if "Grid_Size_X" in r:
            nx = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
            ny = int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"]) * (int(r["Grid_Size_Z"]) // int(r["Workgroup_Size_Z"]))
            return short(r["Kernel_Name"]), f"{nx} x {ny} workgroups"
        return short(r["Kernel_Name"]), f"{int(r['Grid_Size']) // int(r['Workgroup_Size'])} workgroups"
    f = newest(os.path.join(raw, "trace", "**", "*kernel_trace.csv"))
    if f:
        for r in csv.DictReader(open(f)):
            if "dpomp" in r["Kernel_Name"]:
                acc[key(r)]["us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for sub, field in (("pmc_fetch", "fetch"), ("pmc_write", "write")):
        f = newest(os.path.join(raw, sub, "**", "*counter_collection.csv"))
        if f:
            # the counter rows carry the total grid only: the x-by-y shape comes from the kernel trace of the same run
            t = newest(os.path.join(raw, sub, "**", "*kernel_trace.csv"))
            shape = {r["Dispatch_Id"]: key(r) for r in csv.DictReader(open(t))} if t else {}
            for r in csv.DictReader(open(f)):
                if "dpomp" in r["Kernel_Name"]:
                    acc[shape.get(r["Dispatch_Id"]) or key(r)][field].append(float(r["Counter_Value"]))
    out = {}
    for (k, g), v in sorted(acc.items()):
        if not v["us"]:
            continue
        avg = lambda x: sum(x) / len(x) if x else 0.0
        out.setdefault(k, {})[g] = {"launches": len(v["us"]), "avg_us": round(avg(v["us"]), 2), "min_us": round(min(v["us"]), 2),
                                    "hbm_bytes_per_launch_corrected": round((2.0 * avg(v["fetch"]) + avg(v["write"])) * 1024.0)}
    return out


def main(raw, out):
    os.makedirs(out, exist_ok=True)
    stats = newest(os.path.join(raw, "trace", "**", "*kernel_stats.csv"))
    if stats:
        shutil.copy(stats, os.path.join(out, "kernel_stats.csv"))
    b = os.path.join(raw, "bench_under_rocprof.json")
    if os.path.exists(b):
        shutil.copy(b, os.path.join(out, "bench_under_rocprof.json"))

    hbm = {}
    fetch, write = counters(raw, "pmc_fetch"), counters(raw, "pmc_write")
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch[k].get("FETCH_SIZE", []), write[k].get("WRITE_SIZE", [])
        fa, wa = (sum(f) / len(f) if f else 0.0), (sum(w) / len(w) if w else 0.0)
        hbm[k] = {"FETCH_SIZE_KB_avg": fa, "launches_FETCH_SIZE": len(f), "WRITE_SIZE_KB_avg": wa,
                  "launches_WRITE_SIZE": len(w), "hbm_bytes_per_launch_corrected": (2.0 * fa + wa) * 1024.0}
    json.dump(hbm, open(os.path.join(out, "pmc_hbm_traffic.json"), "w"), indent=1)
    json.dump(by_grid(raw), open(os.path.join(out, "by_launch_shape.json"), "w"), indent=1)

    sq = defaultdict(dict)
    for sub in ("pmc_sq1", "pmc_sq2", "pmc_sq3"):
        for k, cs in counters(raw, sub).items():
            for name, vals in cs.items():
                sq[k][name] = sum(vals) / len(vals)
        for k, d in durations(raw, sub).items():
            sq[k].setdefault("avg_duration_us_under_pmc", sum(d) / len(d))
    for k, v in sq.items():
        if v.get("SQ_ACTIVE_INST_VALU") and v.get("SQ_THREAD_CYCLES_VALU"):
            v["lane_utilisation"] = v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"])
        # (GRBM_GUI_ACTIVE spans the idle gaps around a dispatch: for launches of a few microseconds the quotient is not
        # a clock, so it is only derived for dispatches of 50 us and more)
        if v.get("GRBM_GUI_ACTIVE") and v.get("avg_duration_us_under_pmc", 0.0) >= 50.0:
            v["clock_GHz"] = v["GRBM_GUI_ACTIVE"] / 8.0 / (v["avg_duration_us_under_pmc"] * 1e3)   # summed over 8 XCDs
        if v.get("SQ_ACTIVE_INST_VALU") and v.get("GRBM_GUI_ACTIVE"):
            # SQ_ACTIVE_INST_VALU counts 4-cycle issue quads; 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
            v["valu_busy_frac"] = v["SQ_ACTIVE_INST_VALU"] * 4.0 / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    json.dump(sq, open(os.path.join(out, "pmc_sq_counters.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({"kernels": sorted(sq), "stats": bool(stats)}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
