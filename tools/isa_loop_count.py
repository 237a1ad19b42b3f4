#!/usr/bin/env python3
"""VALU-issue ceiling of the Gillespie event loop, per rate evaluator, from the ISA the library is built from.

Compiles the library's device code to gfx950 assembly (same options as discretepomp.jl_amd/build.py), finds in each
requested kernel the event loop -- the innermost loop that holds the short division (v_rcp_f64 without v_div_scale_f64)
-- and counts its instructions by issue class.  Issue cost per wave64 instruction as measured on MI355X with 8 waves per
SIMD (tools/valu_rate_probe.hip -> profiles/r01_final/valu_issue_rates.json): FP64, 64-bit integer, three-operand and
compare/select 32-bit instructions one 4-cycle slot; plain two-operand 32-bit integer instructions 0.65 slot;
v_rcp_f64 / v_sqrt_f64 3.4 slots.  One loop iteration = one event of every live lane of the wave, so

    ceiling = SIMDs * clock * 64 lanes / (4 cycles * slots per iteration)   events/s

Usage: python tools/isa_loop_count.py [--asm file.s] [--out profiles/r03/valu_ceilings.json]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS, CLOCK_HZ = 1024, 2.4e9
LIGHT = {"v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_or_b32", "v_and_b32", "v_lshlrev_b32", "v_lshrrev_b32",
         "v_ashrrev_i32", "v_mov_b32", "v_not_b32"}
TRANS = {"v_rcp_f64", "v_sqrt_f64", "v_rsq_f64"}

# kernel (mangled-name fragment) per BASELINE workload
KERNELS = {
    "sis (configs 1, 2; k_propagate_c<RSir<2>,32>)": "k_propagate_cINS_4RSirILi2ELb0EEELi32ELi0EE",
    "seir (configs 3, 4; k_propagate_c<RSeir<4>,32>)": "k_propagate_cINS_5RSeirILi4ELb0EEELi32ELi0EE",
    "lotka (config 5; k_propagate_c<RLotka,32>)": "k_propagate_cINS_6RLotkaELi32ELi0EE",
    "sis-flat (filters laid end to end; k_propagate_c<RSir<2>,32,kModeFlat>)": "k_propagate_cINS_4RSirILi2ELb0EEELi32ELi1EE",
    "sis static schedule (k_propagate<RSir<2>>)": "k_propagateINS_4RSirILi2ELb0EEEE",
    "seir static schedule (k_propagate<RSeir<4>>)": "k_propagateINS_5RSeirILi4ELb0EEEE",
    "lotka static schedule (k_propagate<RLotka>)": "k_propagateINS_6RLotkaEEE",
}


def device_asm(path=None):
    if path:
        return open(path).read()
    out = os.path.join(tempfile.mkdtemp(prefix="dpomp_isa_"), "dpomp_dev.s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-mllvm",
           "-amdgpu-sched-strategy=iterative-ilp", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-S",
           "--cuda-device-only", "-o", out, os.path.join(ROOT, "discretepomp.jl_amd", "csrc", "dpomp_api.hip")]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return open(out).read()


def kernel_body(asm, frag):
    m = re.search(r"^(_ZN5dpomp\d+%s\w*):" % re.escape(frag), asm, re.M)
    if not m:
        return None, None
    name = m.group(1)
    end = asm.index(".Lfunc_end", m.end())
    meta = {}
    tail = asm[end:end + 6000]
    for key in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy", "LDSByteSize"):
        mm = re.search(r"; %s: (\d+)" % key, tail)
        if mm:
            meta[key] = int(mm.group(1))
    return asm[m.end():end].splitlines(), meta


def blocks_of(lines):
    """Basic blocks of a kernel as (name, loop header it belongs to or None, is an inner-loop header, instruction lines),
    from the compiler's own block comments (`.LBBn_m:  ; in Loop: Header=BBn_k Depth=d`, `; %bb.m: ...`)."""
    blocks, cur = [], None
    for i, ln in enumerate(lines):
        mm = re.match(r"^(?:\.L(BB\d+_\d+):|; %bb\.(\d+):)(.*)$", ln)
        if mm:
            name = mm.group(1) or ("bb." + mm.group(2))
            rest = mm.group(3)
            hdr = re.search(r"Header=(BB\d+_\d+)", rest)
            inner = "Inner Loop Header" in rest
            # the header's own annotation sits on the following comment line(s)
            j = i + 1
            while j < len(lines) and re.match(r"^\s*;", lines[j]) and not re.match(r"^; %bb", lines[j]):
                inner = inner or "Inner Loop Header" in lines[j]
                j += 1
            cur = {"name": name, "header": hdr.group(1) if hdr else None, "inner": inner, "ins": []}
            blocks.append(cur)
        elif cur is not None and re.match(r"^\s+[a-z]", ln):
            cur["ins"].append(ln)
    return blocks


def event_loop(lines):
    """Instruction lines of the event loop: the innermost loop with the short division (v_rcp_f64, no v_div_scale_f64)."""
    blocks = blocks_of(lines)
    best = None
    for h in [b for b in blocks if b["inner"]]:
        body = [ln for b in blocks if b is h or b["header"] == h["name"] for ln in b["ins"]]
        ops = [ln.split()[0] for ln in body]
        if any(o.startswith("v_rcp_f64") for o in ops) and not any(o.startswith("v_div_scale_f64") for o in ops):
            if best is None or len(body) < len(best):
                best = body
    return best


def classify(body):
    c = {"valu": 0, "valu_light": 0, "valu_trans": 0, "salu": 0, "lds": 0, "vmem": 0, "waitcnt_nop": 0, "branch": 0}
    mix = {}
    for ln in body:
        op = ln.split()[0]
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
        if op.startswith("v_"):
            c["valu"] += 1
            mix[base] = mix.get(base, 0) + 1
            if base in LIGHT:
                c["valu_light"] += 1
            elif base in TRANS:
                c["valu_trans"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
        elif op.startswith(("s_waitcnt", "s_nop")):
            c["waitcnt_nop"] += 1
        elif op.startswith(("s_cbranch", "s_branch")):
            c["branch"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    full = c["valu"] - c["valu_light"] - c["valu_trans"]
    slots = full + 0.65 * c["valu_light"] + 3.4 * c["valu_trans"]
    c["issue_slots_per_event"] = round(slots, 2)
    c["cycles_per_wave_event"] = round(4 * slots, 1)
    c["ceiling_events_per_s"] = SIMDS * CLOCK_HZ * 64 / (4 * slots)
    c["valu_mix"] = dict(sorted(mix.items(), key=lambda kv: -kv[1]))
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", default=None, help="device assembly already produced (hipcc -S --cuda-device-only)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    asm = device_asm(args.asm)
    out = {"issue_model": "4-cycle slot per wave64 VALU instruction; plain two-operand 32-bit integer ops 0.65 slot; "
                          "v_rcp_f64 3.4 slots (profiles/r01_final/valu_issue_rates.json)",
           "simds": SIMDS, "clock_hz": CLOCK_HZ, "kernels": {}}
    for label, frag in KERNELS.items():
        lines, meta = kernel_body(asm, frag)
        if lines is None:
            out["kernels"][label] = {"error": "kernel not found"}
            continue
        loop = event_loop(lines)
        if loop is None:
            out["kernels"][label] = {"error": "event loop not found", **meta}
            continue
        c = classify(loop)
        trips = max(1, c["valu_mix"].get("v_rcp_f64", 1))  # events per trip of the loop (the lone-wave kernels run two)
        if trips > 1:
            for key in ("valu", "valu_light", "valu_trans", "salu", "lds", "vmem", "waitcnt_nop", "branch"):
                c[key] = c[key] / trips
            c["issue_slots_per_event"] = round(c["issue_slots_per_event"] / trips, 2)
            c["cycles_per_wave_event"] = round(c["cycles_per_wave_event"] / trips, 1)
            c["ceiling_events_per_s"] *= trips
        out["kernels"][label] = {**meta, "loop_instructions": len(loop), "events_per_trip": trips, **c}
    text = json.dumps(out, indent=1)
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        open(args.out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    sys.exit(main())
