#!/usr/bin/env python3
"""Where the event loop of the kernels that run as lone waves starts within a 32-byte instruction-fetch block
(DPOMP_PAD_SMALL: whole-filter kernel, DPOMP_PAD_STATIC: static propagation kernel; csrc/dpomp_kernels.h).

    python tools/loop_offset_sweep.py build SMALL|STATIC     (here: eight libraries under build_variants/)
    python tools/loop_offset_sweep.py run SMALL|STATIC        (GPU box: microseconds per filter call for each)

Sizes timed: 1 x 200 and 64 x 200 particles (whole-filter kernel), 1 x 1000, 1 x 4096, 1 x 2^14, 1 x 2^16 (static
kernel), Pooley SIS."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = [(1, 200), (64, 200), (1, 1000), (1, 4096), (1, 1 << 14), (1, 1 << 16)]


def lib(which, n):
    return os.path.join(ROOT, "build_variants", f"libdpomp_pad_{which.lower()}{n}.so")


def build(which):
    os.makedirs(os.path.join(ROOT, "build_variants"), exist_ok=True)
    procs = []
    for n in range(8):
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
               "-Wno-unused-function", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", f"-DDPOMP_PAD_{which}={n}",
               "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-o", lib(which, n),
               os.path.join(ROOT, "discretepomp.jl_amd", "csrc", "dpomp_api.hip"), "-ldl"]
        procs.append(subprocess.Popen(cmd, stderr=subprocess.DEVNULL))
    assert all(p.wait() == 0 for p in procs)


def one():
    import numpy as np
    sys.path.insert(0, ROOT)
    import dpomp_amd as dp
    pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), dp.get_observations(os.path.join(ROOT, "tests/golden/pooley.csv")))
    out = []
    for B, P in SIZES:
        th = np.tile(np.array([[0.003], [0.1]]), (1, B))
        for i in range(5):
            pf.loglik(th, P, seed=1)
        t = time.perf_counter()
        for i in range(40):
            pf.loglik(th, P, seed=i)
        out.append((time.perf_counter() - t) / 40 * 1e6)
    print(" ".join(f"{v:7.1f}" for v in out), flush=True)


def run(which):
    print("nops  " + " ".join(f"{B}x{P}".rjust(7) for B, P in SIZES), flush=True)
    for rep in range(2):
        for n in range(8):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, DPOMP_HIP_LIB=lib(which, n)),
                               capture_output=True, text=True, timeout=200)
            print(f"{n:4d}  " + (r.stdout.strip().splitlines() or ["failed"])[-1], flush=True)
            if r.returncode != 0:
                sys.exit(1)


if __name__ == "__main__":
    {"build": build, "run": run}.get(sys.argv[1], lambda *_: one())(*sys.argv[2:3])
