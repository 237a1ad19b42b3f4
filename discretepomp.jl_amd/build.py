"""Builds the HIP extension in-tree: ``libdpomp_hip.so`` (gfx950) and the host-compiled
numerics checker ``libdpomp_hostcheck.so`` (same header, plain C++, used by CPU tests only)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force: bool = False, verbose: bool = True):
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "dpomp.h")]
    lib = os.path.join(HERE, "libdpomp_hip.so")
    if force or _stale(lib, deps):
        cmd = [HIPCC, "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
               "-Wall", "-Wno-unused-function", "-o", lib, os.path.join(CSRC, "dpomp_api.hip")]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    chk = os.path.join(HERE, "libdpomp_hostcheck.so")
    if force or _stale(chk, deps):
        fma = ["-mfma"] if " fma" in open("/proc/cpuinfo").read() else []
        cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", *fma, "-fPIC", "-shared", "-o", chk,
               os.path.join(CSRC, "numerics_hostcheck.cpp")]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv)
