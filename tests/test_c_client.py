"""The ABI is plain C: a gcc-built client (no Python, no torch in the process) drives the library."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "discretepomp.jl_amd")


@pytest.fixture(scope="module")
def client(tmp_path_factory, dp):
    exe = str(tmp_path_factory.mktemp("cabi") / "c_abi_client")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi_client.c"), "-o", exe, "-L", LIBDIR, "-ldpomp_hip",
                           "-Wl,-rpath," + LIBDIR])
    return exe


def test_c_client_fails_loudly_without_a_gpu(client, dp):
    if dp.load_library().dpomp_device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([client, "256", "1"], capture_output=True, text=True)
    assert r.returncode == 3 and "no HIP device" in r.stderr and "no CPU path" in r.stderr


@pytest.mark.gpu
def test_c_client_matches_oracle(client, dp, orc, pooley):
    r = subprocess.run([client, "3000", "77"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ll0, ll1, st0, st1, events = r.stdout.split()
    desc = dp.ModelDescriptor(dp.generate_model("SIS", [100, 1]), pooley)
    a = orc.pf_loglik(desc, 3000, [0.003, 0.1], seed=77, seg=0)
    b = orc.pf_loglik(desc, 3000, [0.0032, 0.09], seed=77, seg=1)
    assert float(ll0) == a.ll and float(ll1) == b.ll and int(st0) == 0 and int(st1) == 0
    assert int(events) == a.events + b.events
