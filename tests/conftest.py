import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Rate tables outside the catalogue are compiled at create time by default (seconds per distinct model on first use).
# The suite creates hundreds of table models, so it runs them in the device interpreter unless a test asks for the
# compiled path (ParticleFilter(..., jit=True): tests/test_table_jit.py); compiled models of a run are cached in a
# directory of its own.
os.environ.setdefault("DPOMP_GENERIC_JIT", "0")
if "DPOMP_CACHE_DIR" not in os.environ:  # a private directory of this run (the library refuses a cache it does not own)
    import atexit
    import shutil
    import tempfile
    os.environ["DPOMP_CACHE_DIR"] = tempfile.mkdtemp(prefix="dpomp_cache_")
    atexit.register(shutil.rmtree, os.environ["DPOMP_CACHE_DIR"], ignore_errors=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # make sure the native pieces exist (cheap when up to date; hipcc cross-compiles without a GPU)
    import __graft_entry__ as ge
    ge.build()


@pytest.fixture(scope="session")
def dp():
    import dpomp_amd
    return dpomp_amd


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def pooley(dp):
    return dp.get_observations(os.path.join(GOLDEN, "pooley.csv"))


def synthetic_obs(dp, times, values, obs_ids=None):
    obs_ids = obs_ids or [1] * len(times)
    return [dp.Observation(float(t), int(i), 1.0, [int(x) for x in v]) for t, i, v in zip(times, obs_ids, values)]


@pytest.fixture(scope="session")
def make_obs(dp):
    return lambda times, values, obs_ids=None: synthetic_obs(dp, times, values, obs_ids)
