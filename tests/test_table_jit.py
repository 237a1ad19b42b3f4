"""Rate tables compiled at create time (the default for models outside the catalogue): the generated source must
reproduce the table interpreter -- and so the oracle -- bit for bit, run every schedule, and be cached."""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from helpers import CATALOGUE_CASES, synthetic_data

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_switch_and_cache_dir_are_exported(dp):
    lib = dp.load_library()
    assert lib.dpomp_set_generic_jit(7) == dp._abi.EINVAL
    assert lib.dpomp_set_generic_jit(0) == 0 and lib.dpomp_set_generic_jit(-1) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name,x0,fd,theta", CATALOGUE_CASES)
def test_compiled_table_equals_interpreter_and_oracle(dp, orc, name, x0, fd, theta):
    model = dp.generate_model(name, x0, freq_dep=fd)
    y = synthetic_data(dp, orc, model, theta)
    jit = dp.ParticleFilter(model, y, force_generic=True, jit=True)
    tab = dp.ParticleFilter(model, y, force_generic=True, jit=False)
    cat = dp.ParticleFilter(model, y)
    assert (jit.evaluator, tab.evaluator, cat.evaluator) == (dp._abi.EVALUATOR_TABLE_COMPILED, dp._abi.EVALUATOR_TABLE,
                                                             dp._abi.EVALUATOR_BUILTIN)
    for variant, P in ((0, 200), (0, 3000), (1, 777), (2, 777), (4, 1500), (16, 1500), (32, 1500), (33, 300)):
        ref = orc.pf_loglik(jit.desc, P, theta, seed=42)
        for pf in (jit, tab, cat):
            pf.set_propagate_variant(variant)
            pop = pf.alloc_population(1, P)
            ll = pf.partial_log_likelihood(pop, theta, 1, len(y), seed=42)
            assert ll[0] == ref.ll and pf.last_event_count() == ref.events
            assert np.array_equal(pop.download(0), ref.pop)
            pop.close()
    sims = [pf.simulate_events(theta, n_sims=8, seed=3)[0] for pf in (jit, tab, cat)]
    assert np.array_equal(sims[0], sims[1]) and np.array_equal(sims[0], sims[2])
    for pf in (jit, tab, cat):
        pf.close()


@pytest.mark.gpu
def test_second_create_of_a_model_comes_from_the_cache(dp, orc, make_obs, tmp_path):
    """First create compiles (seconds), the second finds the code object in this process, and a fresh process finds
    the file."""
    rf = dp.RateFunction([dp.RateTerm(1, ((1,), (2,))), dp.RateTerm(2, ((2,),)), dp.RateTerm(3, ((1, 2),))])
    model = dp.generate_custom_model("SIS-imm", rf, [90, 3], [[-1, 1], [1, -1], [1, 0]])
    y = make_obs([10.0, 20.0], [[0, 20], [0, 40]])
    old = os.environ["DPOMP_CACHE_DIR"]
    os.environ["DPOMP_CACHE_DIR"] = str(tmp_path)
    try:
        t0 = time.perf_counter()
        a = dp.ParticleFilter(model, y, jit=True)
        t_first = time.perf_counter() - t0
        t0 = time.perf_counter()
        b = dp.ParticleFilter(model, y, jit=True)
        t_second = time.perf_counter() - t0
        assert a.evaluator == b.evaluator == dp._abi.EVALUATOR_TABLE_COMPILED
        assert t_second < 0.05, (t_first, t_second)
        files = [f for f in os.listdir(tmp_path) if f.endswith(".dpomprtc")]
        assert len(files) == 1
        ref = orc.pf_loglik(a.desc, 500, [0.004, 0.1, 0.01], seed=1)
        assert a.loglik([0.004, 0.1, 0.01], 500, seed=1)[0] == ref.ll == b.loglik([0.004, 0.1, 0.01], 500, seed=1)[0]
        a.close()
        b.close()
        # a fresh process: the file is found, nothing is compiled (well under the ~2 s a compile takes)
        code = ("import sys, time; sys.path.insert(0, %r); import dpomp_amd as dp\n"
                "rf = dp.RateFunction([dp.RateTerm(1, ((1,), (2,))), dp.RateTerm(2, ((2,),)), dp.RateTerm(3, ((1, 2),))])\n"
                "m = dp.generate_custom_model('SIS-imm', rf, [90, 3], [[-1, 1], [1, -1], [1, 0]])\n"
                "y = [dp.Observation(10.0, 1, 1.0, [0, 20]), dp.Observation(20.0, 1, 1.0, [0, 40])]\n"
                "dp.ParticleFilter(dp.generate_model('SIS', [100, 1]), y).close()\n"   # GPU + library warm
                "t = time.perf_counter(); pf = dp.ParticleFilter(m, y, jit=True); dt = time.perf_counter() - t\n"
                "print('EVAL', pf.evaluator, 'DT', dt, 'LL', repr(float(pf.loglik([0.004, 0.1, 0.01], 500, seed=1)[0])))\n" % ROOT)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, DPOMP_CACHE_DIR=str(tmp_path)))
        assert r.returncode == 0, r.stderr[-2000:]
        words = r.stdout.split()
        assert int(words[words.index("EVAL") + 1]) == dp._abi.EVALUATOR_TABLE_COMPILED
        assert float(words[words.index("DT") + 1]) < 0.05, r.stdout
        assert float(words[words.index("LL") + 1]) == ref.ll
        # a damaged cache file is ignored and replaced
        path = os.path.join(tmp_path, files[0])
        open(path, "wb").write(b"DPOMPRT2 garbage")
        dp.load_library().dpomp_set_generic_jit(1)
    finally:
        os.environ["DPOMP_CACHE_DIR"] = old
        dp.load_library().dpomp_set_generic_jit(-1)


@pytest.mark.gpu
def test_user_source_models_run_every_schedule(dp, orc, pooley, tmp_path):
    src = """__device__ void dpomp_user_rates(const double* theta, const int* x, double* r) {
    r[0] = theta[0] * (double)x[0] * (double)x[1];
    r[1] = theta[1] * (double)x[1];
}"""
    model = dp.generate_custom_model("SIS", dp.RateSource(src, 2, 2), [100, 1], [[-1, 1], [1, -1]])
    pf = dp.ParticleFilter(model, pooley)
    assert pf.evaluator == dp._abi.EVALUATOR_SOURCE
    sis = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), pooley)
    for variant in (1, 2, 4, 16, 32, 33):
        pf.set_propagate_variant(variant)
        assert pf.loglik([0.003, 0.1], 5000, seed=7)[0] == sis.loglik([0.003, 0.1], 5000, seed=7)[0]
    pf.close()
    sis.close()
