"""Run-time compiled rate functions (hipRTC): the pluggable on-device rate evaluator for models outside the
RateTerm table -- the counterpart of the reference's arbitrary `rate_function` closure (hmm_structs.jl:109)."""
import numpy as np
import pytest

# SIS with saturating incidence and an immigration term: neither fits ((theta*f0)*f1)/d nor theta*((f0*f1)/d)
SOURCE = """
__device__ void dpomp_user_rates(const double* theta, const int* x, double* r) {
    const double S = (double)x[0], I = (double)x[1];
    r[0] = theta[0] * S * I / (1.0 + theta[2] * I) + theta[3];   // infection, saturating in I, plus import
    r[1] = theta[1] * I;                                          // recovery
}
"""


def _host(out, th, x):
    out[0] = th[0] * x[0] * x[1] / (1.0 + th[2] * x[1]) + th[3]
    out[1] = th[1] * x[1]
    return out


def _model(dp):
    rf = dp.RateSource(SOURCE, n_events=2, n_params=4, host=_host)
    # (the reference's default prior has one dimension per event -- generate_weak_prior(size(m_transition, 1)),
    #  hmm_examples.jl:237 -- this model has four parameters)
    return dp.generate_custom_model("SIS-sat", rf, [100, 1], [[-1, 1], [1, -1]], prior=dp.generate_weak_prior(4))


def test_rate_source_compiles_without_a_gpu(dp, pooley):
    log = dp.check_rate_source(_model(dp), pooley)
    assert "error" not in log.lower()
    d = dp.ModelDescriptor(_model(dp), pooley)
    assert d.c.builtin_id == 100 and d.n_params == 4


def test_broken_rate_source_reports_the_compiler_diagnostics(dp, pooley):
    bad = dp.RateSource("__device__ void dpomp_user_rates(const double* theta, const int* x, double* r) { r[0] = undefined_symbol; }",
                        n_events=2, n_params=2)
    model = dp.generate_custom_model("bad", bad, [100, 1], [[-1, 1], [1, -1]])
    with pytest.raises(dp.DpompError) as ei:
        dp.check_rate_source(model, pooley)
    assert "undefined_symbol" in str(ei.value) and "rate_source" in str(ei.value)   # line numbers refer to the user's text


def test_oracle_runs_the_same_source_on_the_host(dp, orc, pooley, tmp_path):
    orc.set_user_rates_from_source(SOURCE, str(tmp_path))
    desc = dp.ModelDescriptor(_model(dp), pooley)
    th, x = [0.003, 0.1, 0.01, 0.02], [90, 11]
    assert np.array_equal(orc.rates(desc, th, x), _host(np.zeros(2), th, x))
    r = orc.pf_loglik(desc, 512, th, seed=3)
    assert np.isfinite(r.ll) and r.events > 0


@pytest.mark.gpu
@pytest.mark.parametrize("variant,P", [(1, 3000), (32, 3000), (33, 300)])
def test_rate_source_filter_matches_oracle(dp, orc, pooley, tmp_path, variant, P):
    orc.set_user_rates_from_source(SOURCE, str(tmp_path))
    pf = dp.ParticleFilter(_model(dp), pooley)
    pf.set_propagate_variant(variant)
    theta = np.array([[0.003, 0.0035], [0.1, 0.12], [0.01, 0.02], [0.02, 0.0]])
    pop = pf.alloc_population(2, P)
    ll = pf.partial_log_likelihood(pop, theta, 1, pf.n_obs, seed=13)
    for b in range(2):
        ref = orc.pf_loglik(pf.desc, P, theta[:, b], seed=13, seg=b)
        assert ll[b] == ref.ll and np.array_equal(pop.download(b), ref.pop)
    assert pf.last_event_count() == sum(orc.pf_loglik(pf.desc, P, theta[:, b], seed=13, seg=b).events for b in range(2))
    pf.close()


@pytest.mark.gpu
def test_rate_source_simulation_matches_oracle(dp, orc, tmp_path):
    orc.set_user_rates_from_source(SOURCE, str(tmp_path))
    model = _model(dp)
    res = dp.gillespie_sim(model, [0.003, 0.1, 0.01, 0.02], n_sims=2, seed=5, verbose=False)
    desc = dp.ModelDescriptor(model, dp.generate_observations(100.0, 5, 2))
    for i, r in enumerate(res):
        n, times, types, states = orc.simulate_events(desc, [0.003, 0.1, 0.01, 0.02], seed=5, p=i)
        assert [e.time for e in r.particle.trajectory] == list(times) and r.particle.final_condition == list(states[-1])


# ---- user observation models ------------------------------------------------------------------------------
# Laplace (double-exponential) observation error with a reporting fraction: log p = -|y - rho*I| / b - log(2 b),
# written with IEEE-exact operations only (theta[3] carries log(2b)), so host and device agree to the bit.
SOURCE_OBS = SOURCE + """
__device__ double dpomp_user_obs(const double* theta, const int* x, const long long* y, int n_y) {
    const double mean = 0.9 * (double)x[1];
    double d = (double)y[n_y - 1] - mean;
    if (d < 0.0) d = -d;
    return -d / 3.0 - 1.791759469228055;   // b = 3, log(6)
}
"""


def _obs_host(y, pop, theta):
    return -abs(y.val[-1] - 0.9 * pop[1]) / 3.0 - 1.791759469228055


def _model_obs(dp):
    rf = dp.RateSource(SOURCE_OBS, n_events=2, n_params=4, host=_host)
    return dp.generate_custom_model("SIS-laplace", rf, [100, 1], [[-1, 1], [1, -1]], obs_model=dp.SourceObsModel(_obs_host),
                                    prior=dp.generate_weak_prior(4))


def test_user_obs_model_compiles_and_runs_in_the_oracle(dp, orc, pooley, tmp_path):
    model = _model_obs(dp)
    assert "error" not in dp.check_rate_source(model, pooley).lower()
    desc = dp.ModelDescriptor(model, pooley)
    assert desc.c.obs.kind == 100
    orc.set_user_rates_from_source(SOURCE_OBS, str(tmp_path))
    for I in (0, 17, 65):
        assert orc.obs_loglik(desc, 1, [101 - I, I]) == _obs_host(pooley[1], [101 - I, I], None)
    with pytest.raises(TypeError):   # a user observation model needs the source to live somewhere
        dp.ModelDescriptor(dp.generate_custom_model("x", dp.RateFunction([dp.RateTerm(1, ((1,),))]), [1, 1], [[-1, 1]],
                                                    obs_model=dp.SourceObsModel()), pooley)
    orc.set_user_rates_from_source(SOURCE, str(tmp_path))   # leave the hooks as the other tests expect them


@pytest.mark.gpu
@pytest.mark.parametrize("P", [200, 3000])   # whole-filter kernel and multi-kernel path
def test_user_obs_model_filter_matches_oracle(dp, orc, pooley, tmp_path, P):
    orc.set_user_rates_from_source(SOURCE_OBS, str(tmp_path))
    pf = dp.ParticleFilter(_model_obs(dp), pooley)
    theta = np.array([[0.003, 0.0035], [0.1, 0.12], [0.01, 0.02], [0.02, 0.0]])
    pop = pf.alloc_population(2, P)
    ll = pf.partial_log_likelihood(pop, theta, 1, pf.n_obs, seed=19)
    for b in range(2):
        ref = orc.pf_loglik(pf.desc, P, theta[:, b], seed=19, seg=b)
        assert ll[b] == ref.ll and np.array_equal(pop.download(b), ref.pop)
    pf.close()
    orc.set_user_rates_from_source(SOURCE, str(tmp_path))


@pytest.mark.gpu
def test_rate_source_kernels_are_timed_when_profiling(dp, pooley):
    """Run-time compiled kernels go through the module-launch path; with profiling on they are launched with start/stop
    events (hipExtModuleLaunchKernel) and must give the same results and appear in the statistics."""
    pf = dp.ParticleFilter(_model(dp), pooley)
    theta = np.array([[0.003], [0.1], [0.01], [0.02]])
    plain = {P: pf.loglik(theta, P, seed=3)[0] for P in (200, 5000)}   # whole-filter kernel, multi-kernel path
    pf.set_profiling(True)
    timed = {P: pf.loglik(theta, P, seed=3)[0] for P in (200, 5000)}
    st = pf.kernel_stats()
    pf.set_profiling(False)
    assert timed == plain
    assert st["k_filter_small"][0] == 1 and st["k_filter_small"][1] > 0.0
    assert st["k_propagate"][0] == pf.n_obs and st["k_propagate"][1] > 0.0
    pf.close()


def test_code_object_cache_must_be_private_and_files_are_checksummed(dp, pooley, tmp_path):
    """The cache holds GPU code objects that dpomp_create loads: a directory other users can write to is not used, and
    a cache file whose code object was altered (same length, so only the checksum can tell) is recompiled and replaced.
    (Compile-only path: needs no GPU.)"""
    import os
    src = SOURCE.replace("theta[3]", "theta[3] * 1.0009765625")   # a text of this test's own: nothing cached in the process yet
    rf = dp.RateSource(src, n_events=2, n_params=4, host=_host)
    model = dp.generate_custom_model("SIS-sat-cache", rf, [100, 1], [[-1, 1], [1, -1]], prior=dp.generate_weak_prior(4))
    old = os.environ.get("DPOMP_CACHE_DIR")
    shared = tmp_path / "shared"
    shared.mkdir()
    os.chmod(shared, 0o777)
    private = tmp_path / "private"
    try:
        os.environ["DPOMP_CACHE_DIR"] = str(shared)
        dp.check_rate_source(model, pooley)
        assert os.listdir(shared) == []                      # refused: writable by others
        os.environ["DPOMP_CACHE_DIR"] = str(private)         # does not exist yet: created 0700
        src2 = src.replace("1.0009765625", "1.00048828125")
        model2 = dp.generate_custom_model("SIS-sat-cache2", dp.RateSource(src2, n_events=2, n_params=4, host=_host), [100, 1],
                                          [[-1, 1], [1, -1]], prior=dp.generate_weak_prior(4))
        dp.check_rate_source(model2, pooley)
        files = [f for f in os.listdir(private) if f.endswith(".dpomprtc")]
        assert len(files) == 1 and (os.stat(private).st_mode & 0o077) == 0
        # flip one byte of the code object: a fresh process (nothing cached in memory) must not accept the file -- it
        # compiles again and puts the good file back
        path = os.path.join(private, files[0])
        good = open(path, "rb").read()
        bad = bytearray(good)
        bad[len(bad) // 2] ^= 0x40
        open(path, "wb").write(bytes(bad))
        import subprocess
        import sys
        code = ("import sys; sys.path.insert(0, %r); import dpomp_amd as dp\n"
                "src = %r\n"
                "m = dp.generate_custom_model('SIS-sat-cache2', dp.RateSource(src, n_events=2, n_params=4), [100, 1], [[-1, 1], [1, -1]],"
                " prior=dp.generate_weak_prior(4))\n"
                "y = [dp.Observation(20.0, 1, 1.0, [0, 18]), dp.Observation(40.0, 1, 1.0, [0, 65])]\n"
                "dp.check_rate_source(m, y)\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), src2))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, DPOMP_CACHE_DIR=str(private)))
        assert r.returncode == 0, r.stderr[-2000:]
        replaced = open(path, "rb").read()
        assert replaced != bytes(bad) and replaced[:8] == good[:8]      # compiled again, written again
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, DPOMP_CACHE_DIR=str(private)))
        assert r.returncode == 0 and open(path, "rb").read() == replaced  # ... and that file is accepted as it is
    finally:
        if old is None:
            os.environ.pop("DPOMP_CACHE_DIR", None)
        else:
            os.environ["DPOMP_CACHE_DIR"] = old
