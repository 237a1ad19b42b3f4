"""SMC^2 host driver (run_pibis, src/hmm_ibis.jl:12-135).  The reference pins it only through a seed-1
value of -ln evidence (19.98, test/runtests.jl:48-52); the exact value under that prior is 20.1676."""
import json
import os

import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EXACT = json.load(open(os.path.join(GOLDEN, "pf_golden.json")))["exact_pooley_sis_evidence_u001_u05"]


from helpers import OracleBank, OraclePop  # noqa: E402,F401


def test_theta_level_helpers(dp):
    rng = np.random.default_rng(0)
    w = np.array([0.0, 3.0, 0.0, 1.0])
    idx = dp.rs_systematic(w, rng)
    assert list(idx) in ([1, 1, 1, 3], [1, 1, 3, 3]) and np.all(np.diff(idx) >= 0)
    counts = np.zeros(4)
    for _ in range(4000):
        counts += np.bincount(dp.rs_systematic(w, rng), minlength=4)
    assert np.allclose(counts / counts.sum(), w / w.sum(), atol=0.01)
    theta = np.array([[1.0, 2.0, 4.0], [0.0, 1.0, 0.0]])
    mu, cv = dp.compute_is_mu_covar(theta, np.array([1.0, 1.0, 2.0]))
    assert np.allclose(mu, [2.75, 0.25]) and np.isclose(cv[0, 0], (1.75 ** 2 + 0.75 ** 2 + 2 * 1.25 ** 2) / 4)
    assert np.isclose(cv[0, 1], cv[1, 0]) and dp.compute_ess(np.ones(10)) == 10.0


REFERENCE_SEED1_BME = 19.98  # test/runtests.jl:51 -- the reference's own pinned value of -ln evidence


def test_run_pibis_reproduces_the_reference_value(dp, orc, pooley):
    """Algorithm as written in the reference (independent proposals, no proposal-density correction):
    -ln evidence lands on the reference's pinned 19.98, not on the exact 20.17 -- the bias is the reference's."""
    model = dp.generate_model("SIS", [100, 1])
    prior = dp.UniformProductPrior([0.0, 0.0], [0.01, 0.5])
    vals = []
    for s in range(3):
        rng = np.random.default_rng(40 + s)
        out = dp.run_pibis(OracleBank(dp, orc, model, pooley), prior, prior.rand(rng, 1000), 0.3, True, 1.002, 100, seed=s,
                           rng=rng, verbose=False)
        assert out.n_resample >= 1 and 0.3 < out.acceptance_rate < 0.9 and out.theta.shape == (2, 1000)
        vals.append(out.bme[0])
    assert abs(np.mean(vals) - REFERENCE_SEED1_BME) < 0.15, vals   # sd over seeds ~0.09 at this size
    assert np.mean(vals) < EXACT["neg_log_evidence"] - 0.05         # and visibly below the exact value


@pytest.mark.parametrize("ind_prop", [True, False])
def test_run_pibis_valid_kernels_reach_the_exact_evidence(dp, orc, pooley, ind_prop):
    """With a valid PMMH kernel (independent proposals WITH the q-ratio, or symmetric random-walk proposals)
    the same driver converges on the exact evidence and posterior mean."""
    model = dp.generate_model("SIS", [100, 1])
    prior = dp.UniformProductPrior([0.0, 0.0], [0.01, 0.5])
    bme, beta = [], []
    for s in range(3):
        rng = np.random.default_rng(70 + s)
        out = dp.run_pibis(OracleBank(dp, orc, model, pooley), prior, prior.rand(rng, 1000), 0.3, ind_prop, 1.002, 100,
                           seed=s, rng=rng, verbose=False, proposal_correction=True)
        bme.append(out.bme[0])
        beta.append(out.mu[0])
    assert abs(np.mean(bme) - EXACT["neg_log_evidence"]) < 0.2, bme
    assert abs(np.mean(beta) - EXACT["beta_mean"]) < 0.06 * EXACT["beta_mean"], beta


def test_exact_evidence_quadrature_is_stable():
    from oracle import exact_sis
    coarse = exact_sis.exact_evidence((0, 0), (0.01, 0.5), n=20)
    assert abs(coarse["neg_log_evidence"] - EXACT["neg_log_evidence"]) < 0.05
    assert abs(EXACT["neg_log_evidence"] - 20.1676) < 1e-3 and abs(EXACT["beta_mean"] - 0.003246) < 1e-6


@pytest.mark.gpu
def test_smc2_reference_configuration_on_gpu(dp, pooley):
    """The reference's own smc2 testset (test/runtests.jl:48-52): 4000 theta-particles x 200 state particles,
    prior U(0,0.01) x U(0,0.5).  Reference seed-1 value 19.98; exact 20.1676."""
    model = dp.generate_model("SIS", [100, 1])
    model.prior = dp.UniformProductPrior([0.0, 0.0], [0.01, 0.5])
    # Measured over 20 seeds on MI355X (tools/smc2_spread.py): as written 20.07 +- 0.07 (sd over seeds), corrected
    # 20.177 +- 0.064.  16 seeds each: standard error of a mean ~0.017.
    runs = [dp.run_ibis_analysis(model, pooley, seed=s, verbose=False) for s in range(1, 17)]
    bme = np.array([o.bme[0] for o in runs])
    assert runs[0].theta.shape == (2, 4000)
    assert abs(bme.mean() - REFERENCE_SEED1_BME) < 0.2, bme     # reference: bme[1] ~ 19.98 (Julia index 1), one seed
    assert bme.min() < REFERENCE_SEED1_BME + 0.07                # 19.98 lies inside our seed-to-seed scatter
    fixed = [dp.run_ibis_analysis(model, pooley, seed=s, verbose=False, proposal_correction=True) for s in range(1, 17)]
    bme_fixed = np.array([o.bme[0] for o in fixed])
    # the driver as the reference wrote it carries the reference's bias: visibly BELOW the exact evidence, as its own
    # pinned 19.98 is, and below the corrected driver, which lands on the exact value (margins ~2.5 standard errors)
    assert bme.mean() < EXACT["neg_log_evidence"] - 0.04, (bme.mean(), EXACT["neg_log_evidence"])
    assert bme_fixed.mean() - bme.mean() > 0.04, (bme_fixed.mean(), bme.mean())
    assert abs(bme_fixed.mean() - EXACT["neg_log_evidence"]) < 0.06
    assert abs(np.mean([o.mu[0] for o in fixed]) - EXACT["beta_mean"]) < 0.02 * EXACT["beta_mean"]
    assert abs(np.mean([o.mu[1] for o in fixed]) - EXACT["gamma_mean"]) < 0.03 * EXACT["gamma_mean"]


@pytest.mark.gpu
def test_smc2_on_seir_lands_on_the_exact_evidence(dp):
    """SMC^2 on a model other than Pooley SIS, against a known answer: the exact evidence and posterior mean of the
    synthetic SEIR data under a uniform prior box (tests/golden/seir_evidence.json: forward algorithm over 182 002
    states x 12^3 Gauss-Legendre nodes, quadrature error ~0.006).  With a valid PMMH kernel (independent proposals WITH
    the q-ratio) 2048 theta-particles x 4096 state particles land on both within 3 standard errors over 8 seeds; the
    driver as the reference wrote it (no q-ratio, src/hmm_ibis.jl:104) lands measurably elsewhere in the posterior mean.
    Measured over 12 seeds (tools/smc2_seir_spread.py): corrected 14.419 +- 0.038 (sd over seeds; exact 14.397), recovery
    rate 0.10082 +- 0.00032 (exact 0.10087); as written 14.352 +- 0.039 and 0.09970 +- 0.00040."""
    exact = json.load(open(os.path.join(GOLDEN, "seir_evidence.json")))
    assert exact["quadrature_error_estimate"] < 0.01
    y = dp.get_observations(os.path.join(GOLDEN, "seir_synthetic.csv"))
    model = dp.generate_model("SEIR", [100, 0, 1, 0])
    model.prior = dp.UniformProductPrior(exact["prior"]["lower"], exact["prior"]["upper"])
    n = 8
    out = {}
    for corr in (True, False):
        runs = [dp.run_smc2_analysis(model, y, np_=2048, npf=4096, seed=s, verbose=False, proposal_correction=corr) for s in range(1, n + 1)]
        assert all(r.n_resample >= 1 and r.status == 0 for r in runs)
        out[corr] = (np.array([r.bme[0] for r in runs]), np.array([r.mu for r in runs]))
    bme, mu = out[True]
    se = lambda v, floor: max(float(np.std(v, ddof=1)), floor) / np.sqrt(n)   # (floors: the spread measured over 12 seeds)
    assert abs(bme.mean() - exact["neg_log_evidence"]) < 3 * se(bme, 0.038) + exact["quadrature_error_estimate"], (bme, exact["neg_log_evidence"])
    for k, floor in enumerate((1.0e-4, 6.0e-3, 3.2e-4)):
        assert abs(mu[:, k].mean() - exact["posterior_mean"][k]) < 3 * se(mu[:, k], floor), (k, mu[:, k].mean(), exact["posterior_mean"][k])
    mu_ref = out[False][1]
    # without the q-ratio the sample is pulled towards the proposal mean: the recovery rate sits ~0.0012 too low (10 se)
    assert exact["posterior_mean"][2] - mu_ref[:, 2].mean() > 6.0e-4
    assert abs(mu[:, 2].mean() - exact["posterior_mean"][2]) < abs(mu_ref[:, 2].mean() - exact["posterior_mean"][2])


@pytest.mark.gpu
def test_smc2_gpu_equals_oracle_bank_at_same_seeds(dp, orc, pooley):
    """Same driver, same seeds: the GPU bank and the oracle bank must walk the identical path."""
    model = dp.generate_model("SIS", [100, 1])
    prior = dp.UniformProductPrior([0.0, 0.0], [0.01, 0.5])
    theta0 = prior.rand(np.random.default_rng(5), 64)
    a = dp.run_pibis(OracleBank(dp, orc, model, pooley), prior, theta0, 0.3, True, 1.002, 128, seed=9,
                     rng=np.random.default_rng(11), verbose=False)
    pf = dp.ParticleFilter(model, pooley)
    b = dp.run_pibis(pf, prior, theta0, 0.3, True, 1.002, 128, seed=9, rng=np.random.default_rng(11), verbose=False)
    assert np.array_equal(a.bme, b.bme) and np.array_equal(a.theta, b.theta) and np.array_equal(a.weight, b.weight)
    pf.close()


@pytest.mark.gpu
def test_sharded_bank_on_gpu_matches_local_bank(dp, pooley):
    """world_size-1 nccl group on the one GPU of the box: exercises the zero-copy torch view of the
    device-resident populations and the RCCL all-gather of states used for theta-level migration."""
    import torch
    import torch.distributed as dist
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
    sock.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        model = dp.generate_model("SIS", [100, 1])
        prior = dp.UniformProductPrior([0.0, 0.0], [0.01, 0.5])
        theta0 = prior.rand(np.random.default_rng(5), 96)
        pf = dp.ParticleFilter(model, pooley)
        a = dp.run_pibis(pf, prior, theta0, 0.3, True, 1.002, 256, seed=9, rng=np.random.default_rng(11), verbose=False)
        b = dp.run_pibis(pf, prior, theta0, 0.3, True, 1.002, 256, seed=9, rng=np.random.default_rng(11), verbose=False,
                         distributed=True)
        assert a.n_resample >= 1
        assert np.array_equal(a.bme, b.bme) and np.array_equal(a.theta, b.theta) and np.array_equal(a.weight, b.weight)
        assert b.migration["rows_received"] == 0   # one rank: every ancestor is local, the all-to-all carries nothing
        # loopback: the rank sends the populations its block descends from TO ITSELF through the collective, so
        # all_to_all_single runs on the GPU over RCCL with non-zero split sizes on device tensors
        c = dp.run_pibis(pf, prior, theta0, 0.3, True, 1.002, 256, seed=9, rng=np.random.default_rng(11), verbose=False,
                         distributed="loopback")
        assert c.migration["rows_received"] > 0
        assert np.array_equal(a.bme, c.bme) and np.array_equal(a.theta, c.theta) and np.array_equal(a.weight, c.weight)
        # the view really aliases library memory
        pop = pf.alloc_population(2, 300)
        pf.partial_log_likelihood(pop, theta0[:, :2], 1, 2, seed=3)
        view = pop.state_tensor()
        assert tuple(view.shape) == (2, 2, 1024) and view.dtype == torch.int32
        assert np.array_equal(view[1, :, :300].cpu().numpy().T, pop.download(1))
        pf.close()
    finally:
        dist.destroy_process_group()


def _gpu_smc2_worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dpomp_amd as dp
    y = dp.get_observations(os.path.join(root, "tests", "golden", "pooley.csv"))
    model = dp.generate_model("SIS", [100, 1])
    prior = dp.UniformProductPrior([0.0, 0.0], [0.01, 0.5])
    theta0 = prior.rand(np.random.default_rng(5), 101)
    pf = dp.ParticleFilter(model, y, device=0)   # both ranks share the one GPU of the box
    out = dp.run_pibis(pf, prior, theta0, 0.3, True, 1.002, 300, seed=9, rng=np.random.default_rng(11), verbose=False,
                       distributed=True)
    np.savez(os.path.join(out_dir, f"gpu_smc2_rank{rank}.npz"), bme=out.bme, theta=out.theta, weight=out.weight)
    pf.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_process_sharded_smc2_on_gpu_filters(dp, pooley, tmp_path):
    """Two ranks (gloo rendezvous), each running its block of theta-particles as device-resident filters on the GPU,
    populations migrating between the processes at theta-level resampling: identical to the single-process run."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_gpu_smc2_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    model = dp.generate_model("SIS", [100, 1])
    prior = dp.UniformProductPrior([0.0, 0.0], [0.01, 0.5])
    theta0 = prior.rand(np.random.default_rng(5), 101)
    pf = dp.ParticleFilter(model, pooley)
    ref = dp.run_pibis(pf, prior, theta0, 0.3, True, 1.002, 300, seed=9, rng=np.random.default_rng(11), verbose=False)
    pf.close()
    assert ref.n_resample >= 1
    for r in range(2):
        got = np.load(tmp_path / f"gpu_smc2_rank{r}.npz")
        assert np.array_equal(got["bme"], ref.bme) and np.array_equal(got["theta"], ref.theta) and np.array_equal(got["weight"], ref.weight)
