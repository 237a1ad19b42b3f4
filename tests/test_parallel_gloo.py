"""The N > 1 path on CPU: world_size-2 gloo process group, segments sharded over ranks, one
all-gather of the log-likelihood vector.  The per-rank evaluator is the ORACLE here (tests may use
it; on a GPU box the evaluator is ParticleFilter.loglik) -- what is under test is the sharding,
the global segment index in the RNG key, and the collective."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dpomp_amd as dp
    from dpomp_amd import parallel
    from oracle import oracle as orc
    y = dp.get_observations(os.path.join(ROOT, "tests", "golden", "pooley.csv"))
    desc = dp.ModelDescriptor(dp.generate_model("SIS", [100, 1]), y)
    theta = np.stack([0.0025 + 0.0002 * np.arange(B), 0.08 + 0.01 * np.arange(B)])

    def evaluate(th, base):
        return np.array([orc.pf_loglik(desc, 256, th[:, j], seed=77, seg=base + j).ll for j in range(th.shape[1])])

    full = parallel.sharded_loglik(evaluate, theta)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), full)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 2, 1])
def test_sharded_batch_equals_single_process(tmp_path, dp, orc, B):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), B, str(tmp_path)), nprocs=world, join=True)
    y = dp.get_observations(os.path.join(ROOT, "tests", "golden", "pooley.csv"))
    desc = dp.ModelDescriptor(dp.generate_model("SIS", [100, 1]), y)
    theta = np.stack([0.0025 + 0.0002 * np.arange(B), 0.08 + 0.01 * np.arange(B)])
    want = np.array([orc.pf_loglik(desc, 256, theta[:, b], seed=77, seg=b).ll for b in range(B)])
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert np.array_equal(got, want)  # every rank holds the full, partition-invariant vector


def test_shard_bounds_partition():
    from dpomp_amd import parallel
    for B in (1, 2, 5, 8, 512, 513):
        for world in (1, 2, 3, 8):
            blocks = [parallel.shard_bounds(B, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            assert max(h - l for l, h in blocks) - min(h - l for l, h in blocks) <= 1
            for b in range(B):
                r = parallel.shard_of(b, B, world)
                assert blocks[r][0] <= b < blocks[r][1]


def test_single_process_is_identity(dp):
    from dpomp_amd import parallel
    th = np.array([[0.003, 0.004], [0.1, 0.2]])
    out = parallel.sharded_loglik(lambda t, base: t[0] * 1000 + base, th)
    assert np.array_equal(out, [3.0 + 0, 4.0 + 0])


def test_migration_plan_moves_each_population_once_per_destination():
    """The plan of the theta-level migration, executed on plain arrays: equals nidx-gathering the whole bank, while
    no rank receives more rows than its block has slots."""
    from dpomp_amd import parallel
    rng = np.random.default_rng(3)
    for N, world in ((51, 2), (50, 3), (8, 8), (5, 8), (512, 8)):
        bank = np.arange(N) * 10  # population b is the number 10 b
        for trial in range(4):
            w = rng.random(N) ** (1 + 3 * trial)
            nidx = np.sort(rng.choice(N, size=N, p=w / w.sum()))  # systematic resampling gives sorted ancestors
            if trial == 3:
                nidx = rng.permutation(nidx)                       # but nothing relies on it
            plans = [parallel.migration_plan(nidx, N, world, r) for r in range(world)]
            bounds = [parallel.shard_bounds(N, world, r) for r in range(world)]
            for r, (lo, hi) in enumerate(bounds):
                p = plans[r]
                assert p["recv_counts"].sum() <= hi - lo and p["recv_counts"][r] == 0 and p["send_counts"][r] == 0
                # what arrives: every other rank's send list for r, in rank order
                recv = []
                for q in range(world):
                    if q != r:
                        off = plans[q]["send_counts"][:r].sum()
                        rows = plans[q]["send_rows"][off:off + plans[q]["send_counts"][r]]
                        assert len(rows) == p["recv_counts"][q] and len(np.unique(rows)) == len(rows)
                        recv.extend(bank[bounds[q][0] + rows])
                recv = np.array(recv, dtype=np.int64)
                new = bank[lo:hi][p["local_anc"]]
                new[p["remote_slots"]] = recv[p["remote_pick"]]
                assert np.array_equal(new, bank[nidx[lo:hi]])


def _smc2_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dpomp_amd as dp
    from helpers import OracleBank
    from oracle import oracle as orc
    y = dp.get_observations(os.path.join(ROOT, "tests", "golden", "pooley.csv"))
    model = dp.generate_model("SIS", [100, 1])
    prior = dp.UniformProductPrior([0.0, 0.0], [0.01, 0.5])
    theta0 = prior.rand(np.random.default_rng(5), 51)  # odd count: blocks of 26 and 25
    out = dp.run_pibis(OracleBank(dp, orc, model, y), prior, theta0, 0.3, True, 1.002, 64, seed=9,
                       rng=np.random.default_rng(11), verbose=False, distributed=True)
    np.savez(os.path.join(out_dir, f"smc2_rank{rank}.npz"), bme=out.bme, theta=out.theta, weight=out.weight, nres=out.n_resample,
             rows_received=out.migration["rows_received"], block=out.migration["block"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_smc2_equals_single_process(tmp_path, dp, orc, world):
    """SMC^2 with the theta-particles sharded over 2 / 3 ranks (populations migrate at theta-level resampling, each
    rank receiving at most its block's worth per resampling) walks exactly the path of the single-process run."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import OracleBank
    mp.spawn(_smc2_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    y = dp.get_observations(os.path.join(ROOT, "tests", "golden", "pooley.csv"))
    model = dp.generate_model("SIS", [100, 1])
    prior = dp.UniformProductPrior([0.0, 0.0], [0.01, 0.5])
    theta0 = prior.rand(np.random.default_rng(5), 51)
    ref = dp.run_pibis(OracleBank(dp, orc, model, y), prior, theta0, 0.3, True, 1.002, 64, seed=9,
                       rng=np.random.default_rng(11), verbose=False)
    assert ref.n_resample >= 1
    for r in range(world):
        got = np.load(tmp_path / f"smc2_rank{r}.npz")
        assert np.array_equal(got["bme"], ref.bme) and np.array_equal(got["theta"], ref.theta) and np.array_equal(got["weight"], ref.weight)
        assert 0 < got["rows_received"] <= ref.n_resample * got["block"]  # never the whole bank (51 per resampling)


def _pmmh_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dpomp_amd as dp
    from dpomp_amd import parallel
    from oracle import oracle as orc
    y = dp.get_observations(os.path.join(ROOT, "tests", "golden", "pooley.csv"))
    desc = dp.ModelDescriptor(dp.generate_model("SIS", [100, 1]), y)
    prior = dp.UniformProductPrior([0.0, 0.0], [0.01, 0.5])

    def evaluate(th, seed):  # chains sharded over ranks, log-likelihoods all-gathered
        return parallel.sharded_loglik(
            lambda t, base: np.array([orc.pf_loglik(desc, 64, t[:, j], seed=seed, seg=base + j).ll for j in range(t.shape[1])]), th)

    init = np.array([[0.003, 0.0035, 0.0028], [0.1, 0.12, 0.09]])  # 3 chains on 2 ranks
    out = dp.run_pmcmc(evaluate, prior, init, steps=60, adapt_period=30, seed=5, verbose=False)
    np.save(os.path.join(out_dir, f"pmmh_rank{rank}.npy"), out.samples.theta)
    dist.barrier()
    dist.destroy_process_group()


def test_distributed_pmmh_equals_single_process(tmp_path, dp, orc):
    mp.spawn(_pmmh_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    y = dp.get_observations(os.path.join(ROOT, "tests", "golden", "pooley.csv"))
    desc = dp.ModelDescriptor(dp.generate_model("SIS", [100, 1]), y)
    prior = dp.UniformProductPrior([0.0, 0.0], [0.01, 0.5])
    init = np.array([[0.003, 0.0035, 0.0028], [0.1, 0.12, 0.09]])
    ref = dp.run_pmcmc(lambda th, seed: np.array([orc.pf_loglik(desc, 64, th[:, c], seed=seed, seg=c).ll for c in range(th.shape[1])]),
                       prior, init, steps=60, adapt_period=30, seed=5, verbose=False)
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"pmmh_rank{r}.npy"), ref.samples.theta)


def _bench_leg_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import dpomp_amd as dp
    from helpers import OracleBank
    from oracle import oracle as orc
    leg = bench.smc2_config4(dp, lambda model, y: OracleBank(dp, orc, model, y), 24, 48, distributed=True, reps=1)
    with open(os.path.join(out_dir, f"leg_rank{rank}.json"), "w") as f:
        json.dump(leg, f)
    dist.barrier()
    dist.destroy_process_group()


def test_bench_smc2_leg_gives_the_same_evidence_for_every_number_of_ranks(tmp_path, dp, orc):
    """bench.py's SMC^2 leg (BASELINE configuration 4: SEIR, theta-particles sharded over the ranks) at small size with
    the oracle as the filter bank: two gloo ranks reproduce the single-process evidence bit for bit, and populations
    really travel between them."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench
    from helpers import OracleBank
    mp.spawn(_bench_leg_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    ref = bench.smc2_config4(dp, lambda model, y: OracleBank(dp, orc, model, y), 24, 48, distributed=False, reps=1)
    assert ref["n_resample"] >= 1 and np.all(np.isfinite(ref["minus_ln_evidence"]))
    received = 0
    for r in range(2):
        got = json.load(open(tmp_path / f"leg_rank{r}.json"))
        assert got["minus_ln_evidence"] == ref["minus_ln_evidence"] and got["posterior_mean"] == ref["posterior_mean"]
        assert got["n_resample"] == ref["n_resample"] and got["block"] == 12
        received += got["rows_received"]
    assert received > 0


def test_migration_plan_loopback_sends_local_ancestors_through_the_collective():
    """The test hook behind the nccl world-1 GPU test: with loopback a rank's own populations also travel through the
    all-to-all (send to self), and the plan still reconstructs pop[nidx]."""
    from dpomp_amd.parallel import migration_plan
    rng = np.random.default_rng(3)
    N = 23
    nidx = np.sort(rng.integers(0, N, N))
    bank = np.arange(N) * 10
    plan = migration_plan(nidx, N, 1, 0, loopback=True)
    assert plan["send_counts"].tolist() == [len(np.unique(nidx))] == plan["recv_counts"].tolist()
    assert len(plan["remote_slots"]) == N and np.array_equal(plan["local_anc"], np.arange(N))
    recv = bank[plan["send_rows"]]
    new = bank[plan["local_anc"]].copy()
    new[plan["remote_slots"]] = recv[plan["remote_pick"]]
    assert np.array_equal(new, bank[nidx])
    for world in (2, 3):   # and with several ranks: every rank's block, self included, arrives through the buffers
        from dpomp_amd.parallel import shard_bounds
        plans = [migration_plan(nidx, N, world, r, loopback=True) for r in range(world)]
        for r, p in enumerate(plans):
            lo, hi = shard_bounds(N, world, r)
            recv = []
            for q in range(world):
                qlo = shard_bounds(N, world, q)[0]
                off = plans[q]["send_counts"][:r].sum()
                rows = plans[q]["send_rows"][off:off + plans[q]["send_counts"][r]]
                assert len(rows) == p["recv_counts"][q]
                recv.extend(bank[qlo + rows])
            recv = np.array(recv, dtype=np.int64)
            new = bank[lo:hi][p["local_anc"]].copy()
            new[p["remote_slots"]] = recv[p["remote_pick"]]
            assert np.array_equal(new, bank[nidx[lo:hi]])
