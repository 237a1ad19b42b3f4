#!/usr/bin/env python3
"""Benchmark of the particle-filter hot path: particle-steps/sec on Pooley SIS at 2^20 particles,
plus the log-ML error against the CPU reference path.

One "step" = one full bootstrap particle filter call over the T = 5 Pooley observations with
P = 2^20 particles per GPU (propagate -> weight -> log-mean-weight -> systematic resample, x5).
N GPUs run N independent filters (one per rank, weak scaling) and all-gather their log-ML
estimates over RCCL after every step -- independent pMCMC chains / theta-particles shard across
GPUs and exchange nothing but these estimates.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

POOLEY = [(20.0, [0, 18]), (40.0, [0, 65]), (60.0, [0, 70]), (80.0, [0, 66]), (100.0, [0, 67])]  # data/pooley.csv
THETA = [0.003, 0.1]
EXACT_LL = -15.6873794556  # exact CTMC forward algorithm (oracle/exact_sis.py)
HBM_PEAK_GBS = 8000.0      # MI355X: 8.0 TB/s HBM3E spec
HBM_MEASURED_COPY_GBS = 6290.0  # measured float4-copy rate of the chip (79 % of spec): the practical roof
# VALU-issue ceiling of k_propagate: the SIS event loop is 66 VALU instructions per event (ISA of
# k_propagate_c<RSir<2>,32>: 2 x (10 stream step + 3 uniform), 17 log, 8 division + 2 range test, 6 rates, 5 event choice
# and state update, 2 clock).  Measured issue cost per wave64 instruction on gfx950 (tools/valu_rate_probe.hip ->
# profiles/r01_final/valu_issue_rates.json): FP64, 64-bit integer and three-operand 32-bit ops 4 cycles, plain
# two-operand 32-bit ops (xor/add/sub/or/and/shift: 15 of the 66) ~0.65 of that, v_rcp_f64 3.4 x.  Hence
# 50 + 15 * 0.65 + 3.4 = 63.15 issue slots = 253 cycles per 64 events per SIMD; 1024 SIMDs at 2.4 GHz.  The count is
# taken from the ISA of the build by tools/isa_loop_count.py (-> profiles/*/valu_ceilings.json, read below; the
# constants here are the fallback when no such file is present).
VALU_CYCLES_PER_WAVE_EVENT = (50 + 15 * 0.65 + 3.4) * 4
VALU_CEILING_EVENTS_PER_S = 1024 * 2.4e9 * 64 / VALU_CYCLES_PER_WAVE_EVENT


def valu_ceiling(family="sis"):
    """(events/s ceiling of the compaction kernel's event loop, where it came from) -- latest profiles/*/valu_ceilings.json."""
    root = os.path.join(ROOT, "profiles")
    best = (VALU_CEILING_EVENTS_PER_S, "bench.py constant")
    for d in sorted(os.listdir(root)) if os.path.isdir(root) else []:
        f = os.path.join(root, d, "valu_ceilings.json")
        if os.path.exists(f):
            for label, v in json.load(open(f))["kernels"].items():
                if label.startswith(family + " ") and "k_propagate_c" in label and "ceiling_events_per_s" in v:
                    best = (float(v["ceiling_events_per_s"]), f"profiles/{d}/valu_ceilings.json: {v['valu']} VALU instructions "
                            f"({v['valu_light']} light, {v['valu_trans']} transcendental) = {v['issue_slots_per_event']} issue slots per event")
    return best


def hbm_scale_summary():
    """The streaming kernels where the working set exceeds the 256 MiB Infinity Cache, from the latest committed
    rocprofv3 passes (profiles/*/hbm_scale/<configuration>/by_launch_shape.json, one set of passes per configuration;
    16.8 M particles per launch): bytes really moved / average duration, as a fraction of 8 TB/s, per configuration."""
    root = os.path.join(ROOT, "profiles")
    n = 16384 * 1024  # particles per launch of both configurations
    # bytes moved per particle: k_wsum reads the log-weight and writes the fixed-point weight; k_offsets reads that and
    # writes the int32 boundary; k_gather (only the incremental SEIR pattern runs it) reads the boundary and moves 4 int32
    configs = {"sis_16x2p20": {"k_wsum": 16, "k_offsets<true>": 12},
               "seir_64x2p18": {"k_wsum": 16, "k_offsets<true>": 12},
               "seir_64x2p18_incremental": {"k_gather": 4 + 8 * 4}}
    out = None
    for d in sorted(os.listdir(root)) if os.path.isdir(root) else []:
        base = os.path.join(root, d, "hbm_scale")
        cur = {"source": f"profiles/{d}/hbm_scale/<configuration>/by_launch_shape.json", "particles_per_launch": n,
               "working_set": "576 MiB (SIS, 16 filters x 2^20) and 832 MiB (SEIR, 64 x 2^18)"}
        for cfg, kernels in configs.items():
            f = os.path.join(base, cfg, "by_launch_shape.json")
            if not os.path.exists(f):
                continue
            shapes = json.load(open(f))
            for kname, per_particle in kernels.items():
                for grid, v in shapes.get("dpomp::" + kname, {}).items():
                    nbytes = per_particle * n
                    cur.setdefault(kname.split("<")[0], {})[cfg] = {
                        "grid": grid, "avg_us": v["avg_us"], "moved_bytes": nbytes, "pmc_hbm_bytes": v.get("hbm_bytes_per_launch_corrected"),
                        "frac_of_8TBs": round(nbytes / (v["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 3)}
        if len(cur) > 3:
            out = cur
    return out


SMC2_PRIOR_BOX = ([0.0, 0.0, 0.0], [0.02, 0.6, 0.4])  # tightened from the reference's weak prior U(0,1)^3 (hmm_examples.jl:33-35)


def smc2_config4(dp, bank_factory, n_theta, npf, distributed, seed=3, reps=2, data=None):
    """BASELINE configuration 4: SMC^2 (run_pibis, src/hmm_ibis.jl:12-135) on SEIR with n_theta theta-particles x npf
    state particles, the theta-particles sharded over the ranks when `distributed`; at theta-level resampling
    (:70-75) populations whose ancestor lives on another rank travel in one all-to-all.  Runs `reps` times with the
    same seed (the first pays allocations and the clock ramp) and reports the last: the RNG is keyed by global indices,
    so the evidence is the same bits for every number of ranks."""
    import numpy as np
    y = data or dp.get_observations(os.path.join(ROOT, "tests", "golden", "seir_synthetic.csv"))
    model = dp.generate_model("SEIR", [100, 0, 1, 0])
    prior = dp.UniformProductPrior(*SMC2_PRIOR_BOX)
    out = None
    for rep in range(reps):
        rng = np.random.default_rng(seed)
        theta0 = prior.rand(rng, n_theta)
        bank = bank_factory(model, y)
        t0 = time.perf_counter()
        try:
            res = dp.run_pibis(bank, prior, theta0, 0.3, True, 1.002, npf, seed=seed, rng=rng, verbose=False, distributed=distributed)
        finally:
            close = getattr(bank, "close", None)
            if close:
                close()
        out = {"seconds": time.perf_counter() - t0, "n_resample": int(res.n_resample),
               "minus_ln_evidence": [float(v) for v in np.atleast_1d(res.bme)],
               "posterior_mean": [float(v) for v in res.mu], "acceptance_rate": float(res.acceptance_rate), "status": int(res.status),
               "rows_received": int(res.migration["rows_received"]) if res.migration else 0,
               "block": int(res.migration["block"]) if res.migration else n_theta,
               "theta_particles": n_theta, "state_particles": npf}
    return out


def algorithmic_bytes(P, S):
    """Bytes per launch with Int64 state as the reference holds it (w = 8): the accounting the metric is defined on."""
    return {
        "k_propagate": (2 * S + 1) * 8 * P,   # read state, write state, write log-weight
        "k_wsum": 8 * P,                      # normalise: read weight
        "k_scan_blocks": 0,                   # tile sums only (final observation; fused into k_offsets otherwise)
        "k_offsets": 16 * P,                  # scan: read 8 + write 8
        "k_gather": (2 * S + 1) * 8 * P,      # resample: read cum. weight + gather state + write state
    }


def moved_bytes(P, S):
    """Bytes per launch the kernels really have to move: the state lives in HBM as int32, the offspring boundaries as
    int32, log-weights as f64, fixed-point weights as u64 (written by k_wsum, read by k_offsets).  (The resampling
    gather of a full filter call runs inside the next k_propagate.)"""
    return {
        "k_propagate": (4 * S + 4 * S + 8) * P,
        "k_wsum": (8 + 8) * P,
        "k_scan_blocks": 0,
        "k_offsets": (8 + 4) * P,
        "k_gather": (4 + 4 * S + 4 * S) * P,
    }


def pmc_traffic(kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/README.md: FETCH_SIZE and
    WRITE_SIZE in separate runs, FETCH_SIZE doubled per MI355X_MICROARCH.md).  None if no profile is present."""
    best = None
    root = os.path.join(ROOT, "profiles")
    for d in sorted(os.listdir(root)) if os.path.isdir(root) else []:  # (the latest committed bench passes win)
        for sub in ("", "bench"):
            f = os.path.join(root, d, sub, "pmc_hbm_traffic.json")
            if os.path.exists(f):
                for name, v in json.load(open(f)).items():
                    base = name.replace("void ", "")
                    base = base[len("dpomp::"):] if base.startswith("dpomp::") else base
                    if base.split("<")[0] == kernel:
                        best = (v.get("hbm_bytes_per_launch_corrected"), os.path.join(d, sub).rstrip("/"))
    return best


def cpu_baseline(dp, P_serial, P_parallel):
    """The oracle (C restatement of the reference algorithm) timed on this host: serial, as the
    reference runs, and with all host cores (OpenMP over particles)."""
    from oracle import oracle as orc
    y = [dp.Observation(t, 1, 1.0, v) for t, v in POOLEY]
    desc = dp.ModelDescriptor(dp.generate_model("SIS", [100, 1]), y)
    T = len(y)
    t0 = time.perf_counter()
    r1 = orc.pf_loglik(desc, P_serial, THETA, seed=1, nthreads=1)
    t_serial = time.perf_counter() - t0
    # Every core this job may use: the scheduler affinity of the process, capped by what the OpenMP runtime was told
    # (OMP_NUM_THREADS, which a shared GPU box sets to its per-GPU CPU share).  All three figures go into the line.
    host_cpus = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = host_cpus
    omp_max = orc.max_threads()
    cores = max(1, min(affinity, omp_max))
    if os.environ.get("DPOMP_CPU_THREADS"):
        cores = max(1, int(os.environ["DPOMP_CPU_THREADS"]))
    orc.pf_loglik(desc, 4096, THETA, seed=2, nthreads=cores)  # spin the thread pool up
    t0 = time.perf_counter()
    rp = orc.pf_loglik(desc, P_parallel, THETA, seed=1, nthreads=cores)
    t_par = time.perf_counter() - t0
    return {
        "ll": rp.ll,
        "value": P_parallel * T / t_par,
        "unit": "particle-steps/s",
        "cores": cores,
        "host_cpu_count": host_cpus,
        "affinity_cpus": affinity,
        "omp_max_threads": omp_max,
        "kind": "port",
        "sample": f"oracle C restatement (reference is Julia, absent here); full Pooley SIS PF, "
                  f"{P_parallel} particles on {cores} OpenMP threads = min(affinity {affinity}, OMP max threads {omp_max}) "
                  f"of a host with {host_cpus} logical CPUs ({t_par:.2f} s); serial run "
                  f"{P_serial} particles ({t_serial:.2f} s)",
        "serial_value": P_serial * T / t_serial,
        "serial_events_per_s": r1.events / t_serial,
        "events_per_s": rp.events / t_par,
        "julia_on_path": bool(any(os.access(os.path.join(p, "julia"), os.X_OK) for p in os.environ.get("PATH", "").split(os.pathsep))),
        "note": "the restatement does not heap-allocate per event as the Julia code does "
                "(src/hmm_examples.jl:12-14): it is faster than the real reference, so GPU/CPU ratios are conservative",
    }


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args, argv):
    """`bench.py --gpus N` started as a plain process: start N ranks (one per GPU) with torch.distributed.run as a
    CHILD process and exit with its code.  This process makes no GPU call at all -- not even a device count: every
    rank checks for itself that its LOCAL_RANK names a visible GPU and fails the run otherwise."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world):
    """Launch plumbing only (CPU test of `--gpus N`): rendezvous over gloo, the same rank accounting as the real
    run, NO filter -- the line says so and carries no value."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ranks = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(ranks, torch.tensor([rank], dtype=torch.int64))
        seen = sorted(int(r[0]) for r in ranks)
        dist.barrier()
        dist.destroy_process_group()
    else:
        seen = [0]
    if seen != list(range(world)):
        raise SystemExit(f"rank accounting broken: {seen}")
    if rank == 0:
        print(json.dumps({"metric": "particle-steps/sec", "value": None, "unit": "particle-steps/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "dry_run": True, "ranks_seen": seen}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the clocks of an idle GPU take ~20 filter calls (60 ms) to settle (tools/warmup_probe.py: first 20 calls
    # 3.08 ms each, then 2.97 ms); the whole default run is still well under a second of GPU time
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--particles", type=int, default=1 << 20)
    ap.add_argument("--cpu-serial-particles", type=int, default=1 << 17)
    ap.add_argument("--cpu-parallel-particles", type=int, default=1 << 20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--propagate-variant", type=int, default=0, help="0 = library default; see dpomp_set_propagate_variant")
    ap.add_argument("--profile-every", type=int, default=10,
                    help="HIP-event brackets around the kernels of every n-th timed step (the markers cost GPU time)")
    ap.add_argument("--no-smc2", action="store_true", help="skip the SMC^2 leg (BASELINE configuration 4) after the timed region")
    ap.add_argument("--write-smc2-reference", default=None, metavar="FILE",
                    help="single GPU only: store the SMC^2 leg's evidence in FILE (committed as profiles/r03/smc2_config4_single_gpu.json: "
                         "the value runs on more GPUs must reproduce bit for bit)")
    ap.add_argument("--smc2-theta", type=int, default=512)
    ap.add_argument("--smc2-particles", type=int, default=1 << 16)
    ap.add_argument("--dry-run", action="store_true",
                    help="launch plumbing only (gloo rendezvous, rank accounting, no GPU, no filter): for the CPU test of --gpus N")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:  # plain `python bench.py --gpus N`: become the launcher
        raise SystemExit(launch_ranks(args, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU")
    if args.dry_run:
        return dry_run(args, rank, world)

    import numpy as np
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the particle filter has no CPU path")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:  # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        # n_gpus is only claimed if `world` distinct ranks sit on `world` distinct devices of this node
        me = torch.tensor([rank, local_rank], dtype=torch.int64, device="cuda")
        everyone = torch.zeros((world, 2), dtype=torch.int64, device="cuda")
        dist.all_gather_into_tensor(everyone, me)
        everyone = everyone.cpu().numpy()
        if sorted(everyone[:, 0].tolist()) != list(range(world)) or len(set(everyone[:, 1].tolist())) != world:
            raise SystemExit(f"--gpus {world}: ranks/devices seen {everyone.tolist()} are not {world} distinct GPUs")

    import dpomp_amd as dp

    y = [dp.Observation(t, 1, 1.0, v) for t, v in POOLEY]
    model = dp.generate_model("SIS", [100, 1])
    pf = dp.ParticleFilter(model, y, device=local_rank)
    stream = torch.cuda.current_stream()
    pf.set_stream(stream.cuda_stream)
    if args.propagate_variant:
        pf.set_propagate_variant(args.propagate_variant)
    P, T, S = args.particles, len(y), model.n_states
    n_calls = args.warmup + args.steps
    # The only collective of the path: the log-ML estimates of the independent filters (chains), all-gathered over
    # RCCL.  The chains do not wait for each other's estimates, so the all-gather of step i is asynchronous and
    # overlaps the filter of step i+1; every handle is waited for before the closing fence of the timed region.
    gathered = torch.zeros((n_calls, world), dtype=torch.float64, device="cuda")
    mine_host = torch.zeros(n_calls, dtype=torch.float64).pin_memory() if dist is not None else None
    mine_dev = torch.zeros(n_calls, dtype=torch.float64, device="cuda")
    pending = []

    def step(i):
        ll = pf.loglik(THETA, P, seed=1000 + i, seg_base=rank)
        if dist is not None:
            mine_host[i] = float(ll[0])
            mine_dev[i:i + 1].copy_(mine_host[i:i + 1], non_blocking=True)
            pending.append(dist.all_gather_into_tensor(gathered[i], mine_dev[i:i + 1], async_op=True))
        return ll

    def drain():
        while pending:
            pending.pop().wait()

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    drain()
    pf.set_profiling(max(1, args.profile_every))  # HIP events around the kernels of every n-th step, on the launch stream
    lls, events = [], 0
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
        lls.append(float(out[0]))
        events += pf.last_event_count()
    drain()
    fence()
    elapsed = time.perf_counter() - t0
    pf.set_profiling(False)
    ll_all_ranks = None
    if dist is not None:  # the gathered estimates: this rank's column must be what its filter returned
        got = gathered[args.warmup:].cpu().numpy()
        assert np.array_equal(got[:, rank], np.array(lls)), "all-gathered log-ML differs from the local result"
        ll_all_ranks = float(got.mean())
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0])
        ev = torch.tensor([float(events)], dtype=torch.float64, device="cuda")
        dist.all_reduce(ev)
        events = float(ev[0])

    # BASELINE configuration 4 after the timed region: the one path with real inter-GPU traffic (theta-level migration)
    smc2 = None
    leg = None
    if not args.no_smc2:
        # (a failure of this leg must not take the timed result with it: it is reported in the line instead)
        try:
            leg = smc2_config4(dp, lambda model, y: dp.ParticleFilter(model, y, device=local_rank), args.smc2_theta,
                               args.smc2_particles, distributed=dist is not None)
            leg_error = None
        except Exception as e:  # noqa: BLE001
            leg_error = f"{type(e).__name__}: {e}"
        if dist is not None:  # every rank reports the leg or none does
            ok = torch.tensor([0 if leg_error else 1], dtype=torch.int64, device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok[0]) == 0:
                leg, leg_error = None, leg_error or "the leg failed on another rank"
        if leg is None:
            smc2 = {"error": leg_error}
    if leg is not None:
        rows = [leg["rows_received"]]
        if dist is not None:
            r_all = torch.zeros(world, dtype=torch.int64, device="cuda")
            dist.all_gather_into_tensor(r_all, torch.tensor([leg["rows_received"]], dtype=torch.int64, device="cuda"))
            rows = [int(v) for v in r_all.cpu()]
            t_all = torch.tensor([leg["seconds"]], dtype=torch.float64, device="cuda")
            dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
            leg["seconds"] = float(t_all[0])
        row_bytes = 4 * 4 * ((args.smc2_particles + 1023) // 1024 * 1024)  # int32 state, 4 components, rows padded to 1024 slots
        ref_file = os.path.join(ROOT, "profiles", "r03", "smc2_config4_single_gpu.json")
        ref = json.load(open(ref_file)) if os.path.exists(ref_file) else None
        same_shape = ref and ref.get("theta_particles") == args.smc2_theta and ref.get("state_particles") == args.smc2_particles
        smc2 = {"workload": f"SEIR, {args.smc2_theta} theta-particles x {args.smc2_particles} state particles, run_pibis (SMC^2), "
                            f"theta-particles sharded over {world} rank(s)",
                "seconds": round(leg["seconds"], 4), "n_resample": leg["n_resample"], "rows_received_per_rank": rows,
                "bytes_received_per_rank": [r * row_bytes for r in rows], "minus_ln_evidence": leg["minus_ln_evidence"],
                "posterior_mean": leg["posterior_mean"], "acceptance_rate": round(leg["acceptance_rate"], 4), "status": leg["status"],
                "particle_steps_per_increment_pass": args.smc2_theta * args.smc2_particles,
                "equals_committed_single_gpu_evidence": (leg["minus_ln_evidence"] == ref["minus_ln_evidence"]) if same_shape else None,
                "committed_single_gpu_evidence": "profiles/r03/smc2_config4_single_gpu.json" if same_shape else None}
    if rank == 0:
        stats = pf.kernel_stats()
        alg = algorithmic_bytes(P, S)
        kernels = {}
        for name, (n, ms) in stats.items():
            if n == 0:
                continue
            avg_us = ms / n * 1e3
            k = {"launches": n, "avg_us": round(avg_us, 2)}
            if alg.get(name):
                k["algorithmic_bytes"] = alg[name]
                k["achieved_GBs"] = round(alg[name] / (avg_us * 1e-6) / 1e9, 1)
            kernels[name] = k
        # The memory-bound stage of the path: normalise (k_wsum) + scan (k_offsets); the resampling gather of a full call
        # runs inside the next k_propagate.  The roofline object is quoted for k_offsets, the streaming kernel of that
        # stage with the most algorithmic bytes per launch, in both accountings; at this workload (16 MiB of state, 36 MiB working
        # set) every streaming kernel is launch/latency-bound and served from L2 / Infinity Cache, so `frac` says little
        # about HBM: the fractions at working sets beyond the 256 MiB Infinity Cache are in profiles/ (hbm_scale_probe).
        mv = moved_bytes(P, S)
        for name, k in kernels.items():
            if mv.get(name):
                k["moved_bytes"] = mv[name]
                k["moved_GBs"] = round(mv[name] / (k["avg_us"] * 1e-6) / 1e9, 1)
        rk = "k_offsets" if "k_offsets" in kernels else "k_wsum"
        g = kernels.get(rk, {})
        traffic = pmc_traffic(rk) or (None, "none")
        memory_stage = {
            "kernel": rk,
            "bound": "hbm (launch latency at this size: the 36 MiB working set is served from L2 / Infinity Cache)",
            "achieved": g.get("achieved_GBs"),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(g.get("achieved_GBs", 0.0) / HBM_PEAK_GBS, 4) if g else None,
            "traffic": traffic[0],
            "traffic_source": traffic[1],
            "accounting": "achieved/frac: Int64-state algorithmic bytes (the reference's layout) / kernel time; "
                          "*_moved_bytes: bytes the kernel really moves (int32 state and boundaries)",
            "achieved_moved_bytes": g.get("moved_GBs"),
            "frac_moved_bytes": round(g.get("moved_GBs", 0.0) / HBM_PEAK_GBS, 4) if g else None,
            "peak_measured": HBM_MEASURED_COPY_GBS,
            "working_set_bytes": P * (2 * 4 * S + 8 + 8 + 4),  # two state buffers, log-weights, fixed-point weights, boundaries
            "beyond_infinity_cache": hbm_scale_summary(),
        }
        mem_us = sum(kernels[k]["avg_us"] for k in ("k_wsum", "k_scan_blocks", "k_offsets", "k_gather") if k in kernels)
        mem_bytes = sum(alg[k] for k in ("k_wsum", "k_offsets", "k_gather") if k in kernels)
        memory_stage["stage_achieved_GBs"] = round(mem_bytes / (mem_us * 1e-6) / 1e9, 1) if mem_us else None
        prop = kernels.get("k_propagate", {})
        ceiling, ceiling_source = valu_ceiling("sis")
        events_per_launch = events / world / args.steps / T
        propagate = {
            "avg_us": prop.get("avg_us"),
            "events_per_s": round(events_per_launch / (prop.get("avg_us", 1) * 1e-6)) if prop else None,
            "events_per_launch": round(events_per_launch),
            "events_per_particle_per_call": round(events / world / args.steps / P, 1),
            "hbm_achieved_GBs": prop.get("achieved_GBs"),
            "bound": "VALU issue + wave divergence (not HBM, not MFMA)",
            "valu_ceiling_events_per_s": round(ceiling),
            "lane_utilisation_static_schedule": [round(float(u), 3) for u in pf.lane_utilisation()],
        }
        # The dominant kernel (97 % of GPU time) has neither an HBM nor an MFMA roof: its ceiling is VALU issue --
        # SIMDs x clock x 64 lanes / (4 cycles x issue slots per event), issue slots from the ISA of this build.
        ptraffic = pmc_traffic("k_propagate_c") or pmc_traffic("k_propagate") or (None, "none")
        roofline = {
            "kernel": "k_propagate (k_propagate_c<RSir<2>,32>)",
            "bound": "valu-issue",
            "achieved": propagate["events_per_s"],
            "peak": round(ceiling),
            "unit": "events/s",
            "frac": round(propagate["events_per_s"] / ceiling, 4) if propagate.get("events_per_s") else None,
            "traffic": ptraffic[0],
            "traffic_source": ptraffic[1],
            "share_of_kernel_time": round(prop["avg_us"] * prop["launches"] / sum(k["avg_us"] * k["launches"] for k in kernels.values()), 3) if prop else None,
            "algorithmic_events_per_launch": propagate["events_per_launch"],
            "algorithmic_bytes_per_launch": alg["k_propagate"],
            "peak_source": ceiling_source + "; 1024 SIMDs x 2.4 GHz x 64 lanes / (4 cycles x issue slots)",
            "timing": f"HIP events stamped with each kernel's own begin and end (hipExtLaunchKernelGGL start/stop events on the "
                      f"launch stream), kernels of every {max(1, args.profile_every)}th timed step; the same quantity rocprofv3 reports per "
                      "dispatch (profiles/)",
            "note": "frac = events per launch / average launch duration / VALU-issue ceiling; the memory-bound stage "
                    "(normalise + scan; the resampling gather runs inside the next propagation's loads) is under memory_stage",
            "memory_stage": memory_stage,
        }
        if propagate.get("events_per_s"):
            propagate["valu_frac"] = roofline["frac"]
        mean_ll = float(np.mean(lls))
        result = {
            "metric": "particle-steps/sec",
            "value": world * P * T * args.steps / elapsed,
            "unit": "particle-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "Pooley SIS data set of the reference (5 observations); synthetic in the sense of no dataset download",
            "config": {"workload": "Pooley SIS, 2^20 particles, single bootstrap PF per GPU (propagate+weight+resample, T=5)",
                       "particles_per_filter": P, "filters_per_gpu": 1, "observations": T, "theta": THETA,
                       "parallelism": f"{world} independent filters, all-gather of log-ML" if world > 1 else "single filter"},
            "events_per_s": events / elapsed,
            "log_ml": {"mean": mean_ll, "exact": EXACT_LL, "abs_err_vs_exact": abs(mean_ll - EXACT_LL),
                       "sd_over_steps": float(np.std(lls)), "mean_over_all_ranks": ll_all_ranks},
            "roofline": roofline,
            "propagate": propagate,
            "kernels": kernels,
        }
        if smc2 is not None:
            result["smc2_config4"] = smc2
            if args.write_smc2_reference and world == 1:
                with open(args.write_smc2_reference, "w") as f:
                    json.dump({"theta_particles": args.smc2_theta, "state_particles": args.smc2_particles, "seed": 3,
                               "minus_ln_evidence": smc2["minus_ln_evidence"], "posterior_mean": smc2["posterior_mean"],
                               "n_resample": smc2["n_resample"], "seconds_single_gpu": smc2["seconds"]}, f, indent=1)
        if world == 1 and not args.no_cpu_baseline:
            # log-ML abs error vs the CPU path at identical seed, on the CPU baseline's own run (by default the whole
            # 2^20-particle workload): bit parity expected
            result["cpu_baseline"] = cpu_baseline(dp, args.cpu_serial_particles, args.cpu_parallel_particles)
            Pc = args.cpu_parallel_particles
            gpu_ll = float(pf.loglik(THETA, Pc, seed=1)[0])
            cpu_ll = result["cpu_baseline"].pop("ll")
            result["log_ml"]["abs_err_vs_cpu_same_seed"] = abs(gpu_ll - cpu_ll)
            result["log_ml"]["same_seed_particles"] = Pc
            result["speedup_vs_cpu_allcores"] = result["value"] / result["cpu_baseline"]["value"]
            result["speedup_vs_cpu_serial"] = result["value"] / result["cpu_baseline"]["serial_value"]
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
