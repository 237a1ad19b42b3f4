"""Test-only doubles: a filter bank with the surface of dpomp_amd.ParticleFilter computed by the CPU oracle."""
import numpy as np


class OraclePop:
    def __init__(self, B, P, S):
        self.B, self.P = B, P
        self.pops = [np.zeros((P, S), dtype=np.int64) for _ in range(B)]

    def gather(self, anc):
        self.pops = [self.pops[a].copy() for a in anc]

    def take(self, other, take):
        for b, t in enumerate(take):
            if t >= 0:
                self.pops[b] = other.pops[t].copy()

    def export_state(self):
        import torch
        return torch.from_numpy(np.stack(self.pops))

    def import_state(self, t):
        self.pops = [t[i].numpy().copy() for i in range(t.shape[0])]

    def export_rows(self, rows):
        import torch
        S = self.pops[0].shape[1]
        return torch.from_numpy(np.stack([self.pops[r] for r in rows]) if len(rows) else np.zeros((0, self.P, S), dtype=np.int64))

    def import_rows(self, slots, rows):
        for s, r in zip(slots, rows):
            self.pops[int(s)] = r.numpy().copy()

    def close(self):
        pass


_POOL = None


def threaded(fn, n):
    """[fn(0), ..., fn(n-1)] on a few threads: independent oracle filters (ctypes releases the GIL, one filter is one
    serial C call), so the CPU suite uses the cores it has."""
    global _POOL
    if n < 2:
        return [fn(i) for i in range(n)]
    if _POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(8, len(os.sched_getaffinity(0)))))
    return list(_POOL.map(fn, range(n)))


class OracleBank:
    def __init__(self, dp, orc, model, obs):
        self.orc, self.desc = orc, dp.ModelDescriptor(model, obs)
        self.n_obs, self.n_states = self.desc.n_obs, self.desc.n_states

    def alloc_population(self, B, P):
        return OraclePop(B, P, self.n_states)

    def partial_log_likelihood(self, pop, theta, ymin, ymax, seed=1, seg_base=0):
        theta = np.asarray(theta)
        return np.array(threaded(lambda b: self.orc.pf_partial(self.desc, pop.pops[b], theta[:, b], ymin, ymax, seed=seed,
                                                               seg=seg_base + b).ll, pop.B))


# every catalogue model of generate_model (src/hmm_examples.jl:171-204), density and frequency dependent:
# (name, initial condition, freq_dep, a parameter vector that gives an outbreak)
CATALOGUE_CASES = [
    ("SI", [100, 1], False, [0.003]),
    ("SI", [100, 1], True, [0.3]),
    ("SIR", [100, 1, 0], False, [0.003, 0.1]),
    ("SIR", [100, 1, 0], True, [0.3, 0.1]),
    ("SIS", [100, 1], False, [0.003, 0.1]),
    ("SIS", [100, 1], True, [0.3, 0.1]),
    ("SEI", [100, 0, 1], False, [0.005, 0.2]),
    ("SEI", [100, 0, 1], True, [0.5, 0.2]),
    ("SEIR", [100, 0, 1, 0], False, [0.005, 0.2, 0.1]),
    ("SEIR", [100, 0, 1, 0], True, [0.5, 0.2, 0.1]),
    ("SEIS", [100, 0, 1], False, [0.005, 0.2, 0.1]),
    ("SEIS", [100, 0, 1], True, [0.5, 0.2, 0.1]),
    ("LOTKA", [70, 70], False, [0.5, 0.0025, 0.3]),
    ("ROSSMAC", [100, 10, 200, 20], False, [0.02, 0.05, 0.04, 0.1]),
]


def synthetic_data(dp, orc, model, theta, T=4, tmax=20.0, seed=99, obs_ids=None):
    """Observations simulated with the oracle's forward simulator (the default flow of the reference:
    gillespie_sim with dmy_obs_fn, y.val = state)."""
    times = [tmax * (i + 1) / T for i in range(T)]
    blank = [dp.Observation(t, 1, 1.0, [0] * model.n_states) for t in times]
    d = dp.ModelDescriptor(model, blank)
    traj = orc.simulate(d, 1, theta, seed=seed)  # [T][S][1]
    ids = obs_ids or [1] * T
    return [dp.Observation(t, ids[i], 1.0, [int(v) for v in traj[i, :, 0]]) for i, t in enumerate(times)]


# ---- exact likelihood of a catalogue case (oracle/exact_ctmc.py: sparse generator + uniformisation) -------------------
ROSSMAC_EXACT_X0, ROSSMAC_EXACT_BOUND = [50, 5, 60, 8], [55, 55, 130, 60]  # mosquito box: leaked mass < 1e-14
_EXACT_CACHE = {}


def catalogue_case_exact(dp, orc, name, x0, fd, theta):
    """(model, observations, exact log p(y | theta)) of one catalogue case on data simulated by the oracle's simulator
    (4 observations up to t = 20, y.val = state).  Ross-Macdonald's mosquito population is unbounded, so its exact
    value is computed from a smaller initial condition on a box that holds all but 1e-14 of the probability mass."""
    from oracle import exact_ctmc
    key = (name, bool(fd))
    if key not in _EXACT_CACHE:
        bound = None
        if name == "ROSSMAC":
            x0, bound = ROSSMAC_EXACT_X0, ROSSMAC_EXACT_BOUND
        model = dp.generate_model(name, x0, freq_dep=fd)
        y = synthetic_data(dp, orc, model, theta)
        mkey = name + ("_FD" if fd else "")
        comp = exact_ctmc.MODELS[mkey][2]
        r = exact_ctmc.exact_loglik(mkey, theta, x0, [o.time for o in y], [o.val[comp] for o in y], bound=bound)
        assert r["leak"] < 1e-12 and r["truncation"] < 1e-12
        _EXACT_CACHE[key] = (model, y, r["ll"])
    return _EXACT_CACHE[key]

