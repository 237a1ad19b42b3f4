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

    def close(self):
        pass


class OracleBank:
    def __init__(self, dp, orc, model, obs):
        self.orc, self.desc = orc, dp.ModelDescriptor(model, obs)
        self.n_obs, self.n_states = self.desc.n_obs, self.desc.n_states

    def alloc_population(self, B, P):
        return OraclePop(B, P, self.n_states)

    def partial_log_likelihood(self, pop, theta, ymin, ymax, seed=1, seg_base=0):
        theta = np.asarray(theta)
        return np.array([self.orc.pf_partial(self.desc, pop.pops[b], theta[:, b], ymin, ymax, seed=seed, seg=seg_base + b).ll
                         for b in range(pop.B)])
