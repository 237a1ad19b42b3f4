"""Exact log p(y | theta) for the closed SIS model -- TEST INFRASTRUCTURE.

The PF estimates this quantity; for closed SIS it is computable without any RNG.  With
N = S0 + I0 conserved, I in 0..N is a birth-death CTMC with generator
    q(I -> I+1) = beta * (N - I) * I     (sir_rf event 1, hmm_examples.jl:108, m_transition [-1 1])
    q(I -> I-1) = gamma * I              (event 2, :109, m_transition [1 -1])
The hidden-Markov forward recursion with the Gaussian emission of
partial_gaussian_obs_model (hmm_examples.jl:59-67) on state 2 (= I) gives
    log p(y_1..T) = sum_t log( alpha_{t-1} expm(Q dt_t) . g_t ).
This pins the oracle (and through it the GPU filter) at distribution level: the PF is
unbiased on the likelihood scale, E[exp(ll_hat)] = p(y | theta).
"""
from __future__ import annotations

import math

import numpy as np
from scipy.linalg import expm


def sis_generator(beta: float, gamma: float, N: int, freq_dep: bool = False) -> np.ndarray:
    Q = np.zeros((N + 1, N + 1))
    for i in range(N + 1):
        up = beta * (N - i) * i
        if freq_dep:
            up = up / N
        down = gamma * i
        if i < N:
            Q[i, i + 1] = up
        if i > 0:
            Q[i, i - 1] = down
        Q[i, i] = -(Q[i].sum())
    return Q


def exact_sis_loglik(theta, x0, obs_times, obs_cases, sigma: float = 2.0, freq_dep: bool = False, t0: float = 0.0):
    """Returns (total log-likelihood, per-observation increments)."""
    beta, gamma = float(theta[0]), float(theta[1])
    N = int(x0[0] + x0[1])
    Q = sis_generator(beta, gamma, N, freq_dep)
    alpha = np.zeros(N + 1)
    alpha[int(x0[1])] = 1.0
    tmp1 = math.log(1.0 / (math.sqrt(2.0 * math.pi) * sigma))
    tmp2 = 2.0 * sigma * sigma
    states = np.arange(N + 1)
    incs = []
    t_prev = t0
    cache = {}
    for t, y in zip(obs_times, obs_cases):
        dt = round(float(t) - t_prev, 12)
        if dt not in cache:
            cache[dt] = expm(Q * dt)
        alpha = alpha @ cache[dt]
        g = np.exp(tmp1 - (y - states) ** 2 / tmp2)
        alpha = alpha * g
        c = alpha.sum()
        incs.append(math.log(c))
        alpha = alpha / c
        t_prev = float(t)
    return float(sum(incs)), incs


POOLEY_TIMES = [20.0, 40.0, 60.0, 80.0, 100.0]
POOLEY_CASES = [18, 65, 70, 66, 67]

if __name__ == "__main__":
    for th in [(0.003, 0.1), (0.0032, 0.1), (0.0025, 0.08), (0.004, 0.15)]:
        ll, incs = exact_sis_loglik(th, [100, 1], POOLEY_TIMES, POOLEY_CASES)
        print(th, f"{ll:.10f}", [f"{v:.6f}" for v in incs])


def exact_evidence(lower, upper, x0=(100, 1), times=POOLEY_TIMES, cases=POOLEY_CASES, n=40, sigma=2.0):
    """-ln p(y) and posterior means under a uniform prior on [lower, upper] (2-D), by midpoint quadrature
    of the exact likelihood on an n x n grid (20.1676, beta 0.003246, gamma 0.10832 for
    U(0,0.01) x U(0,0.5), the prior of test/runtests.jl:29)."""
    import numpy as _np
    bs = lower[0] + (upper[0] - lower[0]) * (_np.arange(n) + 0.5) / n
    gs = lower[1] + (upper[1] - lower[1]) * (_np.arange(n) + 0.5) / n
    ll = _np.array([[exact_sis_loglik((b, g), x0, times, cases, sigma)[0] for g in gs] for b in bs])
    m = ll.max()
    w = _np.exp(ll - m)
    z = w.mean()
    return {"neg_log_evidence": float(-(m + math.log(z))), "beta_mean": float((w * bs[:, None]).sum() / w.sum()),
            "gamma_mean": float((w * gs[None, :]).sum() / w.sum())}
