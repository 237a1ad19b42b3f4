"""Exact log p(y | theta) for any catalogue model with a finite (or truncated) state space -- TEST INFRASTRUCTURE.

The particle filter estimates p(y | theta) without bias; for a continuous-time Markov chain on a finite state space
that quantity is computable without any random number: enumerate the states reachable from the initial condition,
build the sparse generator Q from the reference's rate functions and transition matrices, and run the hidden-Markov
forward recursion

    alpha_t = (alpha_{t-1} exp(Q dt_t)) * g_t ,   log p(y_1..T) = sum_t log sum(alpha_t)

with exp(Q dt) applied by uniformisation (Poisson-weighted powers of the jump chain I + Q/Lambda; the truncation
error of every interval is bounded by the Poisson tail that was cut and is returned).  g_t is the Gaussian emission
of partial_gaussian_obs_model (src/hmm_examples.jl:59-67).

Everything here follows the reference's model definitions directly (not the C oracle's rate code), so it pins the
oracle's and the GPU's rate evaluators, transition matrices and observation indices independently of any RNG:

    SIS / SIR   sir_rf   src/hmm_examples.jl:107-110, m_transition :181 / :177   obs seq 2
    SEIR        seir_rf  :117-121, m_transition :189, obs seq 3
    LOTKA       lotka_rf :149-154, m_transition :198, obs seq 2  (unbounded: truncated to a box, the mass that
                leaves the box is absorbed, counted as `leak` and contributes nothing to the likelihood, so the
                value returned is a lower bound that is tight to `leak`)
"""
from __future__ import annotations

import math

import numpy as np
import scipy.sparse as sp

def _n(x):
    return x.sum(1)


# (rate function over an array of states [n, S] -> [n, E]; transition matrix [E, S]; 0-based observed component)
# Names ending in _FD are the frequency-dependent variants (src/hmm_examples.jl:126-144: the infection rate divided by
# the population size).
MODELS = {
    "SI": (lambda th, x: np.stack([th[0] * x[:, 0] * x[:, 1]], 1), np.array([[-1, 1]]), 1),          # si_rf :103-105, :173
    "SI_FD": (lambda th, x: np.stack([th[0] * x[:, 0] * x[:, 1] / _n(x)], 1), np.array([[-1, 1]]), 1),  # si_rf_fd :126-128
    "SIR_FD": (lambda th, x: np.stack([th[0] * x[:, 0] * x[:, 1] / _n(x), th[1] * x[:, 1]], 1),        # sir_rf_fd :130-133
               np.array([[-1, 1, 0], [0, -1, 1]]), 1),
    "SIS_FD": (lambda th, x: np.stack([th[0] * x[:, 0] * x[:, 1] / _n(x), th[1] * x[:, 1]], 1),
               np.array([[-1, 1], [1, -1]]), 1),
    "SEI": (lambda th, x: np.stack([th[0] * x[:, 0] * x[:, 2], th[1] * x[:, 1]], 1),                  # sei_rf :112-115, :185-186
            np.array([[-1, 1, 0], [0, -1, 1]]), 2),
    "SEI_FD": (lambda th, x: np.stack([th[0] * x[:, 0] * x[:, 2] / _n(x), th[1] * x[:, 1]], 1),        # sei_rf_fd :135-138
               np.array([[-1, 1, 0], [0, -1, 1]]), 2),
    "SEIR_FD": (lambda th, x: np.stack([th[0] * x[:, 0] * x[:, 2] / _n(x), th[1] * x[:, 1], th[2] * x[:, 2]], 1),  # :140-144
                np.array([[-1, 1, 0, 0], [0, -1, 1, 0], [0, 0, -1, 1]]), 2),
    "SEIS": (lambda th, x: np.stack([th[0] * x[:, 0] * x[:, 2], th[1] * x[:, 1], th[2] * x[:, 2]], 1),  # seir_rf, :193-194
             np.array([[-1, 1, 0], [0, -1, 1], [1, 0, -1]]), 2),
    "SEIS_FD": (lambda th, x: np.stack([th[0] * x[:, 0] * x[:, 2] / _n(x), th[1] * x[:, 1], th[2] * x[:, 2]], 1),
                np.array([[-1, 1, 0], [0, -1, 1], [1, 0, -1]]), 2),
    # rossmac_rf :159-168, m_transition :203 (mosquito birth / deaths, human and mosquito infection, human recovery);
    # the mosquito population is unbounded: box-truncated like LOTKA (0/0 -> rate 0 where a denominator vanishes)
    "ROSSMAC": (lambda th, x: np.stack([th[0] * (x[:, 2] + x[:, 3]), th[0] * x[:, 2], th[0] * x[:, 3],
                                        th[1] * _safe_div(x[:, 0] * x[:, 3], x[:, 0] + x[:, 1]),
                                        th[2] * _safe_div(x[:, 1] * x[:, 2], x[:, 2] + x[:, 3]), th[3] * x[:, 1]], 1),
                np.array([[0, 0, 1, 0], [0, 0, -1, 0], [0, 0, 0, -1], [-1, 1, 0, 0], [0, 0, -1, 1], [1, -1, 0, 0]]), 1),
    "SIS": (lambda th, x: np.stack([th[0] * x[:, 0] * x[:, 1], th[1] * x[:, 1]], 1),
            np.array([[-1, 1], [1, -1]]), 1),
    "SIR": (lambda th, x: np.stack([th[0] * x[:, 0] * x[:, 1], th[1] * x[:, 1]], 1),
            np.array([[-1, 1, 0], [0, -1, 1]]), 1),
    "SEIR": (lambda th, x: np.stack([th[0] * x[:, 0] * x[:, 2], th[1] * x[:, 1], th[2] * x[:, 2]], 1),
             np.array([[-1, 1, 0, 0], [0, -1, 1, 0], [0, 0, -1, 1]]), 2),
    "LOTKA": (lambda th, x: np.stack([th[0] * x[:, 1], th[1] * x[:, 0] * x[:, 1], th[2] * x[:, 0]], 1),
              np.array([[0, 1], [1, -1], [-1, 0]]), 1),
}


def _safe_div(a, b):
    return np.divide(a, b, out=np.zeros_like(a, dtype=float), where=b != 0)


def _encode(x, base):
    key = np.zeros(x.shape[0], dtype=np.int64)
    for s in range(x.shape[1]):
        key = key * base + x[:, s]
    return key


def reachable_states(trans, x0, rate_fn, bound=None):
    """All states reachable from x0 through events of positive rate (for SOME positive theta: the structure of the
    catalogue's rate functions does not depend on theta), component-wise within [0, bound] (bound: a scalar or one
    limit per component)."""
    S = len(x0)
    if bound is not None:
        bound = np.asarray(bound)  # a scalar, or one upper limit per component
    base = int(bound.max() if bound is not None else sum(x0)) + 2
    ones = np.ones(8)
    known = np.array([x0], dtype=np.int64)
    frontier = known
    known_keys = _encode(known, base)
    while len(frontier):
        r = rate_fn(ones, frontier.astype(float))
        cand = []
        for e in range(trans.shape[0]):
            nx = frontier[r[:, e] > 0] + trans[e]
            ok = (nx >= 0).all(1)
            if bound is not None:
                ok &= (nx <= bound).all(1)
            cand.append(nx[ok])
        cand = np.concatenate(cand)
        keys, idx = np.unique(_encode(cand, base), return_index=True)
        new = ~np.isin(keys, known_keys)
        frontier = cand[idx[new]]
        known = np.concatenate([known, frontier])
        known_keys = np.concatenate([known_keys, keys[new]])
    order = np.argsort(known_keys)
    return known[order], known_keys[order], base


def generator(model, theta, x0, bound=None):
    """(states [n, S], sparse Q^T over n states + one absorbing leak state, total exit rates)."""
    rate_fn, trans, _ = MODELS[model]
    if bound is not None:
        bound = np.asarray(bound)
    states, keys, base = reachable_states(trans, list(x0), rate_fn, bound)
    n = len(states)
    th = np.zeros(8)
    th[:len(theta)] = theta
    r = rate_fn(th, states.astype(float))
    rows, cols, vals = [], [], []
    src = np.arange(n)
    for e in range(trans.shape[0]):
        nx = states + trans[e]
        k = _encode(np.clip(nx, 0, None), base)
        pos = np.searchsorted(keys, k)
        pos_c = np.minimum(pos, n - 1)
        inside = (nx >= 0).all(1) & (keys[pos_c] == k)
        if bound is not None:
            inside &= (nx <= bound).all(1)
        live = r[:, e] > 0
        dst = np.where(inside, pos_c, n)  # n = leak
        rows.append(dst[live])
        cols.append(src[live])
        vals.append(r[live, e])
    tot = r.sum(1)
    rows.append(src)
    cols.append(src)
    vals.append(-tot)
    QT = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n + 1, n + 1))
    return states, QT, tot


def _expm_apply(QT, lam, t, v, tol=1e-13):
    """v exp(Q t) by uniformisation; returns (result, Poisson tail mass that was cut)."""
    if t <= 0.0:
        return v, 0.0
    a = lam * t
    PT = sp.identity(QT.shape[0], format="csr") + QT / lam
    kmax = int(a + 12.0 * math.sqrt(a + 1.0) + 30)
    logw = -a
    acc = np.zeros_like(v)
    mass = 0.0
    for k in range(kmax + 1):
        w = math.exp(logw)
        if w > 0.0:
            acc += w * v
            mass += w
        if k > a and 1.0 - mass < tol:
            break
        v = PT @ v
        logw += math.log(a / (k + 1))
    return acc, max(0.0, 1.0 - mass)


def transient_distribution(model, theta, x0, t, bound=None, t0=0.0):
    """P(X(t) = x | X(t0) = x0) over the reachable states: (states [n, S], probabilities [n], mass that left the box).
    The law the forward simulation and the filter's propagation step sample from."""
    states, QT, tot = generator(model, theta, x0, bound)
    n = len(states)
    lam = float(tot.max()) * 1.02 + 1e-12
    alpha = np.zeros(n + 1)
    alpha[int(np.flatnonzero((states == np.asarray(x0)).all(1))[0])] = 1.0
    alpha, _ = _expm_apply(QT, lam, float(t) - t0, alpha)
    return states, alpha[:n], float(alpha[n])


def exact_loglik(model, theta, x0, obs_times, obs_values, sigma=2.0, bound=None, t0=0.0, obs_ids=None):
    """Returns dict(ll, increments, n_states, leak, truncation).  obs_values[t] = the observed count that the
    Gaussian emission compares with the model's observed component (sum(y.val[y_seq]) of the reference)."""
    _, _, obs_comp = MODELS[model]
    states, QT, tot = generator(model, theta, x0, bound)
    n = len(states)
    lam = float(tot.max()) * 1.02 + 1e-12
    alpha = np.zeros(n + 1)
    start = np.flatnonzero((states == np.asarray(x0)).all(1))
    alpha[int(start[0])] = 1.0
    tmp1 = math.log(1.0 / (math.sqrt(2.0 * math.pi) * sigma))
    tmp2 = 2.0 * sigma * sigma
    xs = states[:, obs_comp].astype(float)
    incs, leak, trunc = [], 0.0, 0.0
    t_prev = t0
    for i, (t, y) in enumerate(zip(obs_times, obs_values)):
        alpha, cut = _expm_apply(QT, lam, float(t) - t_prev, alpha)
        trunc = max(trunc, cut)
        t_prev = float(t)
        leak = max(leak, float(alpha[n]))
        if obs_ids is not None and obs_ids[i] <= 0:
            continue  # propagate only (src/hmm_particle_filter.jl:58)
        g = np.exp(tmp1 - (float(y) - xs) ** 2 / tmp2)
        alpha[:n] *= g
        alpha[n] = 0.0  # trajectories that left the box: dropped (lower bound)
        c = alpha.sum()
        incs.append(math.log(c))
        alpha /= c
    return {"ll": float(sum(incs)), "increments": incs, "n_states": n, "leak": leak, "truncation": trunc}


if __name__ == "__main__":
    import sys
    import time
    t0 = time.time()
    r = exact_loglik("SIS", (0.003, 0.1), [100, 1], [20, 40, 60, 80, 100], [18, 65, 70, 66, 67])
    print("SIS", r["ll"], r["n_states"], r["truncation"], f"{time.time() - t0:.1f}s")
    if "--seir" in sys.argv:
        t0 = time.time()
        r = exact_loglik("SEIR", (0.005, 0.2, 0.1), [100, 0, 1, 0], [20, 40, 60, 80, 100], [19, 24, 6, 0, 0])
        print("SEIR", r, f"{time.time() - t0:.1f}s")
