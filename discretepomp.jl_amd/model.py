"""Host-side mirror of the reference's model plugin surface.

Same names and argument meaning as DiscretePOMP.jl (paths relative to the reference tree):

* ``Observation``                    src/hmm_structs.jl:30-35
* ``DPOMPModel``                     src/hmm_structs.jl:107-116
* ``generate_model``                 src/hmm_examples.jl:99-211
* ``generate_custom_model``          src/hmm_examples.jl:237-239
* ``partial_gaussian_obs_model``     src/hmm_examples.jl:59-67
* ``generate_weak_prior``            src/hmm_examples.jl:33-35
* ``dmy_obs_fn``                     src/hmm_examples.jl:6-8
* ``get_observations``               src/hmm_utils.jl:19-31

Indices that the reference takes 1-based (``seq``, ``y_seq``, ``t0_index``, and the
``theta`` / ``factors`` of a :class:`RateTerm`, which transcribe Julia expressions such as
``parameters[1] * population[1] * population[2]``) are 1-based here too.

The one deliberate difference: ``rate_function`` cannot be an arbitrary closure, because it
has to run on the GPU.  It is a :class:`RateFunction` -- a table of :class:`RateTerm` that is
(a) callable on the host with the reference's in-place signature
``rate_function(output, parameters, population)`` and (b) flattened into the C descriptor the
on-device rate evaluator interprets (or replaced by a compiled built-in for catalogue models).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass, field
from typing import Callable, Iterable, List, Optional, Sequence

import numpy as np

from . import _abi


# --------------------------------------------------------------------------------------
# Observation / data
# --------------------------------------------------------------------------------------
@dataclass
class Observation:
    """A single observation (hmm_structs.jl:30-35). ``obs_id < 1``: not a resampling step."""
    time: float
    obs_id: int = 1
    prop: float = 1.0
    val: List[int] = field(default_factory=list)


def _parse_delimited(path: str):
    rows = []
    with open(path, "r") as f:
        lines = [ln.strip() for ln in f.read().splitlines()]
    lines = [ln for ln in lines if ln]
    header = [c.strip() for c in lines[0].split(",")]
    try:
        [float(c) for c in header]
        body = lines  # no header row
    except ValueError:
        body = lines[1:]
    for ln in body:
        rows.append([float(c.strip()) for c in ln.split(",")])
    return rows


def get_observations(source, time_col: int = 1, type_col: int = 0, val_seq: Optional[Sequence[int]] = None):
    """Array of :class:`Observation` from a CSV path, a 2-D array or a DataFrame (hmm_utils.jl:19-31).

    Times are in column ``time_col`` (1-based), the optional observation type in ``type_col``
    (0 = every row has ``obs_id = 1``), values in ``val_seq`` (default: columns 2..end).
    The reference's data files carry a leading tab and ``", "`` separators (data/pooley.csv).
    """
    if isinstance(source, (str, os.PathLike)):
        rows = _parse_delimited(os.fspath(source))
    elif hasattr(source, "to_numpy"):
        rows = source.to_numpy().tolist()
    else:
        rows = np.asarray(source).tolist()
    obs = []
    for row in rows:
        ncol = len(row)
        cols = list(val_seq) if val_seq is not None else list(range(2, ncol + 1))
        obs_type = 1 if type_col == 0 else int(row[type_col - 1])
        obs.append(Observation(float(row[time_col - 1]), obs_type, 1.0, [int(row[c - 1]) for c in cols]))
    obs.sort(key=lambda o: o.time)
    return obs


def dmy_obs_fn(y: Observation, population, parameters):
    """Dummy observation function (hmm_examples.jl:6-8): ``y.val .= population``."""
    y.val[:] = [int(v) for v in population]


# --------------------------------------------------------------------------------------
# Rate functions
# --------------------------------------------------------------------------------------
@dataclass(frozen=True)
class RateTerm:
    """One event rate.

    ``form="left"``:  ``((theta * f0) * f1) / d`` -- Julia's left-associated product with each
    integer promoted as it is reached, e.g. ``parameters[1] * population[1] * population[2] /
    sum(population)`` (hmm_examples.jl:131).
    ``form="ratio"``: ``theta * ((f0 * f1) / d)`` with the product taken in integers first,
    e.g. ``parameters[2] * (population[1] * population[4] / (population[1] + population[2]))``
    (hmm_examples.jl:165).
    Each factor / the denominator is the sum of the listed (1-based) state components.
    """
    theta: int
    factors: tuple = ()
    denom: tuple = ()
    form: str = "left"

    def evaluate(self, parameters, population) -> float:
        th = float(parameters[self.theta - 1])
        sums = [sum(int(population[i - 1]) for i in f) for f in self.factors]
        d = sum(int(population[i - 1]) for i in self.denom) if self.denom else None
        if self.form == "left":
            r = th
            for s in sums:
                r = r * float(s)
            if d is not None:
                r = r / float(d) if d != 0 else (math.nan if r == 0 or math.isnan(r) else math.copysign(math.inf, r))
            return r
        prod = 1
        for s in sums:
            prod *= s
        q = float(prod)
        if d is not None:
            q = q / float(d) if d != 0 else (math.nan if prod == 0 else math.copysign(math.inf, prod))
        return th * q


def _mask(indices: Iterable[int]) -> int:
    m = 0
    for i in indices:
        if i < 1:
            raise ValueError("state indices are 1-based")
        m |= 1 << (i - 1)
    return m


class RateFunction:
    """Event-rate table with the reference's call signature ``f(output, parameters, population)``."""

    def __init__(self, terms: Sequence[RateTerm], builtin_id: int = _abi.BUILTIN_GENERIC, name: str = "custom"):
        self.terms = [t if isinstance(t, RateTerm) else RateTerm(*t) for t in terms]
        for t in self.terms:
            if len(t.factors) > _abi.MAX_FACTORS:
                raise ValueError(f"a rate term has at most {_abi.MAX_FACTORS} integer factors")
            if t.form not in ("left", "ratio"):
                raise ValueError("RateTerm.form must be 'left' or 'ratio'")
        self.builtin_id = builtin_id
        self.name = name

    @property
    def n_events(self) -> int:
        return len(self.terms)

    @property
    def n_params(self) -> int:
        return max(t.theta for t in self.terms)

    def __call__(self, output, parameters, population):
        for e, t in enumerate(self.terms):
            output[e] = t.evaluate(parameters, population)
        return output


class RateSource:
    """A rate function given as HIP C++ source, compiled at run time (hipRTC) into the propagation kernel --
    for models the :class:`RateTerm` table cannot express.  The text must define

        __device__ void dpomp_user_rates(const double* theta, const int* x, double* r);

    (0-based ``theta[K]``, state ``x[S]``, write the ``E`` un-cumulated rates to ``r``).  ``host`` is an optional
    Python callable with the reference's signature ``f(output, parameters, population)`` for host-side use."""

    builtin_id = _abi.BUILTIN_USER
    terms: list = []

    def __init__(self, source: str, n_events: int, n_params: int, host: Optional[Callable] = None, name: str = "user_rf"):
        self.source, self.n_events, self.n_params, self.host, self.name = source, int(n_events), int(n_params), host, name

    def __call__(self, output, parameters, population):
        if self.host is None:
            raise NotImplementedError("this RateSource has no host implementation (pass host=...)")
        return self.host(output, parameters, population)


# --------------------------------------------------------------------------------------
# Observation model and prior
# --------------------------------------------------------------------------------------
def _as_index_tuple(seq) -> tuple:
    if isinstance(seq, (int, np.integer)):
        return (int(seq),)
    return tuple(int(i) for i in seq)


class GaussianObsModel:
    """``partial_gaussian_obs_model`` (hmm_examples.jl:59-67): callable ``(y, population, theta)``
    returning ``log(1/(sqrt(2pi) sigma)) - (sum(y.val[y_seq]) - sum(population[seq]))^2 / (2 sigma^2)``."""

    def __init__(self, sigma: float = 2.0, seq=(2,), y_seq=None):
        self.sigma = float(sigma)
        self.seq = _as_index_tuple(seq)
        self.y_seq = _as_index_tuple(y_seq) if y_seq is not None else self.seq
        self.tmp1 = math.log(1.0 / (math.sqrt(2.0 * math.pi) * self.sigma))
        self.tmp2 = 2.0 * self.sigma * self.sigma

    def __call__(self, y: Observation, population, theta=None) -> float:
        d = sum(int(y.val[i - 1]) for i in self.y_seq) - sum(int(population[i - 1]) for i in self.seq)
        return self.tmp1 - float(d * d) / self.tmp2


class SourceObsModel:
    """An observation model given as device code: the rate source of the model (:class:`RateSource`) must then also
    define ``__device__ double dpomp_user_obs(const double* theta, const int* x, const long long* y, int n_y)``
    returning ``log p(y | x, theta)`` -- the counterpart of the reference's arbitrary ``obs_model(y, population,
    theta)`` closure (hmm_structs.jl:113).  ``host`` is an optional Python callable with that signature."""

    def __init__(self, host: Optional[Callable] = None):
        self.host = host

    def __call__(self, y: Observation, population, theta=None) -> float:
        if self.host is None:
            raise NotImplementedError("this SourceObsModel has no host implementation (pass host=...)")
        return self.host(y, population, theta)


def partial_gaussian_obs_model(sigma: float = 2.0, seq=(2,), y_seq=None) -> GaussianObsModel:
    return GaussianObsModel(sigma, seq, y_seq)


class UniformProductPrior:
    """``generate_weak_prior(n, b)`` (hmm_examples.jl:33-35): product of U(0, b)."""

    def __init__(self, lower, upper):
        self.lower = np.asarray(lower, dtype=float)
        self.upper = np.asarray(upper, dtype=float)

    def __len__(self):
        return len(self.lower)

    def logpdf(self, theta) -> float:
        th = np.asarray(theta, dtype=float)
        if np.any(th < self.lower) or np.any(th > self.upper):
            return -math.inf
        return float(-np.sum(np.log(self.upper - self.lower)))

    def rand(self, rng: np.random.Generator, n: Optional[int] = None):
        """``rand(prior, n)``: K x n matrix (column per draw), or a K vector when n is None."""
        if n is None:
            return rng.uniform(self.lower, self.upper)
        return rng.uniform(self.lower[:, None], self.upper[:, None], size=(len(self.lower), n))


def generate_weak_prior(n: int, b: float = 1.0) -> UniformProductPrior:
    return UniformProductPrior(np.zeros(n), np.full(n, float(b)))


# --------------------------------------------------------------------------------------
# The model
# --------------------------------------------------------------------------------------
@dataclass
class DPOMPModel:
    """Mirror of the reference's public model struct (hmm_structs.jl:107-116)."""
    model_name: str
    rate_function: "RateFunction | RateSource"
    initial_condition: List[int]
    m_transition: np.ndarray  # E x S
    obs_function: Callable = dmy_obs_fn
    obs_model: GaussianObsModel = field(default_factory=GaussianObsModel)
    prior: object = None
    t0_index: int = 0  # 0: initial time fixed at 0.0; i: theta[i] (1-based)
    max_events_per_interval: int = 0  # 0: library default. Not in the reference (its PF has no cap).

    def __post_init__(self):
        self.initial_condition = [int(v) for v in self.initial_condition]
        self.m_transition = np.asarray(self.m_transition, dtype=np.int64)
        if self.m_transition.ndim != 2:
            raise ValueError("m_transition must be an E x S matrix")
        if self.prior is None:
            self.prior = generate_weak_prior(self.m_transition.shape[0])

    @property
    def n_events(self) -> int:
        return int(self.m_transition.shape[0])

    @property
    def n_states(self) -> int:
        return len(self.initial_condition)

    @property
    def n_params(self) -> int:
        return max(self.rate_function.n_params, self.t0_index)


_T = lambda *a, **k: RateTerm(*a, **k)

_CATALOGUE = {
    # name: (density-dependent terms, builtin id, builtin id of the frequency-dependent variant or None,
    #        m_transition, observed state `seq`, model_name as stored by the reference)
    "SI": ([_T(1, ((1,), (2,)))], _abi.BUILTIN_SI, _abi.BUILTIN_SI_FD, [[-1, 1]], 2, "SI"),
    "SIR": ([_T(1, ((1,), (2,))), _T(2, ((2,),))], _abi.BUILTIN_SIR, _abi.BUILTIN_SIR_FD,
            [[-1, 1, 0], [0, -1, 1]], 2, "SIR"),
    "SIS": ([_T(1, ((1,), (2,))), _T(2, ((2,),))], _abi.BUILTIN_SIR, _abi.BUILTIN_SIR_FD,
            [[-1, 1], [1, -1]], 2, "SIS"),
    "SEI": ([_T(1, ((1,), (3,))), _T(2, ((2,),))], _abi.BUILTIN_SEI, _abi.BUILTIN_SEI_FD,
            [[-1, 1, 0], [0, -1, 1]], 3, "SEI"),
    "SEIR": ([_T(1, ((1,), (3,))), _T(2, ((2,),)), _T(3, ((3,),))], _abi.BUILTIN_SEIR, _abi.BUILTIN_SEIR_FD,
             [[-1, 1, 0, 0], [0, -1, 1, 0], [0, 0, -1, 1]], 3, "SEIR"),
    "SEIS": ([_T(1, ((1,), (3,))), _T(2, ((2,),)), _T(3, ((3,),))], _abi.BUILTIN_SEIR, _abi.BUILTIN_SEIR_FD,
             [[-1, 1, 0], [0, -1, 1], [1, 0, -1]], 3, "SEIS"),
    "LOTKA": ([_T(1, ((2,),)), _T(2, ((1,), (2,))), _T(3, ((1,),))], _abi.BUILTIN_LOTKA, None,
              [[0, 1], [1, -1], [-1, 0]], 2, "PN"),
    "ROSSMAC": ([_T(1, ((3, 4),)), _T(1, ((3,),)), _T(1, ((4,),)),
                 _T(2, ((1,), (4,)), (1, 2), "ratio"), _T(3, ((2,), (3,)), (3, 4), "ratio"), _T(4, ((2,),))],
                _abi.BUILTIN_ROSSMAC, None,
                [[0, 0, 1, 0], [0, 0, -1, 0], [0, 0, 0, -1], [-1, 1, 0, 0], [0, 0, -1, 1], [1, -1, 0, 0]], 2, "SIAB"),
}


def generate_model(model_name: str, initial_condition, freq_dep: bool = False, obs_error: float = 2.0):
    """Predefined model catalogue (hmm_examples.jl:99-211).

    Names: SI, SIR, SIS, SEI, SEIR, SEIS, LOTKA, ROSSMAC.  ``freq_dep`` divides the contact rate
    by ``sum(population)`` for the epidemiological models.  As in the reference an unknown name
    prints a message and returns ``None``.
    """
    entry = _CATALOGUE.get(model_name)
    if entry is None:
        print(f" - SORRY: model name '{model_name}' not recognised.")
        return None
    terms, b_dd, b_fd, trans, seq, stored_name = entry
    n_states = len(trans[0])
    if len(initial_condition) != n_states:
        raise ValueError(f"{model_name} has {n_states} state components, got {len(initial_condition)}")
    if freq_dep and b_fd is not None:  # LOTKA / ROSSMAC ignore freq_dep, as the reference does
        t0 = terms[0]
        terms = [RateTerm(t0.theta, t0.factors, tuple(range(1, n_states + 1)), "left")] + list(terms[1:])
        builtin = b_fd
    else:
        builtin = b_dd
    rf = RateFunction(terms, builtin, name=model_name.lower() + ("_rf_fd" if freq_dep and b_fd is not None else "_rf"))
    trans = np.asarray(trans, dtype=np.int64)
    return DPOMPModel(stored_name, rf, list(initial_condition), trans, dmy_obs_fn,
                      partial_gaussian_obs_model(obs_error, seq=seq), generate_weak_prior(trans.shape[0]), 0)


def generate_custom_model(model_name: str, rate_function, initial_condition, m_transition, obs_function=dmy_obs_fn,
                          obs_error: float = 2.0, obs_model: Optional[GaussianObsModel] = None, prior=None,
                          t0_index: int = 0):
    """``generate_custom_model`` (hmm_examples.jl:237-239).  ``rate_function`` is a
    :class:`RateFunction` or a list of :class:`RateTerm`."""
    if not isinstance(rate_function, (RateFunction, RateSource)):
        rate_function = RateFunction(rate_function)
    m_transition = np.asarray(m_transition, dtype=np.int64)
    if obs_model is None:
        obs_model = partial_gaussian_obs_model(obs_error)
    if prior is None:
        prior = generate_weak_prior(m_transition.shape[0])
    return DPOMPModel(model_name, rate_function, list(initial_condition), m_transition, obs_function, obs_model, prior,
                      t0_index)


# --------------------------------------------------------------------------------------
# Flattening into the C descriptor (what get_private_model does for the Julia PF,
# DiscretePOMP.jl:96-99)
# --------------------------------------------------------------------------------------
class ModelDescriptor:
    """Owns a ``dpomp_model_desc`` and the arrays it points to."""

    def __init__(self, model: DPOMPModel, obs_data: Sequence[Observation], force_generic: bool = False,
                 jit: "bool | None" = None):
        S, E = model.n_states, model.n_events
        if S > _abi.MAX_STATES or E > _abi.MAX_EVENTS:
            raise ValueError(f"model exceeds {_abi.MAX_STATES} states / {_abi.MAX_EVENTS} events")
        if model.m_transition.shape != (E, S):
            raise ValueError("m_transition must be n_events x n_states")
        if model.rate_function.n_events != E:
            raise ValueError("rate_function and m_transition disagree on the number of events")
        user_obs = isinstance(model.obs_model, SourceObsModel)
        if user_obs and not isinstance(model.rate_function, RateSource):
            raise TypeError("a SourceObsModel needs a RateSource whose text defines dpomp_user_obs")
        if not user_obs and not isinstance(model.obs_model, GaussianObsModel):
            raise TypeError("obs_model must come from partial_gaussian_obs_model or be a SourceObsModel (the device evaluates it)")
        if len(obs_data) < 1:
            raise ValueError("no observations")
        K = model.n_params
        if K > _abi.MAX_PARAMS:
            raise ValueError(f"model exceeds {_abi.MAX_PARAMS} parameters")
        self.model = model
        self.n_states, self.n_events, self.n_params, self.n_obs = S, E, K, len(obs_data)
        d = _abi.ModelDescC()
        d.abi_version = _abi.ABI_VERSION
        d.n_states, d.n_events, d.n_params = S, E, K
        d.builtin_id = _abi.BUILTIN_GENERIC if force_generic else model.rate_function.builtin_id
        d.t0_index = int(model.t0_index)
        for s in range(S):
            d.initial_condition[s] = model.initial_condition[s]
        self.rate_source = model.rate_function.source if isinstance(model.rate_function, RateSource) else None
        for e in range(E):
            for s in range(S):
                d.transition[e][s] = int(model.m_transition[e, s])
            if self.rate_source is not None:
                continue
            t = model.rate_function.terms[e]
            d.rate[e].theta_index = t.theta - 1
            d.rate[e].form = _abi.RATE_LEFT if t.form == "left" else _abi.RATE_RATIO
            d.rate[e].n_factors = len(t.factors)
            for f, idx in enumerate(t.factors):
                d.rate[e].factor_mask[f] = _mask(idx)
            d.rate[e].denom_mask = _mask(t.denom)
        V = len(obs_data[0].val)
        if any(len(o.val) != V for o in obs_data):
            raise ValueError("every Observation.val must have the same length")
        om = model.obs_model
        if user_obs:
            d.obs.kind = _abi.OBS_USER
            d.obs.sigma = 1.0
        else:
            d.obs.kind = _abi.OBS_GAUSSIAN
            d.obs.state_mask = _mask(om.seq)
            d.obs.value_mask = _mask(om.y_seq)
            d.obs.sigma = om.sigma
            if max(om.seq) > S or max(om.y_seq) > V:
                raise IndexError("observation model indexes beyond the state / observation vector")
        self.obs_time = np.ascontiguousarray([o.time for o in obs_data], dtype=np.float64)
        self.obs_id = np.ascontiguousarray([o.obs_id for o in obs_data], dtype=np.int64)
        self.obs_value = np.ascontiguousarray([[int(v) for v in o.val] for o in obs_data], dtype=np.int64).reshape(-1)
        d.n_obs = len(obs_data)
        d.n_obs_values = V
        d.obs_time = self.obs_time.ctypes.data_as(C.POINTER(C.c_double))
        d.obs_id = self.obs_id.ctypes.data_as(C.POINTER(C.c_int64))
        d.obs_value = self.obs_value.ctypes.data_as(C.POINTER(C.c_int64))
        d.max_events_per_interval = int(model.max_events_per_interval)
        # a rate table outside the catalogue: compiled at create time (True), interpreted on the device (False), or the
        # library's process-wide default (None) -- decided per model, in the descriptor
        d.flags = 0 if jit is None else (_abi.FLAG_TABLE_COMPILE if jit else _abi.FLAG_TABLE_INTERPRET)
        self.c = d

    def byref(self):
        return C.byref(self.c)
