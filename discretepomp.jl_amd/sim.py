"""Forward simulation and the reference's simulation / observation file formats.

* ``gillespie_sim(model, parameters; tmax, num_obs, n_sims)``   src/DiscretePOMP.jl:134-152, src/hmm_sim.jl:86-102
* ``generate_observations``                                      src/hmm_sim.jl:75-82
* ``Event``, ``Particle``, ``SimResults``                        src/hmm_structs.jl:12-15, :51-58, :83-88
* ``save_to_file(results::SimResults, dpath)``                   src/hmm_utils.jl:35-72

The Gillespie loop runs on the GPU (``dpomp_simulate_events``: one trajectory per lane, events appended
to event-major buffers); this module rebuilds the reference's result structs from the event record.
The reference's ``n_sims > 1`` branch refers to an undefined type and its ``save_to_file`` indexes a
vector of vectors as a matrix (both throw); the documented intent is implemented here.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List

import numpy as np

from .model import DPOMPModel, Observation
from .particle_filter import ParticleFilter

MAX_TRAJ = 196000            # src/DiscretePOMP.jl:40
C_DEFAULT_OBS_PROP = 1.0     # src/hmm_sim.jl:74


@dataclass
class Event:
    time: float
    event_type: int  # 1-based index into the rate function / transition matrix


@dataclass
class Particle:
    theta: np.ndarray
    initial_condition: List[int]
    final_condition: List[int]
    trajectory: List[Event]
    prior: float
    log_like: List[float] = field(default_factory=lambda: [0.0, 0.0])


@dataclass
class SimResults:
    model_name: str
    particle: Particle
    population: List[List[int]]   # system state after every event
    observations: List[Observation]


def generate_observations(tmax: float, num_obs: int, n_states: int) -> List[Observation]:
    """'Blank' observations at tmax/num_obs, 2 tmax/num_obs, ..., tmax (src/hmm_sim.jl:75-82)."""
    step = tmax / num_obs
    return [Observation(step * (i + 1), 1, C_DEFAULT_OBS_PROP, [0] * n_states) for i in range(num_obs)]


def gillespie_sim(model: DPOMPModel, parameters, tmax: float = 100.0, num_obs: int = 5, n_sims: int = 1, seed: int = 1,
                  device: int = 0, verbose: bool = True):
    """Run a Doob-Gillespie simulation of ``model``; returns a :class:`SimResults` (a list of them if
    ``n_sims > 1``).  ``observations[i].val`` is what ``model.obs_function`` makes of the state at each
    observation time (``dmy_obs_fn``: the state itself)."""
    theta = np.asarray(parameters, dtype=np.float64)
    y = generate_observations(tmax, num_obs, model.n_states)
    if verbose:
        print(f"Running: {model.model_name} DGA for θ := {list(theta)}" + (f" x {n_sims}" if n_sims > 1 else ""), end="")
    pf = ParticleFilter(model, y, device=device)
    try:
        counts, times, types, states = pf.simulate_events(theta, n_sims=n_sims, seed=seed, max_events=MAX_TRAJ)
    finally:
        pf.close()
    prior_lp = float(model.prior.logpdf(theta)) if model.prior is not None else 0.0
    trans = np.asarray(model.m_transition)
    out = []
    for i in range(n_sims):
        ev = [Event(float(t), int(e)) for t, e in zip(times[i], types[i])]
        pop = np.asarray(model.initial_condition)[None, :] + np.cumsum(trans[np.asarray(types[i], dtype=np.int64) - 1], axis=0) \
            if len(ev) else np.zeros((0, model.n_states), dtype=np.int64)
        obs = [Observation(o.time, o.obs_id, o.prop, list(o.val)) for o in y]
        log_like = 0.0
        for k, o in enumerate(obs):
            state_k = [int(v) for v in states[k, :, i]]
            log_like += model.obs_model(o, state_k, theta)   # cmpt_ll against the still-blank observation, hmm_sim.jl:69
            model.obs_function(o, state_k, theta)            # observe && obs_function(...)  :97
        final = [int(v) for v in states[-1, :, i]]
        particle = Particle(theta.copy(), list(model.initial_condition), final, ev, prior_lp,
                            [float("-inf") if counts[i] < 0 else log_like, 0.0])
        out.append(SimResults(model.model_name, particle, pop.tolist(), obs))
    if verbose:
        print(" - finished.")
    return out[0] if n_sims == 1 else out


def save_to_file(results: SimResults, dpath: str):
    """``sim.csv`` (time, event, state after the event) and ``obs.csv`` (time, id, values) in ``dpath``
    (src/hmm_utils.jl:35-72; ``dpath`` is used as a prefix exactly like the reference's ``string(dpath, "sim.csv")``)."""
    d = os.path.dirname(dpath) if not dpath.endswith(os.sep) else dpath
    if d:
        os.makedirs(d, exist_ok=True)
    n_states = len(results.particle.initial_condition)
    with open(dpath + "sim.csv", "w") as f:
        f.write("time, event" + "".join(f",{p}" for p in range(1, n_states + 1)))
        for ev, pop in zip(results.particle.trajectory, results.population):
            f.write(f"\n{ev.time},{ev.event_type}" + "".join(f",{v}" for v in pop))
    with open(dpath + "obs.csv", "w") as f:
        f.write("time,id" + "".join(f",{p}" for p in range(1, len(results.observations[0].val) + 1)))
        for o in results.observations:
            f.write(f"\n{o.time},{o.obs_id}" + "".join(f",{v}" for v in o.val))
