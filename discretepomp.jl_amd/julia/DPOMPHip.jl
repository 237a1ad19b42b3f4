# DPOMPHip.jl -- the reference-side binding a DiscretePOMP.jl maintainer would add: the bootstrap
# particle filter behind `get_particle_filter_lpdf` / `partial_log_likelihood!` served by
# libdpomp_hip.so through `ccall` (C ABI: include/dpomp.h).
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: the build image has no Julia.  The tested twin of this file
# is the ctypes host in discretepomp.jl_amd/_abi.py + particle_filter.py, which marshals the same
# struct and calls the same entry points.
module DPOMPHip

import DiscretePOMP

const LIB = get(ENV, "DPOMP_HIP_LIB", joinpath(@__DIR__, "..", "libdpomp_hip.so"))

const ABI_VERSION = 2       # DPOMP_ABI_VERSION
const MAX_STATES = 32       # DPOMP_MAX_STATES (models beyond 8 states / events / parameters are compiled at create time)
const MAX_EVENTS = 32
const MAX_FACTORS = 4

struct RateTermC            # dpomp_rate_term
    theta_index::Int32      # 0-based
    form::Int32             # 0 = ((θ·f0)·f1)/d ; 1 = θ·((f0·f1)/d)
    n_factors::Int32
    factor_mask::NTuple{MAX_FACTORS,UInt32}
    denom_mask::UInt32
end

struct ObsModelC            # dpomp_obs_model
    kind::Int32             # 1 = partial_gaussian_obs_model
    state_mask::UInt32
    value_mask::UInt32
    reserved::Int32
    sigma::Float64
end

struct ModelDescC           # dpomp_model_desc (field order and padding as in include/dpomp.h)
    abi_version::Int32
    n_states::Int32
    n_events::Int32
    n_params::Int32
    builtin_id::Int32
    t0_index::Int32
    initial_condition::NTuple{MAX_STATES,Int64}
    transition::NTuple{MAX_EVENTS * MAX_STATES,Int32}   # transition[e][s], row-major
    rate::NTuple{MAX_EVENTS,RateTermC}
    obs::ObsModelC
    n_obs::Int32
    n_obs_values::Int32
    obs_time::Ptr{Float64}
    obs_id::Ptr{Int64}
    obs_value::Ptr{Int64}
    max_events_per_interval::Int64
    flags::Int32            # DPOMP_FLAG_*: 1 = interpret a rate table, 2 = compile it, 0 = the library's default
    reserved::Int32
end

# catalogue name (DPOMPModel.model_name as stored by generate_model, src/hmm_examples.jl:171-204)
# -> (builtin id dd, builtin id fd, rate table).  A term is (θ index, factors, denominator, form), 1-based.
mask(idx) = isempty(idx) ? UInt32(0) : reduce(|, (UInt32(1) << (i - 1) for i in idx))
term(th, factors, denom = (), form = 0) =
    RateTermC(th - 1, form, length(factors),
              ntuple(i -> i <= length(factors) ? mask(factors[i]) : UInt32(0), MAX_FACTORS), mask(denom))

const CATALOGUE = Dict(
    "SI" => (1, 5, [term(1, ((1,), (2,)))]),
    "SIR" => (2, 6, [term(1, ((1,), (2,))), term(2, ((2,),))]),
    "SIS" => (2, 6, [term(1, ((1,), (2,))), term(2, ((2,),))]),
    "SEI" => (3, 7, [term(1, ((1,), (3,))), term(2, ((2,),))]),
    "SEIR" => (4, 8, [term(1, ((1,), (3,))), term(2, ((2,),)), term(3, ((3,),))]),
    "SEIS" => (4, 8, [term(1, ((1,), (3,))), term(2, ((2,),)), term(3, ((3,),))]),
    "PN" => (9, 9, [term(1, ((2,),)), term(2, ((1,), (2,))), term(3, ((1,),))]),
    "SIAB" => (10, 10, [term(1, ((3, 4),)), term(1, ((3,),)), term(1, ((4,),)),
                        term(2, ((1,), (4,)), (1, 2), 1), term(3, ((2,), (3,)), (3, 4), 1), term(4, ((2,),))]),
)

mutable struct Filter
    handle::Ptr{Cvoid}
    keep::Any          # arrays the descriptor points to
    n_params::Int
    n_obs::Int
    pops::Vector{WeakRef}   # live populations: freed before the handle (finalizers run in no particular order)
end

# The structs above must have the layout the library was compiled with.
function __init__()
    for (which, T) in ((0, ModelDescC), (1, RateTermC), (2, ObsModelC))
        n = ccall((:dpomp_abi_sizeof, LIB), Cint, (Cint,), which)
        n == sizeof(T) || error("DPOMPHip: sizeof($T) = $(sizeof(T)) but libdpomp_hip.so says $n (ABI mismatch)")
    end
end

"Frees the populations that are still alive, then the handle.  Idempotent; also the finalizer."
function close!(f::Filter)
    f.handle == C_NULL && return
    for w in f.pops
        p = w.value
        p === nothing || free!(p)
    end
    empty!(f.pops)
    ccall((:dpomp_destroy, LIB), Cvoid, (Ptr{Cvoid},), f.handle)
    f.handle = C_NULL
    return
end

check(h, rc) = rc == 0 ? nothing :
    error("dpomp: ", unsafe_string(ccall((:dpomp_last_error, LIB), Cstring, (Ptr{Cvoid},), h)))

"""
    Filter(model::DPOMPModel, y::Vector{Observation}; freq_dep=false, obs_error=2.0, seq=nothing, device=0)

Flattens a catalogue `DPOMPModel` and its data (what `get_private_model` does for the Julia filter,
src/DiscretePOMP.jl:96-99).  The observation model must be `partial_gaussian_obs_model`; since a Julia
closure cannot be inspected, its parameters are passed again (`obs_error`, `seq`; defaults follow
`generate_model`).
"""
function Filter(model::DiscretePOMP.DPOMPModel, y::Vector{DiscretePOMP.Observation};
                freq_dep::Bool = false, obs_error::Float64 = 2.0, seq = nothing, device::Integer = 0)
    haskey(CATALOGUE, model.model_name) || error("no device rate table for model '$(model.model_name)'")
    b_dd, b_fd, terms = CATALOGUE[model.model_name]
    S, E = length(model.initial_condition), size(model.m_transition, 1)
    terms = copy(terms)
    if freq_dep && b_fd != b_dd
        t = terms[1]
        terms[1] = RateTermC(t.theta_index, 0, t.n_factors, t.factor_mask, mask(1:S))
    end
    K = max(maximum(t.theta_index for t in terms) + 1, model.t0_index)
    seq === nothing && (seq = model.model_name in ("SEI", "SEIR", "SEIS") ? (3,) : (2,))
    V = length(y[1].val)
    obs_time = Float64[o.time for o in y]
    obs_id = Int64[o.obs_id for o in y]
    obs_value = Int64[o.val[v] for o in y for v in 1:V]           # observation-major
    pad(v, n, z) = ntuple(i -> i <= length(v) ? v[i] : z, n)
    trans = zeros(Int32, MAX_EVENTS * MAX_STATES)
    for e in 1:E, s in 1:S
        trans[(e - 1) * MAX_STATES + s] = model.m_transition[e, s]
    end
    zero_term = RateTermC(0, 0, 0, ntuple(_ -> UInt32(0), MAX_FACTORS), UInt32(0))
    desc = Ref(ModelDescC(ABI_VERSION, S, E, K, freq_dep ? b_fd : b_dd, model.t0_index,
                          pad(Int64.(model.initial_condition), MAX_STATES, Int64(0)), Tuple(trans),
                          pad(terms, MAX_EVENTS, zero_term),
                          ObsModelC(1 #= DPOMP_OBS_GAUSSIAN =#, mask(seq), mask(seq), 0, obs_error),
                          length(y), V, pointer(obs_time), pointer(obs_id), pointer(obs_value), 0, 0, 0))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve obs_time obs_id obs_value begin
        rc = ccall((:dpomp_create, LIB), Cint, (Ref{ModelDescC}, Cint, Ref{Ptr{Cvoid}}), desc, device, h)
    end
    rc == 0 || error("dpomp_create: ", unsafe_string(ccall((:dpomp_last_global_error, LIB), Cstring, ())))
    f = Filter(h[], (obs_time, obs_id, obs_value), K, length(y), WeakRef[])
    finalizer(close!, f)
    return f
end

"""
    Filter(model, y, rate_source::String; n_params, obs_error=2.0, seq=(2,), user_obs=false, device=0)

A model whose rate function is arbitrary device code (compiled at run time with hipRTC): `rate_source` must define
`__device__ void dpomp_user_rates(const double* theta, const int* x, double* r)`.  This is the route for a
`DPOMPModel` whose `rate_function` closure is not one of the catalogue's (src/hmm_structs.jl:109).  With
`user_obs=true` the source also defines `__device__ double dpomp_user_obs(const double* theta, const int* x,
const long long* y, int n_y)` (log density), replacing the Gaussian observation model (src/hmm_structs.jl:113).
"""
function Filter(model::DiscretePOMP.DPOMPModel, y::Vector{DiscretePOMP.Observation}, rate_source::String;
                n_params::Integer, obs_error::Float64 = 2.0, seq = (2,), user_obs::Bool = false,
                device::Integer = 0)
    S, E = length(model.initial_condition), size(model.m_transition, 1)
    V = length(y[1].val)
    obs_time = Float64[o.time for o in y]
    obs_id = Int64[o.obs_id for o in y]
    obs_value = Int64[o.val[v] for o in y for v in 1:V]
    pad(v, n, z) = ntuple(i -> i <= length(v) ? v[i] : z, n)
    trans = zeros(Int32, MAX_EVENTS * MAX_STATES)
    for e in 1:E, s in 1:S
        trans[(e - 1) * MAX_STATES + s] = model.m_transition[e, s]
    end
    zero_term = RateTermC(0, 0, 0, ntuple(_ -> UInt32(0), MAX_FACTORS), UInt32(0))
    desc = Ref(ModelDescC(ABI_VERSION, S, E, n_params, 100 #= DPOMP_BUILTIN_USER =#, model.t0_index,
                          pad(Int64.(model.initial_condition), MAX_STATES, Int64(0)), Tuple(trans),
                          ntuple(_ -> zero_term, MAX_EVENTS), ObsModelC(user_obs ? 100 : 1, mask(seq), mask(seq), 0, obs_error),
                          length(y), V, pointer(obs_time), pointer(obs_id), pointer(obs_value), 0, 0, 0))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve obs_time obs_id obs_value begin
        rc = ccall((:dpomp_create_with_source, LIB), Cint, (Ref{ModelDescC}, Cint, Cstring, Ref{Ptr{Cvoid}}),
                   desc, device, rate_source, h)
    end
    rc == 0 || error("dpomp_create_with_source: ", unsafe_string(ccall((:dpomp_last_global_error, LIB), Cstring, ())))
    f = Filter(h[], (obs_time, obs_id, obs_value), n_params, length(y), WeakRef[])
    finalizer(close!, f)
    return f
end

"Make `essc` do what its name says (the reference accepts and ignores it, src/hmm_particle_filter.jl:63-68); 0 = ignore."
set_ess_threshold!(f::Filter, essc::Float64) =
    check(f.handle, ccall((:dpomp_set_ess_threshold, LIB), Cint, (Ptr{Cvoid}, Cdouble), f.handle, essc))

"""
    simulate_events(f, theta; n_sims=1, seed=1, max_events=196000)

`gillespie_sim` with the `Event` history (src/hmm_sim.jl:86-102) for `n_sims` trajectories at once: returns
`(counts, times, types, states)`; `times[1:counts[i], i]`, `types[...]` (1-based event types) are trajectory i,
`states[i, :, t]` its state at observation t.
"""
function simulate_events(f::Filter, theta::Vector{Float64}, S::Integer; n_sims::Integer = 1, seed::Integer = 1,
                         max_events::Integer = 196000)
    counts = Vector{Int64}(undef, n_sims)
    times = Matrix{Float64}(undef, n_sims, max_events)      # event-major in memory: [p + n_sims*(n-1)]
    types = Matrix{Int32}(undef, n_sims, max_events)
    states = Array{Int64,3}(undef, n_sims, S, f.n_obs)
    GC.@preserve theta counts times types states check(f.handle, ccall((:dpomp_simulate_events, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, UInt64, Int64, Int64, Ptr{Int64}, Ptr{Float64}, Ptr{Int32}, Ptr{Int64}),
        f.handle, theta, n_sims, seed, 0, max_events, counts, times, types, states))
    return counts, permutedims(times), permutedims(types), states
end

"""
    get_particle_filter_lpdf(model, obs_data; np=200, rs_type=1, essc=0.3, seed=1) -> θ -> Float64

Drop-in for `DiscretePOMP.get_particle_filter_lpdf` (src/hmm_utils.jl:281-284).
"""
function get_particle_filter_lpdf(model, obs_data; np::Int64 = 200, rs_type::Int64 = 1, essc::Float64 = 0.3,
                                  seed::Integer = 1, kwargs...)
    f = Filter(model, obs_data; kwargs...)
    check(f.handle, ccall((:dpomp_set_resampler, LIB), Cint, (Ptr{Cvoid}, Cint), f.handle, rs_type))
    calls = Ref(UInt64(0))
    function comp_log_pdf(parameters::Array{Float64,1})
        out = Ref(0.0)
        s = UInt64(seed) * 0x9E3779B97F4A7C15 + calls[]
        calls[] += 1
        GC.@preserve parameters begin
            check(f.handle, ccall((:dpomp_pf_loglik, LIB), Cint,
                  (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, UInt64, Int64, Ref{Float64}, Ptr{UInt32}),
                  f.handle, parameters, 1, np, s, 0, out, C_NULL))
        end
        return out[]
    end
    return comp_log_pdf
end

"""
    loglik(f, theta::Matrix{Float64}, np; seed, seg_base=0) -> Vector{Float64}

B independent filters in one call: `theta` is K x B (Julia's column-major layout IS the ABI's).
"""
function loglik(f::Filter, theta::Matrix{Float64}, np::Integer; seed::Integer = 1, seg_base::Integer = 0)
    B = size(theta, 2)
    out = Vector{Float64}(undef, B)
    GC.@preserve theta out check(f.handle, ccall((:dpomp_pf_loglik, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, UInt64, Int64, Ptr{Float64}, Ptr{UInt32}),
        f.handle, theta, B, np, seed, seg_base, out, C_NULL))
    return out
end

# Device-resident populations: the `pop[p]` matrices run_pibis keeps per θ-particle (src/hmm_ibis.jl:26-30).
mutable struct Population
    f::Filter
    ptr::Ptr{Cvoid}
    B::Int
    P::Int
end

function Population(f::Filter, B::Integer, P::Integer)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(f.handle, ccall((:dpomp_pop_alloc, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Ref{Ptr{Cvoid}}), f.handle, B, P, p))
    pop = Population(f, p[], B, P)
    push!(f.pops, WeakRef(pop))
    finalizer(free!, pop)
    return pop
end

"Frees the device buffers (idempotent; a population whose filter is already closed has been freed by it)."
function free!(pop::Population)
    if pop.ptr != C_NULL && pop.f.handle != C_NULL
        ccall((:dpomp_pop_free, LIB), Cvoid, (Ptr{Cvoid}, Ptr{Cvoid}), pop.f.handle, pop.ptr)
    end
    pop.ptr = C_NULL
    return
end

"partial_log_likelihood!(pop[p], model, θ[:,p], fn_rs, essc, ymin, ymax) for every p at once (src/hmm_ibis.jl:53-56)."
function partial_log_likelihood!(pop::Population, theta::Matrix{Float64}, ymin::Integer, ymax::Integer;
                                 seed::Integer = 1, seg_base::Integer = 0)
    out = Vector{Float64}(undef, pop.B)
    GC.@preserve theta out check(pop.f.handle, ccall((:dpomp_pf_partial, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int32, Int32, UInt64, Int64, Ptr{Float64}, Ptr{UInt32}),
        pop.f.handle, pop.ptr, theta, ymin, ymax, seed, seg_base, out, C_NULL))
    return out
end

"θ-level resampling copy `pop[p] = pop[ancestor[p]]` (src/hmm_ibis.jl:71-75); `ancestors` 1-based."
function gather!(pop::Population, ancestors::Vector{Int64})
    a0 = ancestors .- 1
    GC.@preserve a0 check(pop.f.handle, ccall((:dpomp_pop_gather, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int64}),
                                              pop.f.handle, pop.ptr, a0))
end

"""
    take!(dst, src, take)

Accepted PMMH proposals (src/hmm_ibis.jl:109, `pop[p] .= pop_f`): segment `p` of `dst` becomes segment `take[p]` of
`src` (1-based; 0 = keep `dst[p]`).
"""
function take!(dst::Population, src::Population, take::Vector{Int64})
    t0 = take .- 1            # the ABI is 0-based, -1 = keep
    GC.@preserve t0 check(dst.f.handle, ccall((:dpomp_pop_take, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int64}),
                                              dst.f.handle, dst.ptr, src.ptr, t0))
end

"Lane utilisation of each observation interval in the last call (what the automatic kernel schedule decides on)."
function lane_utilisation(f::Filter)
    out = Vector{Float64}(undef, f.n_obs)
    u = Ref(0.0)
    for k in 1:f.n_obs
        check(f.handle, ccall((:dpomp_get_lane_utilisation, LIB), Cint, (Ptr{Cvoid}, Cint, Ref{Float64}), f.handle, k, u))
        out[k] = u[]
    end
    return out
end

"Segment b (1-based) as the reference's P x S `Array{Int64,2}`."
function download(pop::Population, b::Integer, S::Integer)
    out = Array{Int64,2}(undef, pop.P, S)
    GC.@preserve out check(pop.f.handle, ccall((:dpomp_pop_download, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Int64}),
                                               pop.f.handle, pop.ptr, b - 1, out))
    return out
end

"""
    run_pibis(f, model, theta, ess_rs_crit, ind_prop, alpha, np; n_props=1, seed=1) -> ImportanceSample

SMC^2 (Chopin 2013) with the loop structure, bookkeeping and quirks of `DiscretePOMP.run_pibis`
(src/hmm_ibis.jl:12-135), every per-theta-particle loop of which becomes ONE batched call on device-resident
populations:

    :53-56   gx[p] = partial_log_likelihood!(pop[p], ...)   -> partial_log_likelihood!(pops, theta, obs_min, obs_i)
    :71-75   pop2[p] .= pop[nidx[p]]                        -> gather!(pops, nidx)
    :90-100  pop_f = zeros(...); two partial_log_likelihood! -> the proposals of ALL particles drawn first, their
                                                               filters run as one batch on a second bank `pops_f`
    :108     pop[p] .= pop_f                                -> take!(pops, pops_f, accepted)

Host-side pieces are the reference's own functions (`rs_systematic`, `compute_is_mu_covar!`, `get_prop_density`,
`get_mv_param`, `compute_ess`, `ImportanceSample`).  One deliberate difference: with conditional proposals the
reference adapts `tj` inside the sweep over particles; here a sweep's proposals are drawn with the `tj` of its start
and the accumulated factor is applied afterwards (with `ind_prop = true`, the default, there is no difference).
"""
function run_pibis(f::Filter, model, theta::Array{Float64,2}, ess_rs_crit::Float64, ind_prop::Bool, alpha::Float64,
                   np::Int64; n_props = 1, seed::Integer = 1)
    DP = DiscretePOMP
    outer_p = size(theta, 2)
    start_time = time_ns()
    ess_crit = ess_rs_crit * outer_p
    theta = copy(theta)
    w = ones(outer_p)
    aw = [DP.Distributions.logpdf(model.prior, theta[:, i]) for i in 1:outer_p]
    pops = Population(f, outer_p, np)       # pop[p], :26-30
    pops_f = Population(f, outer_p, np)     # pop_f of every particle's proposal, :90
    mu = zeros(size(theta, 1))
    cv = zeros(size(theta, 1), size(theta, 1))
    k_log = zeros(Int64, 2)
    bme = zeros(2)
    propd = DP.Distributions.MvNormal(Matrix{Float64}(DP.LinearAlgebra.I, size(theta, 1), size(theta, 1)))
    tj = 0.2
    calls = Ref(UInt64(0))
    next_seed() = (calls[] += 1; UInt64(seed) * 0x9E3779B97F4A7C15 + calls[])   # the reference advances a global RNG instead
    obs_min = 1
    for obs_i in 1:f.n_obs
        model.obs_data[obs_i].obs_id > 0 || continue
        gx = partial_log_likelihood!(pops, theta, obs_min, obs_i; seed = next_seed())     # :53-56
        aw .+= gx
        gx .= exp.(gx)
        lml = log(sum(w .* gx) / sum(w))                                                   # :60
        bme[1] += lml
        w .*= gx
        DP.compute_is_mu_covar!(mu, cv, theta, w)
        if DP.compute_ess(w) < ess_crit                                                    # :65
            propd = DP.get_prop_density(cv, propd)
            nidx = DP.rs_systematic(w)                                                     # :70
            theta = theta[:, nidx]
            aw = aw[nidx]
            gather!(pops, Vector{Int64}(nidx))                                             # :74
            mlr = DP.Statistics.mean(gx[nidx]) * exp(lml)
            k_log[1] += outer_p
            mtd_gx = gx[nidx]
            for mki in 1:n_props                                                           # :85
                theta_f = similar(theta)
                for p in 1:outer_p                                                         # :87
                    theta_f[:, p] .= ind_prop ? DP.get_mv_param(propd, 1.0, mu) : DP.get_mv_param(propd, tj, theta[:, p])
                end
                prtf = [DP.Distributions.logpdf(model.prior, theta_f[:, p]) for p in 1:outer_p]
                ok = prtf .!= -Inf
                any(ok) || continue
                th_run = copy(theta_f)
                th_run[:, .!ok] .= theta[:, .!ok]   # proposals outside the prior run a harmless theta and are ignored
                if obs_i == 1                                                              # :92-94
                    gx_f = partial_log_likelihood!(pops_f, th_run, 1, 1; seed = next_seed())
                    aw_f = copy(gx_f)
                else                                                                       # :95-100
                    aw_f = partial_log_likelihood!(pops_f, th_run, 1, obs_i - 1; seed = next_seed())
                    gx_f = partial_log_likelihood!(pops_f, th_run, obs_i, obs_i; seed = next_seed())
                    aw_f .+= gx_f
                end
                aw_f .+= prtf
                take = zeros(Int64, outer_p)
                n_acc = 0
                for p in 1:outer_p
                    ok[p] || continue
                    if exp(aw_f[p] - aw[p]) > rand()                                       # :104 (no proposal-density ratio, as there)
                        mtd_gx[p] = exp(gx_f[p])
                        theta[:, p] .= theta_f[:, p]
                        aw[p] = aw_f[p]
                        take[p] = p
                        n_acc += 1
                    end
                end
                n_acc > 0 && take!(pops, pops_f, take)                                     # :108
                k_log[2] += n_acc
                tj *= alpha^n_acc * 0.999^(count(ok) - n_acc)                              # :110-112, per sweep
            end
            bme[2] += log(mlr / DP.Statistics.mean(mtd_gx))                                # :118
            w .= 1
        else
            bme[2] += log(sum(w .* gx) / sum(w))                                           # :122
        end
        obs_min = obs_i + 1
    end
    DP.compute_is_mu_covar!(mu, cv, theta, w)
    free!(pops); free!(pops_f)
    output = DP.ImportanceSample(mu, cv, theta, w, time_ns() - start_time, -bme)
    println("- finished in ", DP.print_runtime(output.run_time), " (AR = ", round(100.0 * k_log[2] / k_log[1]; sigdigits = 3), "%)")
    return output
end

"""
    run_smc2_analysis(model, obs_data; np=4000, npf=200, ess_rs_crit=0.3, ind_prop=true, alpha=1.002, seed=1, kwargs...)

Drop-in for `DiscretePOMP.run_smc2_analysis` (src/DiscretePOMP.jl:251-257) -- and so for the default algorithm of
`run_ibis_analysis` (:289-302) -- with the particle filter on the GPU.  `kwargs` go to `Filter` (freq_dep, obs_error, seq, device).
"""
function run_smc2_analysis(model::DiscretePOMP.DPOMPModel, obs_data::Array{DiscretePOMP.Observation,1};
                           np = 4000, npf = 200, ess_rs_crit = 0.3, ind_prop = true, alpha = 1.002, seed::Integer = 1, kwargs...)
    f = Filter(model, obs_data; kwargs...)
    mdl = DiscretePOMP.get_private_model(model, obs_data)     # for prior and obs_data, as the reference does (:252)
    theta_init = rand(mdl.prior, np)
    println("Running: ", np, "-particle SMC^2 analysis (model: ", model.model_name, ")")
    try
        return run_pibis(f, mdl, theta_init, ess_rs_crit, ind_prop, alpha, npf; seed = seed)
    finally
        close!(f)
    end
end

end # module
