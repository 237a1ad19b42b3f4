/*
 * dpomp.h -- C ABI of the MI355X-native bootstrap particle filter for
 * discrete-state POMP models (drop-in for the DiscretePOMP.jl particle-filter
 * hot path).
 *
 * The reference has no FFI of its own; the entry points below are what a
 * Julia `ccall` binding for the path would bind.  Each one names the reference
 * interface it replaces (paths relative to the reference tree):
 *
 *   dpomp_create         <- get_private_model              src/DiscretePOMP.jl:96-99
 *                           (DPOMPModel src/hmm_structs.jl:107-116 + Observation[] :30-35
 *                            -> HiddenMarkovModel :119-130)
 *   dpomp_pf_loglik      <- estimate_likelihood            src/hmm_particle_filter.jl:79-84
 *                           and the closure comp_log_pdf   src/hmm_particle_filter.jl:97-99
 *                           (public face get_particle_filter_lpdf src/hmm_utils.jl:281-284)
 *   dpomp_pop_alloc      <- pop = zeros(Int64, P, S)       src/hmm_particle_filter.jl:82,
 *                           per-theta populations          src/hmm_ibis.jl:26-30
 *   dpomp_pf_partial     <- partial_log_likelihood!        src/hmm_particle_filter.jl:39-76
 *   dpomp_pop_gather     <- theta-level ancestor copy      src/hmm_ibis.jl:71-75
 *   dpomp_pop_take       <- accepted PMMH proposal `pop[p] .= pop_f`   src/hmm_ibis.jl:109
 *   dpomp_pop_download   <- reading the caller-owned `pop::Array{Int64,2}` after the call
 *   dpomp_pop_upload     <- writing it before the call
 *   dpomp_simulate       <- gillespie_sim (final states at each observation time only)
 *                           src/hmm_sim.jl:86-102, iterate_particle! :55-70
 *   dpomp_simulate_events<- gillespie_sim with the Event history (`p.trajectory`)
 *
 * Conventions
 *   - every function returns 0 (DPOMP_OK) or a negative DPOMP_E* code; the text of
 *     the last error is available from dpomp_last_error(h) (or
 *     dpomp_last_global_error() when no handle exists yet).  Nothing throws across
 *     the boundary.
 *   - `-inf` in an output log-likelihood is a RESULT (all weights zero), not an error.
 *   - host arrays are caller-owned and only touched during the call; device buffers
 *     are library-owned behind the opaque handles.
 *   - calls block until the result is on the host.  A handle is bound to one device
 *     and is not thread-safe; distinct handles are independent.
 *   - matrices crossing the boundary use the reference's (Julia, column-major)
 *     layouts: theta is K x B (theta[k + K*b]), populations are P x S
 *     (pop[p + P*s]), m_transition is E x S given row-by-row in `transition[e][s]`.
 *   - indices that the reference holds 1-based (t0_index, ymin, ymax) stay 1-based
 *     here so a Julia caller passes them through unchanged.
 *   - randomness is explicit and counter-based: every draw is a pure function of
 *     (seed, global segment index, particle index, observation index, draw number),
 *     so results do not depend on call order, batch composition or how segments are
 *     sharded across GPUs.
 */
#ifndef DPOMP_H
#define DPOMP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPOMP_ABI_VERSION 2

/* Limits of the descriptor.  The reference's DPOMPModel has none (any closure, any size: hmm_structs.jl:107-116); here a
 * model is data.  Models of up to DPOMP_COMPILED_MAX state components, events and parameters with at most two integer
 * factors per rate run on evaluators that are compiled into the library (the catalogue, and a table interpreter as the
 * fallback).  Anything larger -- up to the DPOMP_MAX_* below, e.g. an age-structured model with a dozen compartments --
 * is compiled at create time (hipRTC: rate table or user source turned into an evaluator of exactly its size) and
 * needs hipRTC to be available. */
#define DPOMP_MAX_STATES 32  /* S: state components        */
#define DPOMP_MAX_EVENTS 32  /* E: event types             */
#define DPOMP_MAX_PARAMS 32  /* K: model parameters        */
#define DPOMP_MAX_FACTORS 4  /* integer factors per rate   */
#define DPOMP_COMPILED_MAX 8 /* S, E, K up to which compiled-in evaluators exist (with at most 2 factors per rate) */

/* dpomp_model_desc.flags */
#define DPOMP_FLAG_TABLE_INTERPRET 1 /* a rate table outside the catalogue: interpret it on the device (no hipRTC)  */
#define DPOMP_FLAG_TABLE_COMPILE 2   /* ... compile it at create time, whatever the process-wide default says        */

/* error codes */
#define DPOMP_OK 0
#define DPOMP_EINVAL (-1)   /* bad argument / descriptor              */
#define DPOMP_EDEVICE (-2)  /* HIP runtime error                      */
#define DPOMP_ENOMEM (-3)   /* host or device allocation failed       */
#define DPOMP_ESTATE (-4)   /* call sequence error (e.g. ymin > 1 on a fresh population) */

/* per-segment status bits returned next to the log-likelihood */
#define DPOMP_ST_EVENT_CAP 1u   /* some particle hit max_events_per_interval (its weight was zeroed) */
#define DPOMP_ST_ZERO_WEIGHT 2u /* all weights were zero at some observation: result is -inf */

/* rate-term evaluation forms (association order follows the reference's Julia expressions) */
#define DPOMP_RATE_LEFT 0  /* ((theta * f0) * f1) / d   e.g. hmm_examples.jl:108,:131,:152 */
#define DPOMP_RATE_RATIO 1 /* theta * ((f0 * f1) / d), product in integers  e.g. hmm_examples.jl:165-166 */

/* observation models */
#define DPOMP_OBS_GAUSSIAN 1 /* partial_gaussian_obs_model hmm_examples.jl:59-67 */
#define DPOMP_OBS_USER 100   /* log-density from run-time compiled source: dpomp_user_obs (dpomp_create_with_source) */

/* resamplers (rs_type of get_particle_filter_lpdf, hmm_utils.jl:281) */
#define DPOMP_RS_SYSTEMATIC 1 /* rsp_systematic hmm_pf_resample.jl:24-42 (the reference's only working one) */
#define DPOMP_RS_STRATIFIED 2 /* rsp_stratified :46-63 (throws in the reference; Kitagawa 1996 as documented) */
#define DPOMP_RS_MULTINOMIAL 3 /* rsp_multinomial :5-20 (throws in the reference; inverse-CDF as documented) */

/* built-in (compile-time specialised) rate evaluators; 0 = interpret `rate[]` on device */
#define DPOMP_BUILTIN_GENERIC 0
#define DPOMP_BUILTIN_SI 1       /* si_rf        hmm_examples.jl:103-105 */
#define DPOMP_BUILTIN_SIR 2      /* sir_rf       :107-110 (SIR and SIS)  */
#define DPOMP_BUILTIN_SEI 3      /* sei_rf       :112-115                */
#define DPOMP_BUILTIN_SEIR 4     /* seir_rf      :117-121 (SEIR, SEIS)   */
#define DPOMP_BUILTIN_SI_FD 5    /* si_rf_fd     :126-128                */
#define DPOMP_BUILTIN_SIR_FD 6   /* sir_rf_fd    :130-133                */
#define DPOMP_BUILTIN_SEI_FD 7   /* sei_rf_fd    :135-138                */
#define DPOMP_BUILTIN_SEIR_FD 8  /* seir_rf_fd   :140-144                */
#define DPOMP_BUILTIN_LOTKA 9    /* lotka_rf     :149-154                */
#define DPOMP_BUILTIN_ROSSMAC 10 /* rossmac_rf   :159-168                */
#define DPOMP_BUILTIN_USER 100   /* rates come from source text compiled at run time (dpomp_create_with_source);
                                    `rate[]` is ignored */

/* One event rate:  see DPOMP_RATE_*.  f_i = sum of the state components whose bit is
 * set in factor_mask[i] (computed in integers, then promoted to double as Julia does);
 * d likewise from denom_mask (0 = no division). */
typedef struct dpomp_rate_term {
    int32_t theta_index;  /* 0-based index into theta */
    int32_t form;         /* DPOMP_RATE_* */
    int32_t n_factors;    /* 0..DPOMP_MAX_FACTORS */
    uint32_t factor_mask[DPOMP_MAX_FACTORS];
    uint32_t denom_mask;
} dpomp_rate_term;

/* partial_gaussian_obs_model(sigma; seq, y_seq): log N(sum(y.val[y_seq]) | sum(pop[seq]), sigma) */
typedef struct dpomp_obs_model {
    int32_t kind;         /* DPOMP_OBS_GAUSSIAN */
    uint32_t state_mask;  /* `seq`   : bit i = state component i (0-based) */
    uint32_t value_mask;  /* `y_seq` : bit i = column i of Observation.val */
    int32_t reserved;
    double sigma;
} dpomp_obs_model;

/* Flattened DPOMPModel + Observation[] (hmm_structs.jl:107-116, :30-35). */
typedef struct dpomp_model_desc {
    int32_t abi_version;  /* DPOMP_ABI_VERSION */
    int32_t n_states;     /* S = length(initial_condition) */
    int32_t n_events;     /* E = size(m_transition, 1)     */
    int32_t n_params;     /* K = length(theta)             */
    int32_t builtin_id;   /* DPOMP_BUILTIN_*; the generic `rate[]` must be filled in either way */
    int32_t t0_index;     /* as DPOMPModel.t0_index: 0 = start at t=0, i>0 = theta[i] (1-based) */
    int64_t initial_condition[DPOMP_MAX_STATES];
    int32_t transition[DPOMP_MAX_EVENTS][DPOMP_MAX_STATES]; /* m_transition[e][s] */
    dpomp_rate_term rate[DPOMP_MAX_EVENTS];
    dpomp_obs_model obs;
    int32_t n_obs;        /* T = length(obs_data) */
    int32_t n_obs_values; /* V = length(Observation.val) (same for every observation) */
    const double* obs_time;   /* [T] Observation.time, non-decreasing */
    const int64_t* obs_id;    /* [T] Observation.obs_id (<1: propagate only, no weight/resample) */
    const int64_t* obs_value; /* [T*V] Observation.val, observation-major: obs_value[t*V + v] */
    int64_t max_events_per_interval; /* per particle and interval; <=0 selects the default (1<<22).
                                        The reference PF has no cap (hmm_particle_filter.jl:19-27). */
    int32_t flags;        /* DPOMP_FLAG_*; 0 = the process-wide default (dpomp_set_generic_jit) */
    int32_t reserved;
} dpomp_model_desc;

typedef struct dpomp_handle dpomp_handle;
typedef struct dpomp_pop dpomp_pop;

/* ---- lifecycle ---------------------------------------------------------------- */
int dpomp_device_count(void);
/* sizeof of the structs that cross the boundary (0 dpomp_model_desc, 1 dpomp_rate_term, 2 dpomp_obs_model; -1 for
 * anything else), so a binding in another language can assert its own layout at load time. */
int dpomp_abi_sizeof(int which);
const char* dpomp_version(void);
const char* dpomp_last_global_error(void);

int dpomp_create(const dpomp_model_desc* desc, int device, dpomp_handle** out);
/* A model whose rate function is arbitrary device code -- the closest thing to the reference's
 * `rate_function::Function` closure (hmm_structs.jl:109).  `rate_source` is HIP C++ that defines
 *     __device__ void dpomp_user_rates(const double* theta, const int* x, double* r);
 * writing the E (un-cumulated) event rates for parameter vector theta[K] and state x[S].  If desc->obs.kind is
 * DPOMP_OBS_USER the text must also define the observation model (the reference's `obs_model(y, pop, theta)` closure,
 * hmm_structs.jl:113):
 *     __device__ double dpomp_user_obs(const double* theta, const int* x, const long long* y, int n_y);
 * returning log p(y | x, theta) for Observation.val = y[0..n_y).  It is compiled with
 * hipRTC for gfx950 together with the library's kernels (-ffp-contract=off: results stay reproducible by the
 * same source compiled for the host).  desc->builtin_id must be DPOMP_BUILTIN_USER. */
int dpomp_create_with_source(const dpomp_model_desc* desc, int device, const char* rate_source, dpomp_handle** out);
/* Compile-only check of a rate source (needs no GPU); the compiler log is copied into log[log_size].  With
 * rate_source == NULL the descriptor's rate table is checked: the evaluator dpomp_create would generate from it. */
int dpomp_check_rate_source(const dpomp_model_desc* desc, const char* rate_source, char* log, int64_t log_size);
/* How the handle evaluates rates: a model of the catalogue runs a compiled-in evaluator; a rate table outside the
 * catalogue is turned into source at dpomp_create and compiled with hipRTC like a user's rate source (same arithmetic
 * and bits as the table interpreter, the speed of a compiled-in evaluator); if hipRTC is unavailable, or the switch
 * below is off, the table is interpreted on the device.  Compiled models are cached: in the process and as files under
 * $DPOMP_CACHE_DIR (default ~/.cache/dpomp, "off" = no files), keyed by a hash of everything that goes into the
 * compiler, so only the first create of a model pays the compile (seconds). */
#define DPOMP_EVALUATOR_BUILTIN 1        /* compiled-in (catalogue)                        */
#define DPOMP_EVALUATOR_TABLE 2          /* rate table interpreted on the device           */
#define DPOMP_EVALUATOR_TABLE_COMPILED 3 /* rate table compiled at create time             */
#define DPOMP_EVALUATOR_SOURCE 4         /* user rate source (dpomp_create_with_source)    */
int dpomp_get_evaluator(const dpomp_handle* h);
/* Process-wide default for compiling rate tables: 1 on, 0 off (interpret), -1 = on unless the environment has
 * DPOMP_GENERIC_JIT=0.  Applies to handles created afterwards whose descriptor does not decide for itself
 * (dpomp_model_desc.flags: DPOMP_FLAG_TABLE_INTERPRET / DPOMP_FLAG_TABLE_COMPILE -- the per-model, thread-safe way). */
int dpomp_set_generic_jit(int mode);
void dpomp_destroy(dpomp_handle* h);
const char* dpomp_last_error(const dpomp_handle* h);
/* Run the kernels on an existing HIP stream (hipStream_t passed as void*); NULL = the
 * handle's own stream. */
int dpomp_set_stream(dpomp_handle* h, void* hip_stream);
/* rs_type as in get_particle_filter_lpdf (DPOMP_RS_*; default systematic). */
int dpomp_set_resampler(dpomp_handle* h, int rs_type);

/* ESS-adaptive resampling -- what the reference's dormant `essc` / `ess_rs_crit` argument describes
 * (hmm_utils.jl:276-281) but never does: its test is commented out (hmm_particle_filter.jl:63-68) and it resamples at
 * every weighted observation.  ess_frac = 0 (default) reproduces that.  ess_frac > 0: after weighting, resample only
 * if ESS = (sum w)^2 / sum w^2 < ess_frac * P; otherwise keep the population and carry the log-weights into the next
 * observation (increment = LSE(carried + new) - LSE(carried)).  The carry state lives in the population handle. */
int dpomp_set_ess_threshold(dpomp_handle* h, double ess_frac);

/* Tuning: variant of the propagation kernel.  0 = library default (per observation interval: workgroup
 * compaction on a launch of at least 4096 waves unless the previous call measured a lane utilisation of 0.9 or more for that
 * interval, one particle per lane otherwise; see dpomp_get_lane_utilisation.  Under the compaction schedule the default
 * lays filters of at most 512 particles end to end over the workgroups, so that filters of the reference's default
 * size -- 200 particles -- fill every lane); 1 = one particle per lane; 2, 4 = workgroup-local LDS queue with that many
 * particles per lane and dynamic lane refill; 16, 32 = one particle per lane with workgroup-level compaction every
 * that many events, one filter per workgroup column; 33 = as 32 with the end-to-end layout wherever it applies
 * (P <= 512).  Results do not depend on it. */
int dpomp_set_propagate_variant(dpomp_handle* h, int variant);

/* ---- the particle filter ------------------------------------------------------- */
/* Full filter over observations 1..T for B independent parameter vectors
 * (estimate_likelihood for each theta[:,b]).  out_ll[B]; out_status[B] may be NULL. */
int dpomp_pf_loglik(dpomp_handle* h, const double* theta, int64_t B, int64_t P, uint64_t seed,
                    int64_t seg_base, double* out_ll, uint32_t* out_status);

/* dpomp_pf_loglik for callers that cannot allocate device memory.  It has NO counterpart in the reference: it exists
 * for callers that must not call hipMalloc -- tight serving loops, harnesses that forbid allocation, processes whose
 * device memory is fully budgeted.  Same arguments, same results bit for bit as dpomp_pf_loglik (whatever resampler /
 * ESS threshold the handle is set to), but every working buffer of the call lives in one static device array of
 * DPOMP_STATIC_MAX_BYTES that each GPU the library runs on holds until the process ends; the call performs no
 * hipMalloc / hipFree / hipHostMalloc / event or stream creation (per-kernel profiling is off for it;
 * dpomp_debug_alloc_count() does not move across it).  Like every filter call it ends with a device-to-host copy of
 * the results and a synchronisation of the stream, so it is NOT capturable into a HIP graph.  Calls whose working set
 * (about (8 S + 20) bytes per particle, plus 8 with resamplers 2/3) exceeds the array, or with
 * B > DPOMP_STATIC_MAX_SEGMENTS, are rejected with DPOMP_EINVAL.  All calls on one GPU share that GPU's array, from
 * every handle and thread: calls that overlap on one GPU race -- serialise them.
 * DPOMP_STATIC_MAX_BYTES is a build-time constant of the library (default 16 MiB; `DPOMP_STATIC_MAX_BYTES=<bytes>
 * python discretepomp.jl_amd/build.py` builds another size); dpomp_static_max_bytes() returns the loaded library's. */
#ifndef DPOMP_STATIC_MAX_BYTES
#define DPOMP_STATIC_MAX_BYTES (16u << 20)
#endif
#define DPOMP_STATIC_MAX_SEGMENTS 4096
int dpomp_pf_loglik_static(dpomp_handle* h, const double* theta, int64_t B, int64_t P, uint64_t seed,
                           int64_t seg_base, double* out_ll, uint32_t* out_status);
int64_t dpomp_static_max_bytes(void);

/* Device-resident populations for the incremental form. */
int dpomp_pop_alloc(dpomp_handle* h, int64_t B, int64_t P, dpomp_pop** out);
void dpomp_pop_free(dpomp_handle* h, dpomp_pop* pop);

/* partial_log_likelihood!(pop, model, theta, fn_rs, essc, ymin, ymax) for every segment.
 * ymin == 1 re-initialises the populations; ymin > 1 resumes from obs_time[ymin-1]
 * (1-based, as the reference).  Returns the sum over ymin..ymax of log mean weight. */
int dpomp_pf_partial(dpomp_handle* h, dpomp_pop* pop, const double* theta, int32_t ymin,
                     int32_t ymax, uint64_t seed, int64_t seg_base, double* out_ll,
                     uint32_t* out_status);

/* pop[b] <- pop[ancestors[b]] for all b (0-based ancestors), as hmm_ibis.jl:71-75. */
int dpomp_pop_gather(dpomp_handle* h, dpomp_pop* pop, const int64_t* ancestors);
/* dst[b] <- src[take[b]] for every b with take[b] >= 0 (segments with take[b] < 0 keep their
 * population): installs the populations of accepted proposals, hmm_ibis.jl:109.  dst and src are
 * distinct handles of the same filter with equal P. */
int dpomp_pop_take(dpomp_handle* h, dpomp_pop* dst, const dpomp_pop* src, const int64_t* take);
/* Zero-copy interop (RCCL / torch.distributed population migration): device pointer of the
 * populations as int32 [B][S][stride] (stride = P rounded up to 1024; lanes >= P are padding).
 * Valid until the next call on this population that resamples or gathers (those swap buffers).
 * for_write != 0 also marks the population initialised (the caller is about to fill it). */
int dpomp_pop_device_state(dpomp_handle* h, dpomp_pop* pop, int for_write, void** device_ptr,
                           int64_t* stride);
/* Segment b as a P x S column-major Int64 matrix. */
int dpomp_pop_download(dpomp_handle* h, const dpomp_pop* pop, int64_t b, int64_t* out);
int dpomp_pop_upload(dpomp_handle* h, dpomp_pop* pop, int64_t b, const int64_t* in);

/* Forward simulation of B*P independent trajectories from the initial condition,
 * recording the state at each observation time: out[((b*T + t)*S + s)*P + p]. */
int dpomp_simulate(dpomp_handle* h, const double* theta, int64_t B, int64_t P, uint64_t seed,
                   int64_t seg_base, int64_t* out);

/* gillespie_sim with its event history (hmm_sim.jl:86-102, `push!(p.trajectory, Event(time, et))` :65):
 * P independent trajectories of parameter vector theta[K].  Buffers are EVENT-major:
 * out_times[n*P + p], out_types[n*P + p] (1-based event type) for n < out_counts[p];
 * out_counts[p] = -1 when more than max_events occurred (record truncated, cf. MAX_TRAJ hmm_sim.jl:17-20).
 * out_states[(t*S + s)*P + p] = state at observation time t. */
int dpomp_simulate_events(dpomp_handle* h, const double* theta, int64_t P, uint64_t seed, int64_t seg,
                          int64_t max_events, int64_t* out_counts, double* out_times, int32_t* out_types,
                          int64_t* out_states);

/* ---- inspection (parity tests, bench) ----------------------------------------- */
#define DPOMP_DBG_LOGW 1          /* double[P]  log-weights of the last weighted observation */
#define DPOMP_DBG_OFFSPRING_END 2 /* int32[P]   offspring boundaries N_j of the last resampling */
int dpomp_debug_download(dpomp_handle* h, const dpomp_pop* pop, int what, int64_t b, void* out);

/* Device evaluation of the shared numerics for n inputs: uniforms of particle-stream
 * (seed, seg, p, k), dlog(x), dexp(-x).  Arrays may be NULL. */
int dpomp_selftest_numerics(dpomp_handle* h, int64_t n, uint64_t seed, const double* x_in,
                            double* out_uniform, double* out_log, double* out_exp);

/* Test hook: device quotients a[i] / b[i] by the event loop's range-checked fast division (NaN where b[i] is outside
 * its range, 2^-200 <= b < 2^200) and by the plain device division; both must equal the host's IEEE quotient. */
int dpomp_selftest_division(dpomp_handle* h, int64_t n, const double* a_in, const double* b_in, double* out_fast,
                            double* out_plain);

/* Per-kernel HIP-event timing on the launch stream, accumulated while profiling is on.  on = 0 switches it off,
 * on = n > 0 times the kernels of every n-th filter call (n = 1: every call).  Timed kernels are launched with start
 * and stop events that carry the kernel's own begin and end (the dispatch duration rocprofv3 reports). */
int dpomp_set_profiling(dpomp_handle* h, int on);
int dpomp_kernel_count(void);
const char* dpomp_kernel_name(int kernel_id);
int dpomp_get_kernel_stat(dpomp_handle* h, int kernel_id, int64_t* launches, double* total_ms);
/* Elapsed time of an EMPTY event bracket (two hipEventRecord with nothing between), median of 64, measured when
 * profiling was switched on: what every bracket adds to a kernel's own duration. */
int dpomp_get_null_bracket_ms(dpomp_handle* h, double* ms);
/* Lane utilisation of the propagation towards observation obs_index (1-based) in the most recent call that ran it:
 * sum of per-particle event counts / (64 * sum over waves of the largest count in the wave), i.e. the busy fraction
 * of a one-particle-per-lane schedule.  -1 if that interval has not run yet (or ran in the single-workgroup kernel).
 * The automatic propagate variant uses this figure to decide, per interval, whether workgroup compaction pays. */
int dpomp_get_lane_utilisation(dpomp_handle* h, int obs_index, double* util);
/* Number of allocating HIP calls (hipMalloc, hipHostMalloc, hipEventCreate, hipStreamCreate) the library has made in
 * this process so far -- every such call site goes through one counter.  Test hook: the difference across a call is
 * zero for an entry point that allocates nothing. */
long long dpomp_debug_alloc_count(void);
/* Gillespie events executed by the last pf call (sum over all particles). */
int dpomp_last_event_count(dpomp_handle* h, uint64_t* events);

#ifdef __cplusplus
}
#endif
#endif /* DPOMP_H */
