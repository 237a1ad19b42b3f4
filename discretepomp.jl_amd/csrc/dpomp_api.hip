// dpomp_api.hip -- the C ABI of include/dpomp.h and the host-side step sequencer
// (partial_log_likelihood!, src/hmm_particle_filter.jl:39-76, as a stream of kernel launches
// over a device-resident population; no host synchronisation inside a call).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>
#include <hip/hiprtc.h>

#include "../../include/dpomp.h"
#include "dpomp_embedded_sources.h"
#include "dpomp_kernels.h"

using namespace dpomp;

namespace {

thread_local std::string g_global_error;

// Every allocating HIP call of the library goes through one of these, so that tests can assert that an entry point
// which promises to allocate nothing (dpomp_pf_loglik_static) really does not: dpomp_debug_alloc_count().
std::atomic<long long> g_alloc_calls{0};
template <class T>
hipError_t counted_malloc(T** ptr, size_t bytes) {
    g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
    return hipMalloc((void**)ptr, bytes);
}
template <class T>
hipError_t counted_host_malloc(T** ptr, size_t bytes, unsigned flags) {
    g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
    return hipHostMalloc((void**)ptr, bytes, flags);
}
hipError_t counted_event_create(hipEvent_t* e) {
    g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
    return hipEventCreate(e);
}
hipError_t counted_stream_create(hipStream_t* s, unsigned flags) {
    g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
    return hipStreamCreateWithFlags(s, flags);
}

enum KernelId { KID_INIT = 0, KID_PROPAGATE, KID_WSUM, KID_SCAN_BLOCKS, KID_OFFSETS, KID_GATHER, KID_COPYSEG, KID_FILTER_SMALL, KID_WEIGHTS_TILE, KID_COUNT };
const char* const kKernelNames[KID_COUNT] = {"k_init",    "k_propagate", "k_wsum",          "k_scan_blocks",
                                             "k_offsets", "k_gather",    "k_copy_segments", "k_filter_small",
                                             "k_weights_tile"};

typedef void (*PropFn)(PropArgs);

constexpr int64_t kCompactionMinWaves = 4096;   // launches of at least this many waves run the compaction schedule
// Calls of P <= 256 below this many waves run the whole-filter kernel; above, the filters are laid end to end over the
// compaction kernel's workgroups (measured crossover for 200-particle filters, tools/small_filter_crossover.py: 1000
// filters 0.815 against 0.856 ms, 1500 filters 1.153 against 1.022 ms).
constexpr int64_t kSmallFilterMaxWaves = 4608;
constexpr int kDefaultPropVariant = 32;  // workgroup compaction, first live-lane count after 32 events
// Automatic schedule, per observation interval.  A launch that fills the chip (>= 4096 waves: the issue slots a
// compaction frees are taken by other waves) runs with workgroup compaction unless, the last time this handle ran the
// interval, a one-particle-per-lane schedule kept at least this fraction of the lanes busy (the kernels report that
// figure whichever schedule they ran).  Measured on MI355X (tools/size_sweep.py): compaction wins by 1-4 % at
// utilisations of 0.75-0.87 and by 20-30 % where epidemics die out early (0.3); below 4096 waves the static kernel
// wins by 2-7 %.
constexpr double kCompactionBelowUtil = 0.90;
constexpr double kBoostChainFrac = 0.72;  // head start of the last workgroups: share of a wave's event chain (tools/ab_probe.py)
constexpr int kFusedScanMaxTiles = 2048;  // k_offsets<true>: every workgroup sums <= 64 second-level sums + <= 31 tile sums itself

// Per rate evaluator: the one-particle-per-lane kernel and the queue-refill kernels with 2 and 4
// particles per lane (chosen by how many particles the launch has, see pick_prop).
typedef void (*SimFn)(SimArgs);
typedef void (*SmallFn)(SmallArgs);
struct PropSet {
    PropFn fn1, q2, q4, c16, c32;
    SimFn sim;
    SmallFn small;
    PropFn c32_flat;  // compaction kernel over filters laid end to end (small filters)
};
#define DPOMP_PROPSET(...)                                                                                \
    {k_propagate<__VA_ARGS__>, k_propagate_q<__VA_ARGS__, 2>, k_propagate_q<__VA_ARGS__, 4>,            \
     k_propagate_c<__VA_ARGS__, 16>, k_propagate_c<__VA_ARGS__, 32>, k_simulate_rec<__VA_ARGS__>,       \
     k_filter_small<__VA_ARGS__>, k_propagate_c<__VA_ARGS__, 32, kModeFlat>}

struct PropEntry {
    int builtin, S, E;
    PropSet set;
};

// (rate evaluator, S, E) instances compiled for the catalogue of generate_model
// (src/hmm_examples.jl:171-204); anything else runs the table-driven RGeneric.
const PropEntry kPropTable[] = {
    {DPOMP_BUILTIN_SI, 2, 1, DPOMP_PROPSET(RSi<false>)},
    {DPOMP_BUILTIN_SI_FD, 2, 1, DPOMP_PROPSET(RSi<true>)},
    {DPOMP_BUILTIN_SIR, 3, 2, DPOMP_PROPSET(RSir<3, false>)},
    {DPOMP_BUILTIN_SIR, 2, 2, DPOMP_PROPSET(RSir<2, false>)},
    {DPOMP_BUILTIN_SIR_FD, 3, 2, DPOMP_PROPSET(RSir<3, true>)},
    {DPOMP_BUILTIN_SIR_FD, 2, 2, DPOMP_PROPSET(RSir<2, true>)},
    {DPOMP_BUILTIN_SEI, 3, 2, DPOMP_PROPSET(RSei<false>)},
    {DPOMP_BUILTIN_SEI_FD, 3, 2, DPOMP_PROPSET(RSei<true>)},
    {DPOMP_BUILTIN_SEIR, 4, 3, DPOMP_PROPSET(RSeir<4, false>)},
    {DPOMP_BUILTIN_SEIR, 3, 3, DPOMP_PROPSET(RSeir<3, false>)},
    {DPOMP_BUILTIN_SEIR_FD, 4, 3, DPOMP_PROPSET(RSeir<4, true>)},
    {DPOMP_BUILTIN_SEIR_FD, 3, 3, DPOMP_PROPSET(RSeir<3, true>)},
    {DPOMP_BUILTIN_LOTKA, 2, 3, DPOMP_PROPSET(RLotka)},
    {DPOMP_BUILTIN_ROSSMAC, 4, 6, DPOMP_PROPSET(RRossmac)},
};

PropSet generic_set(int S, int E) {
    const int s = S <= 2 ? 2 : (S <= 4 ? 4 : 8);
    const int e = E <= 2 ? 2 : (E <= 4 ? 4 : 8);
#define DPOMP_G(SS, EE) \
    if (s == SS && e == EE) return PropSet DPOMP_PROPSET(RGeneric<SS, EE>);
    DPOMP_G(2, 2) DPOMP_G(2, 4) DPOMP_G(2, 8) DPOMP_G(4, 2) DPOMP_G(4, 4) DPOMP_G(4, 8) DPOMP_G(8, 2)
    DPOMP_G(8, 4) DPOMP_G(8, 8)
#undef DPOMP_G
    return PropSet{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
}

struct KernelStat {
    int64_t launches = 0;
    double total_ms = 0.0;
};

struct PendingTiming {
    int kid;
    hipEvent_t e0, e1;
};

// Static device memory of dpomp_pf_loglik_static: every working buffer of a filter call (both state buffers,
// log-weights, offspring boundaries, tile sums, control block, parameters) carved from one file-scope array, so that
// entry point calls no allocator.  Each GPU the library runs on holds its own copy of the array at its own address
// until the process ends; host code must never pass the array's NAME to a kernel or a copy (on the host it is a
// stand-in, not a device address) -- only the per-device pointer obtained once with hipGetSymbolAddress.
constexpr size_t kStaticScratchBytes = DPOMP_STATIC_MAX_BYTES;
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_dev[kMaxDevices] = {};  // [device] device address of `scratch` on that GPU

}  // namespace

static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

struct dpomp_handle {
#ifdef DPOMP_TRACE_WG
    unsigned long long* trace_buf = nullptr;
    size_t trace_per = 0;
    int trace_n = 0;
#endif
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    DevModel dm{};
    int T = 0;
    std::vector<double> obs_time;
    std::vector<int64_t> obs_id;
    std::vector<long long> ysum;
    PropSet prop{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool prop_is_generic = false;
    bool big = false;  // more than 8 states / events / parameters or more than 2 factors per rate: run-time compiled evaluator only
    // run-time compiled rate evaluator (dpomp_create_with_source)
    bool rtc = false;
    hipModule_t rtc_module = nullptr;
    hipFunction_t rtc_prop1 = nullptr, rtc_propc = nullptr, rtc_sim = nullptr, rtc_small = nullptr;
    hipFunction_t rtc_q2 = nullptr, rtc_q4 = nullptr, rtc_c16 = nullptr, rtc_c32_flat = nullptr;
    int evaluator = 0;  // DPOMP_EVALUATOR_*
    std::string jit_note;  // why a table model was not compiled (it then runs in the interpreter)
    // observation data on the device (k_filter_small loops over the observations inside the kernel)
    double* d_obs_time = nullptr;
    long long* d_ysum = nullptr;
    int32_t* d_obs_id = nullptr;
    long long* d_obs_val = nullptr;  // [T][V] Observation.val (user observation models)
    int n_obs_values = 0;
    bool user_obs = false;
    bool small_filter = true;  // whole-filter-per-workgroup kernel for P <= 256 (dpomp_set_propagate_variant(h, 1) etc. disable it)
    int force_rpl = 0;  // propagate variant: 0 = auto; 1 = one particle per lane; 2, 4 = LDS queue with that
                        // many particles per lane; 16, 32 = compaction every that many events
    int rs_type = DPOMP_RS_SYSTEMATIC;
    double ess_frac = 0.0;  // > 0: ESS-adaptive resampling (dpomp_set_ess_threshold)
    bool profiling = false;        // brackets on for the current call
    int profiling_every = 0;       // 0 = off, n = bracket the kernels of every n-th filter call
    uint64_t profiled_calls = 0, call_counter = 0;
    double null_bracket_ms = 0.0;
    KernelStat stats[KID_COUNT];
    std::vector<hipEvent_t> event_pool;
    size_t events_used = 0;
    std::vector<PendingTiming> pending;
    uint64_t last_events = 0;
    std::vector<double> lane_util;  // [T] lane utilisation a one-particle-per-lane schedule had / would have had, last call (-1 = unknown)
    std::vector<double> wave_chain; // [T] mean over waves of the longest event chain in the wave, last call (-1 = unknown)
    int resident_wgs = 2048;        // workgroups of the compaction kernel the device holds at once (8 per CU)
    bool no_head_start = false;     // DPOMP_HEAD_START=0: experiment switch
    int tiles_per_wg = 0;           // DPOMP_TILES_PER_WG: experiment switch (k_offsets; 0 = by grid size)
    bool no_flat = false;           // DPOMP_FLAT=0: experiment switch (small filters one per workgroup column)
    double boost_frac = kBoostChainFrac;
    int boost_wgs = 0;              // 0 = half of resident_wgs
    std::string err;
    // scratch for dpomp_pf_loglik
    dpomp_pop* scratch_pop = nullptr;
    // pinned staging of dpomp_pf_loglik_static (allocated at create: that entry point allocates nothing)
    char* static_h_result = nullptr;
    double* static_h_theta = nullptr;
};

struct dpomp_pop {
    int64_t B = 0, P = 0, Pp = 0;
    int nblk = 0, S = 0, K = 0, T = 0;
    int32_t* state[2] = {nullptr, nullptr};
    int cur = 0;
    double* logw = nullptr;
    int32_t* offs = nullptr;
    unsigned long long* wfix = nullptr;    // [B][Pp] fixed-point weights of the current observation
    unsigned long long* blocksum = nullptr;
    unsigned long long* blockprefix = nullptr;
    int32_t* tile_end = nullptr;           // [B][Pp/256] first_parent map of the last systematic resampling
    double* posw = nullptr;                // [B][Pp]   (allocated on first use of resampler 2 / 3)
    double* tile_end_pos = nullptr;        // [B][nblk]
    double* rsu = nullptr;                 // [B]
    // ESS-adaptive mode (allocated on first use)
    unsigned long long* blocksum2 = nullptr;  // [B][nblk][2]
    int32_t* carry[2] = {nullptr, nullptr};   // [B] ping-pong: weights carried into the next weighted observation
    double* lprev[2] = {nullptr, nullptr};    // [B]
    int32_t* noresample = nullptr;            // [B]
    int carry_cur = 0;
    char* ctl = nullptr;                   // control block, see pop_alloc_impl
    size_t ctl_bytes = 0, ctl_result_bytes = 0, sched_offset = 0;
    char* h_result = nullptr;    // pinned host staging: result download and parameter upload of a call are true
    double* h_theta = nullptr;   // asynchronous copies (pageable memory makes each a blocking, staged transfer)
    unsigned long long* segmax = nullptr;  // [T][B]   (inside ctl)
    unsigned long long* wtot = nullptr;    // [B]
    double* ll = nullptr;                  // [B]      (inside ctl)
    uint32_t* status = nullptr;            // [B]      (inside ctl)
    double* theta = nullptr;               // [B][K]
    unsigned long long* events = nullptr;  // [1]      (inside ctl)
    int64_t* ancestors = nullptr;          // [B]
    bool initialised = false;
    int last_weighted_obs = -1;
    bool static_mem = false;  // buffers carved from the library's static device array: nothing to allocate or free
};

namespace {

int fail(dpomp_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_global_error = buf;
    return code;
}

#define HIPCHK(h, expr)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(h, e_ == hipErrorOutOfMemory ? DPOMP_ENOMEM : DPOMP_EDEVICE, "%s: %s", \
                        #expr, hipGetErrorString(e_));                                         \
    } while (0)

// Device scratch of one API call: freed when the call returns, early error returns of HIPCHK included.
struct DeviceScratch {
    std::vector<void*> owned;
    ~DeviceScratch() {
        for (void* p : owned) (void)hipFree(p);
    }
    hipError_t doubles(double** out, size_t n) {
        void* p = nullptr;
        const hipError_t e = counted_malloc(&p, n * sizeof(double));
        if (e == hipSuccess) owned.push_back(p);
        *out = static_cast<double*>(p);
        return e;
    }
};

hipEvent_t next_event(dpomp_handle* h) {
    if (h->events_used == h->event_pool.size()) {
        hipEvent_t e;
        if (counted_event_create(&e) != hipSuccess) return nullptr;
        h->event_pool.push_back(e);
    }
    return h->event_pool[h->events_used++];
}

// Per-kernel timing: when a call is profiled its kernels are launched with hipExtLaunchKernelGGL /
// hipExtModuleLaunchKernel, which stamp a start and a stop event with the kernel's OWN begin and end (what rocprofv3
// reports as the dispatch duration), instead of bracketing the launch with two hipEventRecord markers: those cost
// ~1.5 us each on the GPU timeline and their bracket includes the launch gap (k_gather read 10.4 us in a bracket,
// 8.0 us under rocprofv3).
template <class F, class... Args>
void timed_launch(dpomp_handle* h, int kid, F kernel, dim3 grid, dim3 block, unsigned shmem, hipStream_t st, Args... args) {
    if (h->profiling) {
        hipEvent_t e0 = next_event(h), e1 = next_event(h);
        if (e0 && e1) {
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, st, e0, e1, 0, args...);
            h->pending.push_back({kid, e0, e1});
            return;
        }
    }
    hipLaunchKernelGGL(kernel, grid, block, shmem, st, args...);
}
// Run-time compiled kernels (module functions); `params` as for hipModuleLaunchKernel.
hipError_t timed_module_launch(dpomp_handle* h, int kid, hipFunction_t f, dim3 grid, dim3 block, hipStream_t st, void** params) {
    if (h->profiling) {
        hipEvent_t e0 = next_event(h), e1 = next_event(h);
        if (e0 && e1) {
            const hipError_t e = hipExtModuleLaunchKernel(f, grid.x * block.x, grid.y * block.y, grid.z * block.z, block.x, block.y,
                                                          block.z, 0, st, params, nullptr, e0, e1, 0);
            if (e == hipSuccess) h->pending.push_back({kid, e0, e1});
            return e;
        }
    }
    return hipModuleLaunchKernel(f, grid.x, grid.y, grid.z, block.x, block.y, block.z, 0, st, params, nullptr);
}
// Decides whether this filter call is bracketed (dpomp_set_profiling(h, n): every n-th call).
inline void begin_profiled_call(dpomp_handle* h) {
    h->profiling = h->profiling_every > 0 && (h->call_counter % (uint64_t)h->profiling_every) == 0;
    h->call_counter += 1;
    if (h->profiling) h->profiled_calls += 1;
}

// Resolving a bracket costs a host call per kernel; doing it at the end of every filter call would leave the GPU idle
// meanwhile, so brackets pile up (the calls are synchronous: their events have completed) and are resolved when the
// statistics are read, profiling is switched off, or the pile gets large.
constexpr size_t kMaxPendingBrackets = 8192;
void collect_timings(dpomp_handle* h);
void collect_timings_if_many(dpomp_handle* h) {
    if (h->pending.size() >= kMaxPendingBrackets) collect_timings(h);
}

void collect_timings(dpomp_handle* h) {
    for (const PendingTiming& p : h->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            h->stats[p.kid].launches += 1;
            h->stats[p.kid].total_ms += ms;
        }
    }
    h->pending.clear();
    h->events_used = 0;
}

// A model beyond what the compiled-in evaluators (and DevModel's fixed arrays) hold: it runs on an evaluator generated
// for its size at create time.
bool model_is_big(const dpomp_model_desc* d) {
    if (d->n_states > DPOMP_COMPILED_MAX || d->n_events > DPOMP_COMPILED_MAX || d->n_params > DPOMP_COMPILED_MAX) return true;
    if (d->builtin_id != DPOMP_BUILTIN_USER)
        for (int e = 0; e < d->n_events; ++e)
            if (d->rate[e].n_factors > 2) return true;
    return false;
}

int validate_desc(const dpomp_model_desc* d) {
    if (!d) return fail(nullptr, DPOMP_EINVAL, "descriptor is NULL");
    if (d->abi_version != DPOMP_ABI_VERSION)
        return fail(nullptr, DPOMP_EINVAL, "abi_version %d, library speaks %d", d->abi_version, DPOMP_ABI_VERSION);
    if (d->n_states < 1 || d->n_states > DPOMP_MAX_STATES) return fail(nullptr, DPOMP_EINVAL, "n_states %d out of 1..%d", d->n_states, DPOMP_MAX_STATES);
    if (d->n_events < 1 || d->n_events > DPOMP_MAX_EVENTS) return fail(nullptr, DPOMP_EINVAL, "n_events %d out of 1..%d", d->n_events, DPOMP_MAX_EVENTS);
    if (d->n_params < 1 || d->n_params > DPOMP_MAX_PARAMS) return fail(nullptr, DPOMP_EINVAL, "n_params %d out of 1..%d", d->n_params, DPOMP_MAX_PARAMS);
    if (d->t0_index < 0 || d->t0_index > d->n_params) return fail(nullptr, DPOMP_EINVAL, "t0_index %d out of 0..%d", d->t0_index, d->n_params);
    if (d->n_obs < 1 || !d->obs_time || !d->obs_id) return fail(nullptr, DPOMP_EINVAL, "no observations");
    if (d->n_obs_values < 0 || d->n_obs_values > 32 || (d->n_obs_values > 0 && !d->obs_value))
        return fail(nullptr, DPOMP_EINVAL, "n_obs_values %d invalid", d->n_obs_values);
    if (d->obs.kind != DPOMP_OBS_GAUSSIAN && d->obs.kind != DPOMP_OBS_USER) return fail(nullptr, DPOMP_EINVAL, "unknown observation model kind %d", d->obs.kind);
    if (d->obs.kind == DPOMP_OBS_USER && d->builtin_id != DPOMP_BUILTIN_USER)
        return fail(nullptr, DPOMP_EINVAL, "a user observation model needs a run-time compiled model (DPOMP_BUILTIN_USER + dpomp_create_with_source)");
    if (d->obs.kind == DPOMP_OBS_GAUSSIAN && !(d->obs.sigma > 0.0)) return fail(nullptr, DPOMP_EINVAL, "obs.sigma must be positive");
    if (d->n_states < 32 && (d->obs.state_mask >> d->n_states)) return fail(nullptr, DPOMP_EINVAL, "obs.state_mask names a state >= n_states");
    if (d->flags & ~(DPOMP_FLAG_TABLE_INTERPRET | DPOMP_FLAG_TABLE_COMPILE)) return fail(nullptr, DPOMP_EINVAL, "unknown flags 0x%x", d->flags);
    if ((d->flags & DPOMP_FLAG_TABLE_INTERPRET) && (d->flags & DPOMP_FLAG_TABLE_COMPILE))
        return fail(nullptr, DPOMP_EINVAL, "flags ask for both an interpreted and a compiled rate table");
    if (d->n_obs_values < 32 && (d->obs.value_mask >> d->n_obs_values)) return fail(nullptr, DPOMP_EINVAL, "obs.value_mask names a value column >= n_obs_values");
    for (int s = 0; s < d->n_states; ++s)
        if (d->initial_condition[s] < 0 || d->initial_condition[s] >= ((int64_t)1 << 30))
            return fail(nullptr, DPOMP_EINVAL, "initial_condition[%d] out of 0..2^30", s);
    for (int e = 0; e < d->n_events; ++e) {
        for (int s2 = 0; s2 < d->n_states; ++s2)
            if (d->transition[e][s2] < -127 || d->transition[e][s2] > 127) return fail(nullptr, DPOMP_EINVAL, "transition[%d][%d] out of int8 range", e, s2);
        if (d->builtin_id == DPOMP_BUILTIN_USER) continue;  // rates come from compiled source, not from the table
        const dpomp_rate_term& t = d->rate[e];
        if (t.theta_index < 0 || t.theta_index >= d->n_params) return fail(nullptr, DPOMP_EINVAL, "rate[%d].theta_index out of range", e);
        if (t.form != DPOMP_RATE_LEFT && t.form != DPOMP_RATE_RATIO) return fail(nullptr, DPOMP_EINVAL, "rate[%d].form unknown", e);
        if (t.n_factors < 0 || t.n_factors > DPOMP_MAX_FACTORS) return fail(nullptr, DPOMP_EINVAL, "rate[%d].n_factors out of range", e);
        for (int f = 0; f < t.n_factors; ++f)
            if (t.factor_mask[f] == 0 || (d->n_states < 32 && (t.factor_mask[f] >> d->n_states))) return fail(nullptr, DPOMP_EINVAL, "rate[%d].factor_mask[%d] invalid", e, f);
        if (d->n_states < 32 && (t.denom_mask >> d->n_states)) return fail(nullptr, DPOMP_EINVAL, "rate[%d].denom_mask invalid", e);
    }
    for (int t = 1; t < d->n_obs; ++t)
        if (d->obs_time[t] < d->obs_time[t - 1]) return fail(nullptr, DPOMP_EINVAL, "obs_time must be non-decreasing");
    return DPOMP_OK;
}

void pop_release(dpomp_pop* p) {
    if (!p) return;
    if (p->static_mem) return;  // a view of the static device array on the caller's stack: nothing is owned
    for (int i = 0; i < 2; ++i) (void)hipFree(p->state[i]);
    (void)hipFree(p->logw);
    (void)hipFree(p->offs);
    (void)hipFree(p->wfix);
    (void)hipFree(p->blocksum);
    (void)hipFree(p->blockprefix);
    (void)hipFree(p->tile_end);
    (void)hipFree(p->posw);
    (void)hipFree(p->tile_end_pos);
    (void)hipFree(p->rsu);
    (void)hipFree(p->blocksum2);
    for (int i = 0; i < 2; ++i) {
        (void)hipFree(p->carry[i]);
        (void)hipFree(p->lprev[i]);
    }
    (void)hipFree(p->noresample);
    (void)hipFree(p->ctl);
    (void)hipFree(p->wtot);
    (void)hipFree(p->theta);
    (void)hipFree(p->ancestors);
    (void)hipHostFree(p->h_result);
    (void)hipHostFree(p->h_theta);
    delete p;
}

// Shape of a population and of its control block:
// [ll B f64][events u64][status B u32, padded to 8 B][per-interval counters][segmax T*B u64][second-level weight sums
// T*B*ceil(nblk/32) u64] -> one clear and one result download (of the first four) per call.  Per observation interval: executed events and 64 * sum of per-wave maxima, as partial
// sums on 128-byte lines of their own (dpomp_kernels.h: sched_add).
void pop_set_shape(dpomp_handle* h, dpomp_pop* p, int64_t B, int64_t P) {
    p->B = B;
    p->P = P;
    p->Pp = (P + kTile - 1) / kTile * kTile;
    p->nblk = (int)(p->Pp / kTile);
    p->S = h->dm.S;
    p->K = h->dm.K;
    p->T = h->T;
    const size_t n_ll = (size_t)B * 8, n_ev = 8, n_st = ((size_t)B * 4 + 7) / 8 * 8, n_sm = (size_t)p->T * B * 8;
    const size_t n_sc = (size_t)p->T * kSchedStride * 8;
    p->sched_offset = n_ll + n_ev + n_st;
    p->ctl_result_bytes = n_ll + n_ev + n_st + n_sc;
    p->ctl_bytes = p->ctl_result_bytes + n_sm;
}
void pop_bind_ctl(dpomp_pop* p) {
    const size_t n_ll = (size_t)p->B * 8, n_ev = 8;
    p->ll = reinterpret_cast<double*>(p->ctl);
    p->events = reinterpret_cast<unsigned long long*>(p->ctl + n_ll);
    p->status = reinterpret_cast<uint32_t*>(p->ctl + n_ll + n_ev);
    p->segmax = reinterpret_cast<unsigned long long*>(p->ctl + p->ctl_result_bytes);
}

int check_pop_dims(dpomp_handle* h, int64_t B, int64_t P) {
    if (B < 1 || P < 1 || P >= ((int64_t)1 << 31) - kTile) return fail(h, DPOMP_EINVAL, "B=%lld P=%lld out of range", (long long)B, (long long)P);
    if (B > 65535) return fail(h, DPOMP_EINVAL, "B=%lld exceeds 65535 segments per call", (long long)B);
    return DPOMP_OK;
}

int pop_alloc_impl(dpomp_handle* h, int64_t B, int64_t P, dpomp_pop** out) {
    if (int rc = check_pop_dims(h, B, P)) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    dpomp_pop* p = new dpomp_pop;
    pop_set_shape(h, p, B, P);
    const size_t n_state = (size_t)B * p->S * p->Pp;
#define DPOMP_ALLOC(ptr, bytes)                                                    \
    do {                                                                           \
        hipError_t e_ = counted_malloc((void**)&(ptr), (bytes));                        \
        if (e_ != hipSuccess) {                                                    \
            pop_release(p);                                                        \
            return fail(h, DPOMP_ENOMEM, "hipMalloc(%zu): %s", (size_t)(bytes), hipGetErrorString(e_)); \
        }                                                                          \
    } while (0)
    DPOMP_ALLOC(p->state[0], n_state * sizeof(int32_t));
    DPOMP_ALLOC(p->state[1], n_state * sizeof(int32_t));
    DPOMP_ALLOC(p->logw, (size_t)B * p->Pp * sizeof(double));
    DPOMP_ALLOC(p->offs, (size_t)B * p->Pp * sizeof(int32_t));
    DPOMP_ALLOC(p->wfix, (size_t)B * p->Pp * 8);
    DPOMP_ALLOC(p->blocksum, (size_t)B * p->nblk * 8);
    DPOMP_ALLOC(p->blockprefix, (size_t)B * p->nblk * 8);
    DPOMP_ALLOC(p->tile_end, (size_t)B * (p->Pp / kParentBlock) * 4);
    DPOMP_ALLOC(p->rsu, (size_t)B * 8);
    DPOMP_ALLOC(p->ctl, p->ctl_bytes);
    pop_bind_ctl(p);
    DPOMP_ALLOC(p->wtot, (size_t)B * 8);
    DPOMP_ALLOC(p->theta, (size_t)B * p->K * 8);
    DPOMP_ALLOC(p->ancestors, (size_t)B * 8);
#undef DPOMP_ALLOC
    if (counted_host_malloc((void**)&p->h_result, p->ctl_result_bytes, hipHostMallocDefault) != hipSuccess ||
        counted_host_malloc((void**)&p->h_theta, (size_t)B * p->K * 8, hipHostMallocDefault) != hipSuccess) {
        pop_release(p);
        return fail(h, DPOMP_ENOMEM, "hipHostMalloc of the staging buffers failed");
    }
    // padding lanes [P, Pp) of the state rows are never written by the filter: give them a defined value once
    if (hipMemset(p->state[0], 0, n_state * sizeof(int32_t)) != hipSuccess || hipMemset(p->state[1], 0, n_state * sizeof(int32_t)) != hipSuccess) {
        pop_release(p);
        return fail(h, DPOMP_EDEVICE, "hipMemset of the new population failed");
    }
    *out = p;
    return DPOMP_OK;
}

// The same buffers carved out of one block of device memory starting at `base` (256-byte aligned pieces), for the
// resampler and the ESS mode the handle is set to.  Returns the bytes used; with base == nullptr only the size is
// computed (the pointers written are then meaningless offsets).
size_t pop_carve(dpomp_handle* h, dpomp_pop* p, unsigned char* base) {
    size_t used = 0;
    auto take = [&](size_t bytes) {
        unsigned char* r = reinterpret_cast<unsigned char*>(reinterpret_cast<uintptr_t>(base) + used);
        used += (bytes + 255) / 256 * 256;
        return r;
    };
    const size_t B = (size_t)p->B, n_state = B * p->S * p->Pp;
    p->state[0] = reinterpret_cast<int32_t*>(take(n_state * 4));
    p->state[1] = reinterpret_cast<int32_t*>(take(n_state * 4));
    p->logw = reinterpret_cast<double*>(take(B * p->Pp * 8));
    p->offs = reinterpret_cast<int32_t*>(take(B * p->Pp * 4));
    p->wfix = reinterpret_cast<unsigned long long*>(take(B * p->Pp * 8));
    p->blocksum = reinterpret_cast<unsigned long long*>(take(B * p->nblk * 8));
    p->blockprefix = reinterpret_cast<unsigned long long*>(take(B * p->nblk * 8));
    p->tile_end = reinterpret_cast<int32_t*>(take(B * (p->Pp / kParentBlock) * 4));
    p->rsu = reinterpret_cast<double*>(take(B * 8));
    p->ctl = reinterpret_cast<char*>(take(p->ctl_bytes));
    p->wtot = reinterpret_cast<unsigned long long*>(take(B * 8));
    p->theta = reinterpret_cast<double*>(take(B * p->K * 8));
    if (h->rs_type != DPOMP_RS_SYSTEMATIC) {
        p->posw = reinterpret_cast<double*>(take(B * p->Pp * 8));
        p->tile_end_pos = reinterpret_cast<double*>(take(B * p->nblk * 8));
    }
    if (h->ess_frac > 0.0) {
        p->blocksum2 = reinterpret_cast<unsigned long long*>(take(B * p->nblk * 16));
        for (int i = 0; i < 2; ++i) {
            p->carry[i] = reinterpret_cast<int32_t*>(take(B * 4));
            p->lprev[i] = reinterpret_cast<double*>(take(B * 8));
        }
        p->noresample = reinterpret_cast<int32_t*>(take(B * 4));
    }
    return used;
}

// Lazily allocated groups of a population (resamplers 2/3, ESS mode): allocated into locals and committed together,
// so a failure half way leaves the population as it was.
int pop_ensure_pos(dpomp_handle* h, dpomp_pop* p) {
    if (p->posw) return DPOMP_OK;
    if (p->static_mem) return fail(h, DPOMP_ESTATE, "static population was carved without the stratified/multinomial buffers");
    double *a = nullptr, *b = nullptr;
    if (counted_malloc((void**)&a, (size_t)p->B * p->Pp * 8) != hipSuccess || counted_malloc((void**)&b, (size_t)p->B * p->nblk * 8) != hipSuccess) {
        (void)hipFree(a);
        (void)hipFree(b);
        return fail(h, DPOMP_ENOMEM, "hipMalloc of the scaled cumulative weights failed");
    }
    p->posw = a;
    p->tile_end_pos = b;
    return DPOMP_OK;
}
int pop_ensure_adaptive(dpomp_handle* h, dpomp_pop* p) {
    if (p->blocksum2) return DPOMP_OK;
    if (p->static_mem) return fail(h, DPOMP_ESTATE, "static population was carved without the ESS buffers");
    void* q[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t B = (size_t)p->B;
    const size_t bytes[6] = {B * p->nblk * 16, B * 4, B * 4, B * 8, B * 8, B * 4};
    for (int i = 0; i < 6; ++i)
        if (counted_malloc(&q[i], bytes[i]) != hipSuccess) {
            for (int j = 0; j < 6; ++j) (void)hipFree(q[j]);
            return fail(h, DPOMP_ENOMEM, "hipMalloc of the ESS-adaptive buffers failed");
        }
    p->blocksum2 = static_cast<unsigned long long*>(q[0]);
    p->carry[0] = static_cast<int32_t*>(q[1]);
    p->carry[1] = static_cast<int32_t*>(q[2]);
    p->lprev[0] = static_cast<double*>(q[3]);
    p->lprev[1] = static_cast<double*>(q[4]);
    p->noresample = static_cast<int32_t*>(q[5]);
    return DPOMP_OK;
}

// Clears device memory with a kernel instead of hipMemsetAsync (the call may sit on a stream that is being captured,
// and memset nodes of a captured graph have replayed wrongly).  bytes is a multiple of 4.
hipError_t clear_device(hipStream_t st, void* ptr, size_t bytes) {
    const size_t words = bytes / 4;
    if (!words) return hipSuccess;
    hipLaunchKernelGGL(k_zero, dim3((unsigned)((words + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, static_cast<uint32_t*>(ptr), (int64_t)words);
    return hipGetLastError();
}

hipError_t launch_propagate(dpomp_handle* h, int variant, int mode, dim3 grid, dim3 block, hipStream_t st, PropArgs& a);

// End of a filter call: the first `bytes` of the control block onto the host (asynchronous copy into the pinned staging
// buffer, one synchronisation).  Tried instead and not kept: a last kernel that stores the results straight into the
// pinned buffer and raises a flag the host polls -- 4 us less on a call in which nothing happens, nothing measurable on
// a real filter, slower with many launches in flight.
int finish_call(dpomp_handle* h, dpomp_pop* p, hipStream_t st, size_t bytes) {
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(p->h_result, p->ctl, bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    collect_timings_if_many(h);
    return DPOMP_OK;
}

// The sequencer.  ymin/ymax 1-based as in the reference.
int run_partial(dpomp_handle* h, dpomp_pop* p, const double* theta, int ymin, int ymax, uint64_t seed,
                int64_t seg_base, uint32_t domain, double* out_ll, uint32_t* out_status,
                int32_t* record_states /* device [T][B][S][Pp] or null */) {
    const int T = h->T;
    if (!theta) return fail(h, DPOMP_EINVAL, "theta is NULL");
    if (ymin < 1 || ymax > T || ymin > ymax) return fail(h, DPOMP_EINVAL, "ymin=%d ymax=%d out of 1..%d", ymin, ymax, T);
    if (ymin > 1 && !p->initialised) return fail(h, DPOMP_ESTATE, "ymin=%d on a population that was never initialised (call with ymin=1 or upload first)", ymin);
    if (seg_base < 0 || seg_base + p->B > 0xFFFFFFFFll) return fail(h, DPOMP_EINVAL, "seg_base out of range");
    HIPCHK(h, hipSetDevice(h->device));
    if (p->T != T || p->S != h->dm.S || p->K != h->dm.K)
        return fail(h, DPOMP_EINVAL, "population belongs to another filter (T/S/K = %d/%d/%d, this handle %d/%d/%d)", p->T, p->S, p->K, T, h->dm.S, h->dm.K);
    if (h->rs_type != DPOMP_RS_SYSTEMATIC) {
        if (int rc = pop_ensure_pos(h, p)) return rc;
    }
    hipStream_t st = h->stream;
    const int64_t B = p->B;
    const dim3 block(kBlock);
    const dim3 grid_tiles((unsigned)p->nblk, (unsigned)B);
    if (p->static_mem) h->profiling = false;  // (timing brackets create events on demand)
    else begin_profiled_call(h);

    const bool adaptive = h->ess_frac > 0.0 && domain == DOMAIN_PROPAGATE;
    if (adaptive) {
        const bool fresh = !p->blocksum2;
        if (int rc = pop_ensure_adaptive(h, p)) return rc;
        if (fresh || ymin == 1) HIPCHK(h, clear_device(st, p->carry[p->carry_cur], (size_t)B * 4));
        if (fresh) HIPCHK(h, clear_device(st, p->carry[p->carry_cur ^ 1], (size_t)B * 4));
    }
    {   // first kernel of the call: clear the control block; a small batch's parameters ride in its arguments
        const int n_theta = (int)(B * p->K);
        ThetaInline ti{};
        if (n_theta <= kThetaInline) {
            memcpy(ti.v, theta, (size_t)n_theta * 8);
        } else {
            memcpy(p->h_theta, theta, (size_t)n_theta * 8);
            HIPCHK(h, hipMemcpyAsync(p->theta, p->h_theta, (size_t)n_theta * 8, hipMemcpyHostToDevice, st));
        }
        const int64_t words = (int64_t)(p->ctl_bytes / 4);
        hipLaunchKernelGGL(k_begin, dim3((unsigned)((words + kBlock - 1) / kBlock)), block, 0, st, reinterpret_cast<uint32_t*>(p->ctl),
                           words, p->theta, ti, n_theta <= kThetaInline ? n_theta : 0);
    }

    // A filter that fits one workgroup runs whole in ONE kernel (all observations ymin..ymax, everything between
    // the steps kept in LDS / registers) -- the regime of the reference's defaults (200 particles per filter).
    // (Thousands of such filters in one call fill the chip with the multi-kernel path, whose workgroup compaction then
    // beats the whole-filter kernel's static lanes: 4000 x 200 particles 2.83 against 2.90 ms.)
    if (p->P <= kBlock && h->small_filter && h->force_rpl == 0 && h->rs_type == DPOMP_RS_SYSTEMATIC && !adaptive &&
        domain == DOMAIN_PROPAGATE && !record_states && B * ((p->P + 63) / 64) < kSmallFilterMaxWaves) {
        SmallArgs sa;
        sa.m = h->dm;
        sa.state = p->state[p->cur];
        sa.logw = p->logw;
        sa.offs = p->offs;
        sa.theta = p->theta;
        sa.obs_time = h->d_obs_time;
        sa.ysum = h->d_ysum;
        sa.obs_id = h->d_obs_id;
        sa.yval = h->d_obs_val;
        sa.n_yval = h->n_obs_values;
        sa.ll = p->ll;
        sa.events = p->events;
        sa.status = p->status;
        sa.Pp = p->Pp;
        sa.P = (int32_t)p->P;
        sa.T = T;
        sa.ymin = ymin;
        sa.ymax = ymax;
        sa.sh = weight_shift(p->P);
        sa.seg_base = (uint32_t)seg_base;
        sa.seed = seed;
        {
            if (h->rtc) {
                void* params[] = {&sa};
                HIPCHK(h, timed_module_launch(h, KID_FILTER_SMALL, h->rtc_small, dim3((unsigned)B), block, st, params));
            } else {
                timed_launch(h, KID_FILTER_SMALL, h->prop.small, dim3((unsigned)B), block, 0, st, sa);
            }
        }
        p->initialised = true;
        HIPCHK(h, hipGetLastError());
        if (int rc = finish_call(h, p, st, p->sched_offset)) return rc;  // (the per-interval counters are not used here)
        const char* const result_data = p->h_result;
        if (out_ll) memcpy(out_ll, result_data, (size_t)B * 8);
        memcpy(&h->last_events, result_data + (size_t)B * 8, 8);
        if (out_status) memcpy(out_status, result_data + (size_t)B * 8 + 8, (size_t)B * 4);
        return DPOMP_OK;
    }

    if (ymin == 1) p->initialised = true;  // pop[i,:] .= initial_condition (:41-45) happens inside the first propagation
    double t_prev = (ymin == 1) ? 0.0 : h->obs_time[ymin - 2];  // :45-47

    // Schedule of the propagation towards observation k (0-based).  Default: workgroup compaction once the launch holds
    // at least half of what the chip keeps resident (>= 4096 waves: with the head start of the last workgroups it beats
    // the plain kernel by 5-9 % from 2^18 particles on, tools/size_probe.py) unless the interval kept the lanes busy last
    // time; below that the plain kernel wins (the issue slots a compaction frees have no other wave to go to).
    // lane_util = -1 (not measured yet) -> compaction.
    // (force_rpl 33 = the compaction schedule with the end-to-end layout of small filters forced on: tests, experiments)
    auto pick_variant = [&](int k) {
        if (h->big) return 1;  // (the only multi-kernel schedule built for models beyond 8 states / events)
        if (h->force_rpl) return h->force_rpl == 33 ? 32 : h->force_rpl;
        return (B * ((p->P + 63) / 64) >= kCompactionMinWaves && h->lane_util[k] < kCompactionBelowUtil) ? kDefaultPropVariant : 1;
    };
    // Filters of a few hundred particles are laid end to end over the workgroups of the compaction kernel (kModeFlat):
    // one filter per workgroup column leaves the last wave of every 200-particle filter with 8 of 64 lanes.
    const bool use_flat = p->P <= kFlatMaxP && B * p->P < ((int64_t)1 << 31) &&
                          (h->force_rpl == 33 || (h->force_rpl == 0 && !h->no_flat && p->P % kBlock != 0));
    // A systematic resampling that is followed by another propagation in this call is not carried out by a gather pass:
    // the next propagation reads every slot's state from its ancestor (PropArgs::parents) and writes to the other buffer.
    bool pending_gather = false;

    for (int obs_i = ymin; obs_i <= ymax; ++obs_i) {  // :54
        const int k = obs_i - 1;
        const bool weighted = h->obs_id[k] > 0 && domain == DOMAIN_PROPAGATE;  // :58
        PropArgs a;
        a.m = h->dm;
        a.state = p->state[p->cur];
        a.parents = nullptr;
        a.offs = nullptr;
        a.first_parent = nullptr;
        a.noresample = nullptr;
        if (pending_gather) {
            a.parents = p->state[p->cur];
            a.state = p->state[p->cur ^ 1];
            a.offs = p->offs;
            a.first_parent = p->tile_end;
            a.noresample = adaptive ? p->noresample : nullptr;
        }
        a.logw = p->logw;
        a.theta = p->theta;
        a.segmax = p->segmax + (size_t)k * B;
        a.events = p->events;
        a.status = p->status;
        a.Pp = p->Pp;
        a.P = (int32_t)p->P;
        a.obs_k = (uint32_t)k;
        a.seg_base = (uint32_t)seg_base;
        a.domain = domain;
        a.seed = seed;
        a.t_prev = t_prev;
        a.t_max = h->obs_time[k];
        a.ysum = h->ysum[k];
        a.use_theta_t0 = (obs_i == 1 && h->dm.t0_index > 0) ? 1 : 0;
        a.from_x0 = obs_i == 1 ? 1 : 0;
        a.want_weight = weighted ? 1 : 0;
        a.carry_in = adaptive ? p->carry[p->carry_cur] : nullptr;
        a.yval = h->d_obs_val + (size_t)k * h->n_obs_values;
        a.n_yval = h->n_obs_values;
        a.sched = reinterpret_cast<unsigned long long*>(p->ctl + p->sched_offset) + kSchedStride * (size_t)k;
        a.B = (int32_t)B;
        {
            const int rpl = pick_variant(k);
            const int mode = (rpl == 32 && use_flat) ? kModeFlat : kModeBlock;
            const int64_t chunk = (int64_t)kBlock * ((rpl == 2 || rpl == 4) ? rpl : 1);
            // Head start of the compaction kernel: the last half round of workgroups, for kBoostChainFrac of the event
            // chain a wave had in this interval at the previous call (none before anything has been measured).
            a.boost_events = 0;
            a.boost_from = 0;
            const int64_t n_wg = mode == kModeFlat ? (B * p->P + kBlock - 1) / kBlock : (p->P + kBlock - 1) / kBlock * B;
            if ((rpl == 16 || rpl == 32) && h->wave_chain[k] > 0.0 && !h->no_head_start) {
                a.boost_from = (int32_t)std::max<int64_t>(0, n_wg - (h->boost_wgs ? h->boost_wgs : h->resident_wgs / 2));
                a.boost_events = (int32_t)std::min(1.0e9, std::max(1.0, h->boost_frac * h->wave_chain[k]));
            }
#ifdef DPOMP_TRACE_WG
            {
                static unsigned long long* tr_buf = nullptr;
                const size_t per = (size_t)((p->P + kBlock - 1) / kBlock) * B * 8;
                if (!tr_buf) counted_malloc((void**)&tr_buf, per * 8 * 64);
                a.trace = (k < 64) ? tr_buf + per * k : nullptr;
                h->trace_buf = tr_buf;
                h->trace_per = per;
                h->trace_n = k + 1;
            }
#endif
            // iterate_particles! :56
            const dim3 grid = mode == kModeFlat ? dim3((unsigned)n_wg, 1u) : dim3((unsigned)((p->P + chunk - 1) / chunk), (unsigned)B);
            HIPCHK(h, launch_propagate(h, rpl, mode, grid, block, st, a));
            if (pending_gather) {
                p->cur ^= 1;
                pending_gather = false;
            }
        }
        if (record_states)
            HIPCHK(h, hipMemcpyAsync(record_states + (size_t)k * B * p->S * p->Pp, p->state[p->cur],
                                     (size_t)B * p->S * p->Pp * 4, hipMemcpyDeviceToDevice, st));
        if (weighted) {
            p->last_weighted_obs = k;
            WeightArgs w;
            w.logw = p->logw;
            w.segmax = p->segmax + (size_t)k * B;
            w.blocksum = p->blocksum;
            w.blockprefix = p->blockprefix;
            w.wtot = p->wtot;
            w.ll = p->ll;
            w.status = p->status;
            w.offs = p->offs;
            w.wfix = p->wfix;
            w.tile_end = p->tile_end;
            w.posw = p->posw;
            w.tile_end_pos = p->tile_end_pos;
            w.rsu = p->rsu;
            w.ess_frac = adaptive ? h->ess_frac : 0.0;
            w.blocksum2 = p->blocksum2;
            w.carry_in = adaptive ? p->carry[p->carry_cur] : nullptr;
            w.lprev_in = adaptive ? p->lprev[p->carry_cur] : nullptr;
            w.carry_out = adaptive ? p->carry[p->carry_cur ^ 1] : nullptr;
            w.lprev_out = adaptive ? p->lprev[p->carry_cur ^ 1] : nullptr;
            w.noresample = adaptive ? p->noresample : nullptr;
            w.last_obs = obs_i == T ? 1 : 0;
            if (adaptive) p->carry_cur ^= 1;  // the kernels below write the new carry state, the next observation reads it
            w.rs_type = h->rs_type;
            w.Pp = p->Pp;
            w.P = (int32_t)p->P;
            w.nblk = p->nblk;
            w.sh = weight_shift(p->P);
            // k_offsets: 4 tiles to a workgroup once that still leaves every CU 8 workgroups (what a workgroup needs
            // once per launch is then paid once per 4 tiles); DPOMP_TILES_PER_WG overrides
            w.tiles_per_wg = h->tiles_per_wg > 0 ? h->tiles_per_wg : (p->nblk >= 4 && (int64_t)B * p->nblk >= 4 * (int64_t)h->resident_wgs ? 4 : 1);
            w.obs_k = (uint32_t)k;
            w.seg_base = (uint32_t)seg_base;
            w.seed = seed;
            const bool resample = obs_i < T;  // :62  (the ESS criterion is commented out in the reference, :63-68)
            // a filter of at most one tile: weights, their sum, the :60 update and the offspring boundaries in ONE launch
            const bool one_tile = p->nblk == 1 && !adaptive && h->rs_type == DPOMP_RS_SYSTEMATIC;
            if (one_tile) {  // (1 / 2 / 4 particles per thread: every thread of the workgroup has one to weigh from 256 particles down)
                auto kernel = p->P <= kBlock ? k_weights_tile<1> : p->P <= 2 * kBlock ? k_weights_tile<2> : k_weights_tile<4>;
                timed_launch(h, KID_WEIGHTS_TILE, kernel, dim3((unsigned)B), block, 0, st, w, resample ? 1 : 0);
            }
            else timed_launch(h, KID_WSUM, k_wsum, grid_tiles, block, 0, st, w);
            const bool fused = resample && p->nblk <= kFusedScanMaxTiles;
            if (!fused && !one_tile) {
                // output += log(cum_weight[end]/P)  :60
                timed_launch(h, KID_SCAN_BLOCKS, k_scan_blocks, dim3((unsigned)B), block, 0, st, w);
            }
            if (resample) {
                if (!one_tile) {
                    // fused: also the tile-sum scan and the :60 update
                    const dim3 grid_off((unsigned)((p->nblk + w.tiles_per_wg - 1) / w.tiles_per_wg), (unsigned)B);
                    if (fused) timed_launch(h, KID_OFFSETS, k_offsets<true>, grid_off, block, 0, st, w);
                    else timed_launch(h, KID_OFFSETS, k_offsets<false>, grid_off, block, 0, st, w);
                }
                if (h->rs_type != DPOMP_RS_SYSTEMATIC) {
                    GatherPosArgs gp;
                    gp.src = p->state[p->cur];
                    gp.dst = p->state[p->cur ^ 1];
                    gp.posw = p->posw;
                    gp.tile_end_pos = p->tile_end_pos;
                    gp.Pp = p->Pp;
                    gp.P = (int32_t)p->P;
                    gp.S = p->S;
                    gp.nblk = p->nblk;
                    gp.rs_type = h->rs_type;
                    gp.noresample = adaptive ? p->noresample : nullptr;
                    gp.obs_k = (uint32_t)k;
                    gp.seg_base = (uint32_t)seg_base;
                    gp.seed = seed;
                    timed_launch(h, KID_GATHER, k_gather_pos, grid_tiles, block,
                                 (unsigned)(p->nblk <= kGatherPosStageTiles ? (size_t)p->nblk * 8 : 0), st, gp);
                    p->cur ^= 1;
                    t_prev = h->obs_time[k];
                    continue;
                }
                // old_p .= pop; fn_rs(pop, old_p, cum_weight)  :66-67 -- inside the next propagation of this call if there is
                // one (the LDS-queue schedules load whole chunks and keep the separate pass)
                if (obs_i < ymax) {
                    const int next = pick_variant(k + 1);
                    if (next != 2 && next != 4) {
                        pending_gather = true;
                        t_prev = h->obs_time[k];
                        continue;
                    }
                }
                GatherArgs g;
                g.src = p->state[p->cur];
                g.dst = p->state[p->cur ^ 1];
                g.offs = p->offs;
                g.first_parent = p->tile_end;
                g.noresample = adaptive ? p->noresample : nullptr;
                g.Pp = p->Pp;
                g.P = (int32_t)p->P;
                g.S = p->S;
                g.nblk = p->nblk;
                {
                    // old_p .= pop; fn_rs(pop, old_p, cum_weight)  :66-67
                    timed_launch(h, KID_GATHER, k_gather, grid_tiles, block, 0, st, g);
                }
                p->cur ^= 1;
            }
        }
        t_prev = h->obs_time[k];  // :72
    }
    if (int rc = finish_call(h, p, st, p->ctl_result_bytes)) return rc;
#ifdef DPOMP_TRACE_WG
    if (const char* f = getenv("DPOMP_TRACE_WG_FILE")) {
        std::vector<unsigned long long> host(h->trace_per * h->trace_n);
        hipMemcpy(host.data(), h->trace_buf, host.size() * 8, hipMemcpyDeviceToHost);
        if (FILE* fp = fopen(f, "wb")) {
            const unsigned long long hdr[2] = {(unsigned long long)h->trace_n, (unsigned long long)h->trace_per};
            fwrite(hdr, 8, 2, fp);
            fwrite(host.data(), 8, host.size(), fp);
            fclose(fp);
        }
    }
#endif
    const char* const result_data = p->h_result;
    if (out_ll) memcpy(out_ll, result_data, (size_t)B * 8);
    memcpy(&h->last_events, result_data + (size_t)B * 8, 8);
    if (out_status) memcpy(out_status, result_data + (size_t)B * 8 + 8, (size_t)B * 4);
    {
        const unsigned long long* sc = reinterpret_cast<const unsigned long long*>(result_data + p->sched_offset);
        for (int k = ymin - 1; k < ymax; ++k) {
            unsigned long long ev = 0, slots = 0;
            for (int s2 = 0; s2 < kSchedSlots; ++s2) {
                ev += sc[kSchedStride * k + s2 * kSchedLine];
                slots += sc[kSchedStride * k + (kSchedSlots + s2) * kSchedLine];
            }
            h->last_events += ev;  // the propagate kernels count events per interval
            if (slots && domain == DOMAIN_PROPAGATE) {
                h->lane_util[k] = (double)ev / (double)slots;
                h->wave_chain[k] = (double)slots / (64.0 * (double)(B * ((p->P + 63) / 64)));
            }
        }
    }
    return DPOMP_OK;
}

// ---- hipRTC: user rate functions --------------------------------------------------------------------------
struct RtcApi {
    void* lib = nullptr;
    hiprtcResult (*create)(hiprtcProgram*, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
    hiprtcResult (*add_name)(hiprtcProgram, const char*) = nullptr;
    hiprtcResult (*compile)(hiprtcProgram, int, const char* const*) = nullptr;
    hiprtcResult (*log_size)(hiprtcProgram, size_t*) = nullptr;
    hiprtcResult (*get_log)(hiprtcProgram, char*) = nullptr;
    hiprtcResult (*lowered)(hiprtcProgram, const char*, const char**) = nullptr;
    hiprtcResult (*code_size)(hiprtcProgram, size_t*) = nullptr;
    hiprtcResult (*get_code)(hiprtcProgram, char*) = nullptr;
    hiprtcResult (*destroy)(hiprtcProgram*) = nullptr;
};

RtcApi* rtc_api() {  // loaded on first use only: the library itself does not depend on hipRTC
    static RtcApi api;
    if (api.lib) return &api;
    void* lib = nullptr;
    for (const char* name : {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"})
        if ((lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    if (!lib) return nullptr;
#define DPOMP_SYM(field, sym) *(void**)(&api.field) = dlsym(lib, sym)
    DPOMP_SYM(create, "hiprtcCreateProgram");
    DPOMP_SYM(add_name, "hiprtcAddNameExpression");
    DPOMP_SYM(compile, "hiprtcCompileProgram");
    DPOMP_SYM(log_size, "hiprtcGetProgramLogSize");
    DPOMP_SYM(get_log, "hiprtcGetProgramLog");
    DPOMP_SYM(lowered, "hiprtcGetLoweredName");
    DPOMP_SYM(code_size, "hiprtcGetCodeSize");
    DPOMP_SYM(get_code, "hiprtcGetCode");
    DPOMP_SYM(destroy, "hiprtcDestroyProgram");
#undef DPOMP_SYM
    if (!api.create || !api.compile || !api.get_code || !api.lowered) return nullptr;
    api.lib = lib;
    return &api;
}

constexpr int kRtcKernels = 8;
const char* const kRtcNames[kRtcKernels] = {"dpomp::k_propagate<dpomp::RUser>",       "dpomp::k_simulate_rec<dpomp::RUser>",
                                            "dpomp::k_filter_small<dpomp::RUser>",    "dpomp::k_propagate_c<dpomp::RUser, 32>",
                                            "dpomp::k_propagate_q<dpomp::RUser, 2>",  "dpomp::k_propagate_q<dpomp::RUser, 4>",
                                            "dpomp::k_propagate_c<dpomp::RUser, 16>",
                                            "dpomp::k_propagate_c<dpomp::RUser, 32, dpomp::kModeFlat>"};
constexpr int kRtcKernelsBig = 3;  // the leading entries: what a model beyond 8 states / events is built with
const char* const kRtcOptions[] = {"--offload-arch=gfx950", "-ffp-contract=off", "-O3", "-std=c++17",
                                   "-mllvm", "-amdgpu-sched-strategy=iterative-ilp",
                                   "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"};  // as the library itself is built
constexpr int kRtcOptionCount = (int)(sizeof kRtcOptions / sizeof kRtcOptions[0]);

// ---- cache of compiled models -------------------------------------------------------------------------------
// Compiling the kernels with a rate function takes seconds; the result depends only on the text that goes into the
// compiler.  Code objects are therefore kept under a 128-bit hash of (kernel headers, wrapper, rate source, options,
// HIP runtime version): in this process, and as files in $DPOMP_CACHE_DIR (default ~/.cache/dpomp; "off" disables
// the files).  A cache file = magic, the lowered kernel names, the code object; written to a temporary name and
// renamed, so a reader never sees half a file; anything unreadable or of the wrong shape is ignored and recompiled.
struct RtcBinary {
    std::vector<char> code;
    std::string lowered[kRtcKernels];
};
std::mutex g_rtc_cache_mutex;
std::map<std::string, RtcBinary> g_rtc_cache;
int g_generic_jit = -1;  // dpomp_set_generic_jit: -1 = environment / default (on), 0 = off, 1 = on

std::string hash_hex(const std::string& text) {
    uint64_t a = 0xcbf29ce484222325ull, b = 0x84222325cbf29ce4ull;
    for (unsigned char c : text) {
        a = (a ^ c) * 0x100000001b3ull;
        b = (b ^ (c + 0x9e)) * 0x100000001b3ull;
        b ^= b >> 29;
    }
    char buf[40];
    snprintf(buf, sizeof buf, "%016llx%016llx", (unsigned long long)a, (unsigned long long)b);
    return buf;
}

std::string cache_dir() {
    const char* d = getenv("DPOMP_CACHE_DIR");
    if (d && !strcmp(d, "off")) return "";
    if (d && *d) return d;
    const char* home = getenv("HOME");
    if (!home || !*home) return "";
    return std::string(home) + "/.cache/dpomp";
}

// The cache holds GPU code objects that dpomp_create loads: the directory must belong to the caller and be writable
// by nobody else, or another local user could plant a code object under the name a model hashes to.  Missing
// directories are created private (0700); a directory that fails the check is not used (models are compiled afresh).
bool cache_dir_is_private(const std::string& dir) {
    struct stat st;
    if (stat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return false;
    return st.st_uid == geteuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}
bool make_private_dirs(const std::string& path) {
    for (size_t i = 1; i <= path.size(); ++i)
        if (i == path.size() || path[i] == '/') (void)mkdir(path.substr(0, i).c_str(), i == path.size() ? 0700 : 0755);
    return cache_dir_is_private(path);
}

bool cache_file_read(const std::string& file, RtcBinary& out) {
    FILE* f = fopen(file.c_str(), "rb");
    if (!f) return false;
    bool ok = false;
    char magic[8];
    uint32_t n = 0;
    if (fread(magic, 1, 8, f) == 8 && !memcmp(magic, "DPOMPRT5", 8) && fread(&n, 4, 1, f) == 1 && n == (uint32_t)kRtcKernels) {
        ok = true;
        for (int i = 0; ok && i < kRtcKernels; ++i) {
            uint32_t len = 0;
            ok = fread(&len, 4, 1, f) == 1 && len > 0 && len < 4096;
            if (ok) {
                out.lowered[i].resize(len);
                ok = fread(&out.lowered[i][0], 1, len, f) == len;
            }
        }
        uint64_t cs = 0;
        char sum[32];
        ok = ok && fread(&cs, 8, 1, f) == 1 && cs > 0 && cs < ((uint64_t)1 << 30) && fread(sum, 1, 32, f) == 32;
        if (ok) {
            out.code.resize((size_t)cs);
            ok = fread(out.code.data(), 1, (size_t)cs, f) == (size_t)cs && fgetc(f) == EOF;
            // (a checksum of the code object: a truncated or altered file is recompiled, not loaded)
            ok = ok && !memcmp(sum, hash_hex(std::string(out.code.data(), out.code.size())).c_str(), 32);
        }
    }
    fclose(f);
    return ok;
}

void cache_file_write(const std::string& dir, const std::string& file, const RtcBinary& b) {
    if (!make_private_dirs(dir)) return;
    char tmp[64];
    snprintf(tmp, sizeof tmp, ".tmp.%d.%p", (int)getpid(), (const void*)&b);
    const std::string tmpname = file + tmp;
    FILE* f = fopen(tmpname.c_str(), "wb");
    if (!f) return;
    const uint32_t n = kRtcKernels;
    bool ok = fwrite("DPOMPRT5", 1, 8, f) == 8 && fwrite(&n, 4, 1, f) == 1;
    for (int i = 0; ok && i < kRtcKernels; ++i) {
        const uint32_t len = (uint32_t)b.lowered[i].size();
        ok = fwrite(&len, 4, 1, f) == 1 && fwrite(b.lowered[i].data(), 1, len, f) == len;
    }
    const uint64_t cs = b.code.size();
    const std::string sum = hash_hex(std::string(b.code.data(), b.code.size()));
    ok = ok && fwrite(&cs, 8, 1, f) == 1 && fwrite(sum.data(), 1, 32, f) == 32 && fwrite(b.code.data(), 1, b.code.size(), f) == b.code.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmpname.c_str(), file.c_str()) != 0) (void)remove(tmpname.c_str());
}

// Compiles kernels + rate source (or finds the result in the cache).  On success `bin` holds the gfx950 code object
// and the mangled kernel names; `from_cache` tells which.
int rtc_compile(const dpomp_model_desc* d, const char* rate_source, RtcBinary& bin, std::string& log, bool* from_cache = nullptr) {
    if (from_cache) *from_cache = false;
    char head[512];
    snprintf(head, sizeof head, "#include \"dpomp_kernels.h\"\nnamespace dpomp {\n#line 1 \"rate_source\"\n");
    const bool user_obs = d->obs.kind == DPOMP_OBS_USER;
    const bool big = model_is_big(d);
    char tail[1536];
    snprintf(tail, sizeof tail,
             "\nstruct RUser : RBase<%d, %d> {\n  static constexpr bool kUserObs = %s;\n"
             "  static __device__ __forceinline__ void rates(const DevModel&, const double* th, "
             "const int (&x)[%d], double (&r)[%d]) { dpomp_user_rates(th, x, r); }\n%s",
             d->n_states, d->n_events, user_obs ? "true" : "false", d->n_states, d->n_events,
             user_obs ? "  static __device__ __forceinline__ double user_obs(const double* th, const int* x, const long long* y, int ny) "
                        "{ return dpomp_user_obs(th, x, y, ny); }\n"
                      : "");
    std::string consts;
    if (big) {  // initial condition and transition rows as constants of the evaluator (they do not fit DevModel)
        consts += "  static constexpr bool kBig = true;\n  static __device__ __forceinline__ int x0(int s) {\n    switch (s) {\n";
        for (int i = 0; i < d->n_states; ++i)
            consts += "      case " + std::to_string(i) + ": return " + std::to_string((long long)d->initial_condition[i]) + ";\n";
        consts += "    }\n    return 0;\n  }\n";
        consts += "  static __device__ __forceinline__ void apply_event(int e, int (&x)[" + std::to_string(d->n_states) + "]) {\n    switch (e) {\n";
        for (int e = 0; e < d->n_events; ++e) {
            consts += "      case " + std::to_string(e) + ":";
            for (int i = 0; i < d->n_states; ++i)
                if (d->transition[e][i]) consts += " x[" + std::to_string(i) + "] += " + std::to_string(d->transition[e][i]) + ";";
            consts += " break;\n";
        }
        consts += "    }\n  }\n";
    }
    const std::string src = std::string(head) + rate_source + tail + consts + "};\n}  // namespace dpomp\n";
    // a big model runs the one-particle-per-lane kernel, the whole-filter kernel and the simulator; the compaction and
    // LDS-queue schedules move contexts through fixed 32-byte LDS records and are built for small models only
    const int n_names = big ? kRtcKernelsBig : kRtcKernels;

    std::string keytext = src;
    for (const char* o : kRtcOptions) keytext += std::string("\x01") + o;
    for (int i = 0; i < n_names; ++i) keytext += std::string("\x02") + kRtcNames[i];
    int rt_version = 0;
    (void)hipRuntimeGetVersion(&rt_version);
    keytext += "\x03" + std::to_string(rt_version) + "\x03" + kSrcKernels + "\x03" + kSrcNumerics + "\x03" + kSrcLogTable;
    const std::string key = hash_hex(keytext);
    const std::string dir = cache_dir();
    const std::string file = dir.empty() ? "" : dir + "/" + key + ".gfx950.dpomprtc";
    {
        std::lock_guard<std::mutex> lock(g_rtc_cache_mutex);
        auto it = g_rtc_cache.find(key);
        if (it != g_rtc_cache.end()) {
            bin = it->second;
            if (from_cache) *from_cache = true;
            return DPOMP_OK;
        }
    }
    if (!file.empty() && cache_dir_is_private(dir) && cache_file_read(file, bin)) {
        std::lock_guard<std::mutex> lock(g_rtc_cache_mutex);
        g_rtc_cache[key] = bin;
        if (from_cache) *from_cache = true;
        return DPOMP_OK;
    }

    RtcApi* api = rtc_api();
    if (!api) {
        log = "hipRTC (libhiprtc.so) could not be loaded";
        return DPOMP_EDEVICE;
    }
    const char* headers[3] = {kSrcKernels, kSrcNumerics, kSrcLogTable};
    const char* names[3] = {"dpomp_kernels.h", "dpomp_numerics.h", "dpomp_log_table.h"};
    hiprtcProgram prog;
    if (api->create(&prog, src.c_str(), "dpomp_user_model.hip", 3, headers, names) != HIPRTC_SUCCESS) {
        log = "hiprtcCreateProgram failed";
        return DPOMP_EDEVICE;
    }
    for (int i = 0; i < n_names; ++i) api->add_name(prog, kRtcNames[i]);
    const hiprtcResult r = api->compile(prog, kRtcOptionCount, kRtcOptions);
    size_t ls = 0;
    if (api->log_size && api->log_size(prog, &ls) == HIPRTC_SUCCESS && ls > 1) {
        log.resize(ls);
        api->get_log(prog, &log[0]);
    }
    int rc = DPOMP_OK;
    if (r != HIPRTC_SUCCESS) {
        if (log.empty()) log = "hiprtcCompileProgram failed";
        rc = DPOMP_EINVAL;
    } else {
        for (int i = 0; i < kRtcKernels; ++i) {
            const char* ln = nullptr;
            if (i >= n_names) bin.lowered[i] = "-";  // (not built for this model)
            else if (api->lowered(prog, kRtcNames[i], &ln) != HIPRTC_SUCCESS || !ln) rc = DPOMP_EDEVICE;
            else bin.lowered[i] = ln;
        }
        size_t cs = 0;
        if (rc == DPOMP_OK && api->code_size(prog, &cs) == HIPRTC_SUCCESS && cs) {
            bin.code.resize(cs);
            if (api->get_code(prog, bin.code.data()) != HIPRTC_SUCCESS) rc = DPOMP_EDEVICE;
        } else if (rc == DPOMP_OK) {
            rc = DPOMP_EDEVICE;
        }
    }
    api->destroy(&prog);
    if (rc == DPOMP_OK) {
        if (!file.empty()) cache_file_write(dir, file, bin);
        std::lock_guard<std::mutex> lock(g_rtc_cache_mutex);
        g_rtc_cache[key] = bin;
    }
    return rc;
}

// Loads a compiled model into the handle (module + the seven kernels).
int rtc_load(dpomp_handle* h, const RtcBinary& bin, std::string& why) {
    hipError_t me = hipModuleLoadData(&h->rtc_module, bin.code.data());
    hipFunction_t* fns[kRtcKernels] = {&h->rtc_prop1, &h->rtc_sim, &h->rtc_small, &h->rtc_propc, &h->rtc_q2, &h->rtc_q4, &h->rtc_c16,
                                       &h->rtc_c32_flat};
    for (int i = 0; me == hipSuccess && i < kRtcKernels; ++i)
        if (bin.lowered[i] != "-") me = hipModuleGetFunction(fns[i], h->rtc_module, bin.lowered[i].c_str());
    if (me != hipSuccess) {
        why = hipGetErrorString(me);
        if (h->rtc_module) (void)hipModuleUnload(h->rtc_module);
        h->rtc_module = nullptr;
        return DPOMP_EDEVICE;
    }
    h->rtc = true;
    return DPOMP_OK;
}

// The rate table of a descriptor as device source: the arithmetic of RGeneric::rates (dpomp_kernels.h), operation for
// operation, with everything the interpreter decides at run time (which parameter, which components, which form)
// decided by the text -- so a table model runs at the speed of a compiled-in evaluator with the same bits.
std::string rate_table_source(const dpomp_model_desc* d) {
    auto msum = [&](uint32_t mask) {
        std::string e;
        for (int i = 0; i < d->n_states; ++i)
            if (mask & (1u << i)) e += (e.empty() ? "x[" : " + x[") + std::to_string(i) + "]";
        return "(" + e + ")";
    };
    std::string s = "__device__ __forceinline__ void dpomp_user_rates(const double* theta, const int* x, double* r) {\n";
    for (int e = 0; e < d->n_events; ++e) {
        const dpomp_rate_term& t = d->rate[e];
        const std::string th = "theta[" + std::to_string(t.theta_index) + "]", re = "r[" + std::to_string(e) + "]";
        if (t.form == DPOMP_RATE_LEFT) {
            s += "    { double v = " + th + ";";
            for (int f = 0; f < t.n_factors; ++f) s += " v = v * (double)" + msum(t.factor_mask[f]) + ";";  // (any number of factors)
            if (t.denom_mask) s += " v = v / (double)" + msum(t.denom_mask) + ";";
            s += " " + re + " = v; }\n";
        } else {
            s += "    { long long p = 1;";
            for (int f = 0; f < t.n_factors; ++f) s += " p *= (long long)" + msum(t.factor_mask[f]) + ";";
            s += " double q = (double)p;";
            if (t.denom_mask) s += " q = q / (double)" + msum(t.denom_mask) + ";";
            s += " " + re + " = " + th + " * q; }\n";
        }
    }
    return s + "}\n";
}

bool generic_jit_enabled(const dpomp_model_desc* d) {
    if (d->flags & DPOMP_FLAG_TABLE_INTERPRET) return false;
    if (d->flags & DPOMP_FLAG_TABLE_COMPILE) return true;
    if (g_generic_jit >= 0) return g_generic_jit != 0;
    const char* env = getenv("DPOMP_GENERIC_JIT");
    return !(env && !strcmp(env, "0"));
}

// One launch site for the propagation kernel: compiled-in template instance or run-time compiled module function.
// (mode: kModeBlock / kModeFlat of the compaction kernel, variant 32 only)
hipError_t launch_propagate(dpomp_handle* h, int variant, int mode, dim3 grid, dim3 block, hipStream_t st, PropArgs& a) {
    if (h->rtc) {
        void* params[] = {&a};
        hipFunction_t f = variant == 4 ? h->rtc_q4 : variant == 2 ? h->rtc_q2 : variant == 16 ? h->rtc_c16 : variant == 32 ? h->rtc_propc : h->rtc_prop1;
        if (variant == 32 && mode == kModeFlat) f = h->rtc_c32_flat;
        return timed_module_launch(h, KID_PROPAGATE, f, grid, block, st, params);
    }
    PropFn fn = variant == 4 ? h->prop.q4 : variant == 2 ? h->prop.q2 : variant == 16 ? h->prop.c16 : variant == 32 ? h->prop.c32 : h->prop.fn1;
    if (variant == 32 && mode == kModeFlat) fn = h->prop.c32_flat;
    timed_launch(h, KID_PROPAGATE, fn, grid, block, 0, st, a);
    return hipGetLastError();
}

int create_impl(const dpomp_model_desc* d, int device, const char* rate_source, dpomp_handle** out);

}  // namespace

extern "C" {

int dpomp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* dpomp_version(void) { return "dpomp-hip 0.2 (gfx950, abi 2)"; }

const char* dpomp_last_global_error(void) { return g_global_error.c_str(); }

int dpomp_create(const dpomp_model_desc* d, int device, dpomp_handle** out) {
    if (d && d->builtin_id == DPOMP_BUILTIN_USER)
        return fail(nullptr, DPOMP_EINVAL, "builtin_id DPOMP_BUILTIN_USER needs dpomp_create_with_source");
    return create_impl(d, device, nullptr, out);
}

int dpomp_create_with_source(const dpomp_model_desc* d, int device, const char* rate_source, dpomp_handle** out) {
    if (!rate_source) return fail(nullptr, DPOMP_EINVAL, "rate_source is NULL");
    if (d && d->builtin_id != DPOMP_BUILTIN_USER)
        return fail(nullptr, DPOMP_EINVAL, "dpomp_create_with_source needs builtin_id = DPOMP_BUILTIN_USER");
    return create_impl(d, device, rate_source, out);
}

int dpomp_check_rate_source(const dpomp_model_desc* d, const char* rate_source, char* log, int64_t log_size) {
    if (log && log_size > 0) log[0] = 0;
    int rc = validate_desc(d);
    if (rc != DPOMP_OK) return rc;
    std::string table;  // rate_source == NULL: the source the descriptor's rate table is turned into at create time
    if (!rate_source) {
        if (d->builtin_id == DPOMP_BUILTIN_USER) return fail(nullptr, DPOMP_EINVAL, "rate_source is NULL");
        table = rate_table_source(d);
        rate_source = table.c_str();
    }
    RtcBinary bin;
    std::string text;
    rc = rtc_compile(d, rate_source, bin, text);
    if (log && log_size > 0) {
        strncpy(log, text.c_str(), (size_t)log_size - 1);
        log[log_size - 1] = 0;
    }
    if (rc != DPOMP_OK) fail(nullptr, rc, "rate source does not compile: %.300s", text.c_str());
    return rc;
}

}  // extern "C"

namespace {

int create_impl(const dpomp_model_desc* d, int device, const char* rate_source, dpomp_handle** out) {
    if (!out) return fail(nullptr, DPOMP_EINVAL, "out is NULL");
    *out = nullptr;
    int rc = validate_desc(d);
    if (rc != DPOMP_OK) return rc;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return fail(nullptr, DPOMP_EDEVICE, "no HIP device available (%s); this library has no CPU path", hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(nullptr, DPOMP_EINVAL, "device %d out of 0..%d", device, ndev - 1);
    dpomp_handle* h = new dpomp_handle;
    h->device = device;
    e = hipSetDevice(device);
    if (e == hipSuccess) e = counted_stream_create(&h->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        rc = fail(nullptr, DPOMP_EDEVICE, "hipStreamCreate: %s", hipGetErrorString(e));
        delete h;
        return rc;
    }
    h->stream = h->own_stream;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
            h->resident_wgs = cus * 8;  // 8 waves per SIMD, 4 SIMDs per CU, 4 waves per workgroup
        // experiment switch: DPOMP_HEAD_START=0 turns the head start off, DPOMP_HEAD_START=<frac>,<workgroups> sets the
        // share of the event chain and the number of workgroups that get it (tools/size_probe.py)
        if (const char* hs = getenv("DPOMP_HEAD_START")) {
            double f = 0.0;
            int n = 0;
            const int got = sscanf(hs, "%lf,%d", &f, &n);
            if (got >= 1 && f <= 0.0) h->no_head_start = true;
            if (got >= 1 && f > 0.0) h->boost_frac = f;
            if (got == 2 && n > 0) h->boost_wgs = n;
        }
    }
    {   // device address of the static scratch array on THIS GPU (each device holds its own copy), once per device
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (device < kMaxDevices && !g_scratch_dev[device]) {
            unsigned char* ptr = nullptr;
            e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
            if (e != hipSuccess || !ptr) {
                rc = fail(nullptr, DPOMP_EDEVICE, "hipGetSymbolAddress(scratch) on device %d: %s", device, hipGetErrorString(e));
                (void)hipStreamDestroy(h->own_stream);
                delete h;
                return rc;
            }
            g_scratch_dev[device] = ptr;
        }
    }
    {   // pinned staging of dpomp_pf_loglik_static, sized for its largest call
        const size_t n_res = (size_t)DPOMP_STATIC_MAX_SEGMENTS * 12 + 16 + (size_t)d->n_obs * kSchedStride * 8 + 64;
        if (counted_host_malloc((void**)&h->static_h_result, n_res, hipHostMallocDefault) != hipSuccess ||
            counted_host_malloc((void**)&h->static_h_theta, (size_t)DPOMP_STATIC_MAX_SEGMENTS * DPOMP_MAX_PARAMS * 8, hipHostMallocDefault) != hipSuccess) {
            rc = fail(nullptr, DPOMP_ENOMEM, "hipHostMalloc of the static staging buffers failed");
            dpomp_destroy(h);
            return rc;
        }
    }
    h->T = d->n_obs;
    h->lane_util.assign((size_t)d->n_obs, -1.0);
    h->wave_chain.assign((size_t)d->n_obs, -1.0);
    h->obs_time.assign(d->obs_time, d->obs_time + d->n_obs);
    h->obs_id.assign(d->obs_id, d->obs_id + d->n_obs);
    h->ysum.resize(d->n_obs);
    for (int t = 0; t < d->n_obs; ++t) {  // sum(y.val[y_seq])  hmm_examples.jl:64
        long long s = 0;
        for (int v = 0; v < d->n_obs_values; ++v)
            if (d->obs.value_mask & (1u << v)) s += d->obs_value[(int64_t)t * d->n_obs_values + v];
        h->ysum[t] = s;
    }
    DevModel& m = h->dm;
    m.S = d->n_states;
    m.E = d->n_events;
    m.K = d->n_params;
    m.t0_index = d->t0_index;
    m.obs_state_mask = d->obs.state_mask;
    int64_t cap = d->max_events_per_interval > 0 ? d->max_events_per_interval : ((int64_t)1 << 22);
    if (cap > 0x7FFFFFFF) cap = 0x7FFFFFFF;
    m.cap = (int32_t)cap;
    const double sigma = d->obs.kind == DPOMP_OBS_GAUSSIAN ? d->obs.sigma : 1.0;
    m.obs_tmp1 = std::log(1.0 / (std::sqrt(2.0 * M_PI) * sigma));  // hmm_examples.jl:61
    m.obs_tmp2 = 2.0 * sigma * sigma;                              // :62
    h->big = model_is_big(d);
    for (int ev = 0; ev < 8; ++ev) {  // (a big model's rows, rate terms and initial condition live in its generated evaluator)
        uint64_t w = 0;
        if (ev < d->n_events && !h->big)
            for (int s = 0; s < d->n_states; ++s) w |= (uint64_t)(uint8_t)(int8_t)d->transition[ev][s] << (8 * s);
        m.trans_packed[ev] = w;
        const dpomp_rate_term& t = d->rate[ev < d->n_events ? ev : 0];
        m.rate[ev] = {t.theta_index, t.form, t.n_factors, t.factor_mask[0], t.factor_mask[1], t.denom_mask};
        m.x0[ev] = ev < d->n_states ? (int32_t)d->initial_condition[ev] : 0;
    }
    for (const PropEntry& pe : kPropTable)
        if (!h->big && pe.builtin == d->builtin_id && pe.S == d->n_states && pe.E == d->n_events) h->prop = pe.set;
    h->evaluator = h->prop.fn1 ? DPOMP_EVALUATOR_BUILTIN : DPOMP_EVALUATOR_TABLE;
    if (rate_source) {
        RtcBinary bin;
        std::string text;
        rc = rtc_compile(d, rate_source, bin, text);
        if (rc != DPOMP_OK) {
            rc = fail(nullptr, rc, "rate source does not compile: %.400s", text.c_str());
            dpomp_destroy(h);
            return rc;
        }
        if (rtc_load(h, bin, text) != DPOMP_OK) {
            rc = fail(nullptr, DPOMP_EDEVICE, "loading the compiled rate module: %s", text.c_str());
            dpomp_destroy(h);
            return rc;
        }
        h->evaluator = DPOMP_EVALUATOR_SOURCE;
    } else if (h->big) {  // a table too large for the interpreter: compiled, or not at all
        RtcBinary bin;
        std::string text;
        if (d->flags & DPOMP_FLAG_TABLE_INTERPRET) text = "DPOMP_FLAG_TABLE_INTERPRET is set";
        if (!text.empty() || rtc_compile(d, rate_table_source(d).c_str(), bin, text) != DPOMP_OK || rtc_load(h, bin, text) != DPOMP_OK) {
            rc = fail(nullptr, DPOMP_EINVAL, "a model with more than %d states / events / parameters or more than 2 factors per rate has no "
                      "compiled-in evaluator and must be compiled at create time (hipRTC): %.300s", DPOMP_COMPILED_MAX, text.c_str());
            dpomp_destroy(h);
            return rc;
        }
        h->evaluator = DPOMP_EVALUATOR_TABLE_COMPILED;
    } else if (!h->prop.fn1) {
        h->prop = generic_set(d->n_states, d->n_events);  // the table interpreter: always there as the fallback
        h->prop_is_generic = true;
        if (generic_jit_enabled(d)) {  // a table model outside the catalogue: compile its table (cached), else interpret it
            RtcBinary bin;
            std::string text;
            if (rtc_compile(d, rate_table_source(d).c_str(), bin, text) == DPOMP_OK && rtc_load(h, bin, text) == DPOMP_OK)
                h->evaluator = DPOMP_EVALUATOR_TABLE_COMPILED;
            else
                h->jit_note = text;
        }
    }
    if (const char* env = getenv("DPOMP_RPL")) h->force_rpl = atoi(env);
    if (const char* env = getenv("DPOMP_FLAT")) h->no_flat = atoi(env) == 0;
    if (const char* env = getenv("DPOMP_TILES_PER_WG")) h->tiles_per_wg = std::min(64, std::max(0, atoi(env)));
    {
        std::vector<int32_t> oid(d->n_obs);
        for (int t = 0; t < d->n_obs; ++t) oid[t] = d->obs_id[t] > 0 ? 1 : 0;
        hipError_t oe = counted_malloc((void**)&h->d_obs_time, (size_t)d->n_obs * 8);
        if (oe == hipSuccess) oe = counted_malloc((void**)&h->d_ysum, (size_t)d->n_obs * 8);
        if (oe == hipSuccess) oe = counted_malloc((void**)&h->d_obs_id, (size_t)d->n_obs * 4);
        if (oe == hipSuccess) oe = hipMemcpy(h->d_obs_time, h->obs_time.data(), (size_t)d->n_obs * 8, hipMemcpyHostToDevice);
        if (oe == hipSuccess) oe = hipMemcpy(h->d_ysum, h->ysum.data(), (size_t)d->n_obs * 8, hipMemcpyHostToDevice);
        if (oe == hipSuccess) oe = hipMemcpy(h->d_obs_id, oid.data(), (size_t)d->n_obs * 4, hipMemcpyHostToDevice);
        h->n_obs_values = d->n_obs_values;
        const size_t nval = (size_t)d->n_obs * (d->n_obs_values > 0 ? d->n_obs_values : 1);
        std::vector<long long> vals(nval, 0);
        for (size_t i = 0; i < (size_t)d->n_obs * d->n_obs_values; ++i) vals[i] = d->obs_value[i];
        if (oe == hipSuccess) oe = counted_malloc((void**)&h->d_obs_val, nval * 8);
        if (oe == hipSuccess) oe = hipMemcpy(h->d_obs_val, vals.data(), nval * 8, hipMemcpyHostToDevice);
        if (oe != hipSuccess) {
            rc = fail(nullptr, DPOMP_EDEVICE, "uploading the observation data: %s", hipGetErrorString(oe));
            dpomp_destroy(h);
            return rc;
        }
    }
    *out = h;
    return DPOMP_OK;
}

}  // namespace

extern "C" {

void dpomp_destroy(dpomp_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->rtc_module) (void)hipModuleUnload(h->rtc_module);
    (void)hipFree(h->d_obs_time);
    (void)hipFree(h->d_ysum);
    (void)hipFree(h->d_obs_id);
    (void)hipFree(h->d_obs_val);
    if (h->scratch_pop) pop_release(h->scratch_pop);
    (void)hipHostFree(h->static_h_result);
    (void)hipHostFree(h->static_h_theta);
    for (hipEvent_t e : h->event_pool) (void)hipEventDestroy(e);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

const char* dpomp_last_error(const dpomp_handle* h) { return h ? h->err.c_str() : g_global_error.c_str(); }

int dpomp_set_stream(dpomp_handle* h, void* hip_stream) {
    if (!h) return DPOMP_EINVAL;
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return DPOMP_OK;
}

int dpomp_set_resampler(dpomp_handle* h, int rs_type) {
    if (!h) return DPOMP_EINVAL;
    if (rs_type < DPOMP_RS_SYSTEMATIC || rs_type > DPOMP_RS_MULTINOMIAL) return fail(h, DPOMP_EINVAL, "rs_type %d unknown", rs_type);
    h->rs_type = rs_type;
    return DPOMP_OK;
}

int dpomp_set_ess_threshold(dpomp_handle* h, double ess_frac) {
    if (!h) return DPOMP_EINVAL;
    if (!(ess_frac >= 0.0) || ess_frac > 1.0e6) return fail(h, DPOMP_EINVAL, "ess threshold must be >= 0 (0 = resample at every observation)");
    h->ess_frac = ess_frac;
    return DPOMP_OK;
}

int dpomp_set_propagate_variant(dpomp_handle* h, int rpl) {
    if (!h) return DPOMP_EINVAL;
    if (rpl != 0 && rpl != 1 && rpl != 2 && rpl != 4 && rpl != 16 && rpl != 32 && rpl != 33)
        return fail(h, DPOMP_EINVAL, "propagate variant must be 0, 1, 2, 4, 16, 32 or 33");
    if (h->big && rpl > 1)
        return fail(h, DPOMP_EINVAL, "a model with more than %d states / events / parameters runs the one-particle-per-lane schedule only", DPOMP_COMPILED_MAX);
    h->force_rpl = rpl;
    return DPOMP_OK;
}

int dpomp_pop_alloc(dpomp_handle* h, int64_t B, int64_t P, dpomp_pop** out) {
    if (!h || !out) return DPOMP_EINVAL;
    *out = nullptr;
    return pop_alloc_impl(h, B, P, out);
}

void dpomp_pop_free(dpomp_handle* h, dpomp_pop* pop) {
    if (h) (void)hipSetDevice(h->device);
    pop_release(pop);
}

int dpomp_pf_partial(dpomp_handle* h, dpomp_pop* pop, const double* theta, int32_t ymin, int32_t ymax,
                     uint64_t seed, int64_t seg_base, double* out_ll, uint32_t* out_status) {
    if (!h) return DPOMP_EINVAL;
    if (!pop || !out_ll) return fail(h, DPOMP_EINVAL, "pop or out_ll is NULL");
    return run_partial(h, pop, theta, ymin, ymax, seed, seg_base, DOMAIN_PROPAGATE, out_ll, out_status, nullptr);
}

int dpomp_pf_loglik(dpomp_handle* h, const double* theta, int64_t B, int64_t P, uint64_t seed,
                    int64_t seg_base, double* out_ll, uint32_t* out_status) {
    if (!h) return DPOMP_EINVAL;
    if (!out_ll) return fail(h, DPOMP_EINVAL, "out_ll is NULL");
    if (!h->scratch_pop || h->scratch_pop->B != B || h->scratch_pop->P != P) {
        if (h->scratch_pop) {
            pop_release(h->scratch_pop);
            h->scratch_pop = nullptr;
        }
        int rc = pop_alloc_impl(h, B, P, &h->scratch_pop);
        if (rc != DPOMP_OK) return rc;
    }
    return run_partial(h, h->scratch_pop, theta, 1, h->T, seed, seg_base, DOMAIN_PROPAGATE, out_ll, out_status, nullptr);
}

int dpomp_pf_loglik_static(dpomp_handle* h, const double* theta, int64_t B, int64_t P, uint64_t seed,
                           int64_t seg_base, double* out_ll, uint32_t* out_status) {
    if (!h) return DPOMP_EINVAL;
    if (!out_ll) return fail(h, DPOMP_EINVAL, "out_ll is NULL");
    if (int rc = check_pop_dims(h, B, P)) return rc;
    if (B > DPOMP_STATIC_MAX_SEGMENTS)
        return fail(h, DPOMP_EINVAL, "dpomp_pf_loglik_static: B=%lld exceeds its maximum of %d filters per call", (long long)B, DPOMP_STATIC_MAX_SEGMENTS);
    unsigned char* const base = h->device < kMaxDevices ? g_scratch_dev[h->device] : nullptr;
    if (!base) return fail(h, DPOMP_ESTATE, "no static scratch memory on device %d", h->device);
    dpomp_pop view;  // on the stack: a view of the static array, owns nothing
    view.static_mem = true;
    pop_set_shape(h, &view, B, P);
    const size_t need = pop_carve(h, &view, nullptr);
    if (need > kStaticScratchBytes)
        return fail(h, DPOMP_EINVAL, "dpomp_pf_loglik_static: B=%lld x P=%lld needs %zu bytes of working memory, the static scratch array holds %zu "
                    "(use dpomp_pf_loglik, which allocates)", (long long)B, (long long)P, need, kStaticScratchBytes);
    pop_carve(h, &view, base);
    pop_bind_ctl(&view);
    view.h_result = h->static_h_result;
    view.h_theta = h->static_h_theta;
    return run_partial(h, &view, theta, 1, h->T, seed, seg_base, DOMAIN_PROPAGATE, out_ll, out_status, nullptr);
}

int dpomp_set_generic_jit(int mode) {
    if (mode < -1 || mode > 1) return DPOMP_EINVAL;
    g_generic_jit = mode;
    return DPOMP_OK;
}

int dpomp_get_evaluator(const dpomp_handle* h) { return h ? h->evaluator : DPOMP_EINVAL; }

int dpomp_abi_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(dpomp_model_desc);
        case 1: return (int)sizeof(dpomp_rate_term);
        case 2: return (int)sizeof(dpomp_obs_model);
        default: return -1;
    }
}

int dpomp_pop_gather(dpomp_handle* h, dpomp_pop* p, const int64_t* ancestors) {
    if (!h) return DPOMP_EINVAL;
    if (!p || !ancestors) return fail(h, DPOMP_EINVAL, "pop or ancestors is NULL");
    if (p->T != h->T || p->S != h->dm.S || p->K != h->dm.K) return fail(h, DPOMP_EINVAL, "population belongs to another filter");
    for (int64_t b = 0; b < p->B; ++b)
        if (ancestors[b] < 0 || ancestors[b] >= p->B) return fail(h, DPOMP_EINVAL, "ancestors[%lld]=%lld out of range", (long long)b, (long long)ancestors[b]);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(p->ancestors, ancestors, (size_t)p->B * 8, hipMemcpyHostToDevice, h->stream));
    const int64_t seg_ints = (int64_t)p->S * p->Pp;
    {
        timed_launch(h, KID_COPYSEG, k_copy_segments, dim3((unsigned)(seg_ints / kTile), (unsigned)p->B), dim3(kBlock), 0,
                     h->stream, (const int32_t*)p->state[p->cur], p->state[p->cur ^ 1], (const int64_t*)p->ancestors, seg_ints);
    }
    p->cur ^= 1;
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    collect_timings_if_many(h);
    return DPOMP_OK;
}

int dpomp_pop_take(dpomp_handle* h, dpomp_pop* dst, const dpomp_pop* src, const int64_t* take) {
    if (!h) return DPOMP_EINVAL;
    if (!dst || !src || !take || dst == src) return fail(h, DPOMP_EINVAL, "dst/src/take invalid");
    if (dst->P != src->P || dst->S != src->S) return fail(h, DPOMP_EINVAL, "populations differ in shape");
    if (dst->T != h->T || dst->S != h->dm.S || dst->K != h->dm.K) return fail(h, DPOMP_EINVAL, "population belongs to another filter");
    bool any = false;
    for (int64_t b = 0; b < dst->B; ++b) {
        if (take[b] >= src->B) return fail(h, DPOMP_EINVAL, "take[%lld]=%lld out of range", (long long)b, (long long)take[b]);
        any |= take[b] >= 0;
    }
    if (!any) return DPOMP_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(dst->ancestors, take, (size_t)dst->B * 8, hipMemcpyHostToDevice, h->stream));
    const int64_t seg_ints = (int64_t)dst->S * dst->Pp;
    {
        timed_launch(h, KID_COPYSEG, k_take_segments, dim3((unsigned)(seg_ints / kTile), (unsigned)dst->B), dim3(kBlock), 0, h->stream,
                     (const int32_t*)src->state[src->cur], dst->state[dst->cur], (const int64_t*)dst->ancestors, seg_ints);
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    collect_timings_if_many(h);
    dst->initialised = true;
    return DPOMP_OK;
}

int dpomp_pop_device_state(dpomp_handle* h, dpomp_pop* p, int for_write, void** device_ptr, int64_t* stride) {
    if (!h) return DPOMP_EINVAL;
    if (!p || !device_ptr || !stride) return fail(h, DPOMP_EINVAL, "NULL argument");
    *device_ptr = p->state[p->cur];
    *stride = p->Pp;
    if (for_write) p->initialised = true;
    return DPOMP_OK;
}

int dpomp_pop_download(dpomp_handle* h, const dpomp_pop* p, int64_t b, int64_t* out) {
    if (!h) return DPOMP_EINVAL;
    if (!p || !out || b < 0 || b >= p->B) return fail(h, DPOMP_EINVAL, "bad pop/out/b");
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<int32_t> tmp((size_t)p->S * p->Pp);
    HIPCHK(h, hipMemcpyAsync(tmp.data(), p->state[p->cur] + (size_t)b * p->S * p->Pp, tmp.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int s = 0; s < p->S; ++s)
        for (int64_t i = 0; i < p->P; ++i) out[i + p->P * s] = tmp[(size_t)s * p->Pp + i];
    return DPOMP_OK;
}

int dpomp_pop_upload(dpomp_handle* h, dpomp_pop* p, int64_t b, const int64_t* in) {
    if (!h) return DPOMP_EINVAL;
    if (!p || !in || b < 0 || b >= p->B) return fail(h, DPOMP_EINVAL, "bad pop/in/b");
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<int32_t> tmp((size_t)p->S * p->Pp, 0);
    for (int s = 0; s < p->S; ++s)
        for (int64_t i = 0; i < p->P; ++i) {
            const int64_t v = in[i + p->P * s];
            if (v < 0 || v >= ((int64_t)1 << 30)) return fail(h, DPOMP_EINVAL, "state value %lld out of 0..2^30", (long long)v);
            tmp[(size_t)s * p->Pp + i] = (int32_t)v;
        }
    HIPCHK(h, hipMemcpyAsync(p->state[p->cur] + (size_t)b * p->S * p->Pp, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    p->initialised = true;
    return DPOMP_OK;
}

int dpomp_simulate(dpomp_handle* h, const double* theta, int64_t B, int64_t P, uint64_t seed, int64_t seg_base, int64_t* out) {
    if (!h) return DPOMP_EINVAL;
    if (!out) return fail(h, DPOMP_EINVAL, "out is NULL");
    dpomp_pop* p = nullptr;
    int rc = pop_alloc_impl(h, B, P, &p);
    if (rc != DPOMP_OK) return rc;
    int32_t* rec = nullptr;
    const size_t n = (size_t)h->T * B * p->S * p->Pp;
    if (counted_malloc((void**)&rec, n * 4) != hipSuccess) {
        pop_release(p);
        return fail(h, DPOMP_ENOMEM, "hipMalloc(%zu) for the simulation record", n * 4);
    }
    rc = run_partial(h, p, theta, 1, h->T, seed, seg_base, DOMAIN_SIMULATE, nullptr, nullptr, rec);
    if (rc == DPOMP_OK) {
        std::vector<int32_t> tmp(n);
        hipError_t e = hipMemcpy(tmp.data(), rec, n * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(h, DPOMP_EDEVICE, "hipMemcpy: %s", hipGetErrorString(e));
        else
            for (int t = 0; t < h->T; ++t)
                for (int64_t b = 0; b < B; ++b)
                    for (int s = 0; s < p->S; ++s)
                        for (int64_t i = 0; i < P; ++i)
                            out[((b * h->T + t) * p->S + s) * P + i] = tmp[(((size_t)t * B + b) * p->S + s) * p->Pp + i];
    }
    (void)hipFree(rec);
    pop_release(p);
    return rc;
}

int dpomp_simulate_events(dpomp_handle* h, const double* theta, int64_t P, uint64_t seed, int64_t seg, int64_t max_events,
                          int64_t* out_counts, double* out_times, int32_t* out_types, int64_t* out_states) {
    if (!h) return DPOMP_EINVAL;
    if (!theta || !out_counts || !out_times || !out_types || !out_states) return fail(h, DPOMP_EINVAL, "NULL argument");
    if (P < 1 || max_events < 1 || P * max_events > ((int64_t)1 << 31)) return fail(h, DPOMP_EINVAL, "P * max_events out of range (1 .. 2^31)");
    if (seg < 0 || seg > 0xFFFFFFFFll) return fail(h, DPOMP_EINVAL, "seg out of range");
    HIPCHK(h, hipSetDevice(h->device));
    const int S = h->dm.S, T = h->T, K = h->dm.K;
    SimArgs a;
    a.m = h->dm;
    double *d_theta = nullptr, *d_obs = nullptr, *d_time = nullptr;
    long long *d_counts = nullptr, *d_states = nullptr;
    int32_t* d_type = nullptr;
    const size_t nev = (size_t)P * max_events;
    hipError_t e = counted_malloc((void**)&d_theta, (size_t)K * 8);
    if (e == hipSuccess) e = counted_malloc((void**)&d_obs, (size_t)T * 8);
    if (e == hipSuccess) e = counted_malloc((void**)&d_counts, (size_t)P * 8);
    if (e == hipSuccess) e = counted_malloc((void**)&d_states, (size_t)T * S * P * 8);
    if (e == hipSuccess) e = counted_malloc((void**)&d_time, nev * 8);
    if (e == hipSuccess) e = counted_malloc((void**)&d_type, nev * 4);
    int rc = DPOMP_OK;
    if (e != hipSuccess) {
        rc = fail(h, DPOMP_ENOMEM, "hipMalloc for the event record: %s", hipGetErrorString(e));
    } else {
        (void)hipMemcpyAsync(d_theta, theta, (size_t)K * 8, hipMemcpyHostToDevice, h->stream);
        (void)hipMemcpyAsync(d_obs, h->obs_time.data(), (size_t)T * 8, hipMemcpyHostToDevice, h->stream);
        a.theta = d_theta;
        a.obs_time = d_obs;
        a.counts = d_counts;
        a.ev_time = d_time;
        a.ev_type = d_type;
        a.states = d_states;
        a.max_events = max_events;
        a.P = (int32_t)P;
        a.T = T;
        a.seg = (uint32_t)seg;
        a.seed = seed;
        if (h->rtc) {
            void* params[] = {&a};
            e = hipModuleLaunchKernel(h->rtc_sim, (unsigned)((P + kBlock - 1) / kBlock), 1, 1, kBlock, 1, 1, 0, h->stream, params, nullptr);
        } else {
            hipLaunchKernelGGL(h->prop.sim, dim3((unsigned)((P + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, a);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(out_counts, d_counts, (size_t)P * 8, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(out_states, d_states, (size_t)T * S * P * 8, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(out_times, d_time, nev * 8, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(out_types, d_type, nev * 4, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(h, DPOMP_EDEVICE, "simulate_events: %s", hipGetErrorString(e));
    }
    (void)hipFree(d_theta);
    (void)hipFree(d_obs);
    (void)hipFree(d_counts);
    (void)hipFree(d_states);
    (void)hipFree(d_time);
    (void)hipFree(d_type);
    return rc;
}

int dpomp_debug_download(dpomp_handle* h, const dpomp_pop* p, int what, int64_t b, void* out) {
    if (!h) return DPOMP_EINVAL;
    if (!p) p = h->scratch_pop;
    if (!p || !out || b < 0 || b >= p->B) return fail(h, DPOMP_EINVAL, "bad pop/out/b");
    HIPCHK(h, hipSetDevice(h->device));
    if (what == DPOMP_DBG_LOGW) {
        HIPCHK(h, hipMemcpy(out, p->logw + (size_t)b * p->Pp, (size_t)p->P * 8, hipMemcpyDeviceToHost));
    } else if (what == DPOMP_DBG_OFFSPRING_END) {
        HIPCHK(h, hipMemcpy(out, p->offs + (size_t)b * p->Pp, (size_t)p->P * 4, hipMemcpyDeviceToHost));
    } else {
        return fail(h, DPOMP_EINVAL, "unknown debug item %d", what);
    }
    return DPOMP_OK;
}

int dpomp_selftest_numerics(dpomp_handle* h, int64_t n, uint64_t seed, const double* x_in, double* out_uniform,
                            double* out_log, double* out_exp) {
    if (!h) return DPOMP_EINVAL;
    if (n < 1) return fail(h, DPOMP_EINVAL, "n < 1");
    HIPCHK(h, hipSetDevice(h->device));
    DeviceScratch scratch;
    double *dx = nullptr, *du = nullptr, *dl = nullptr, *de = nullptr;
    HIPCHK(h, scratch.doubles(&dx, (size_t)n));
    HIPCHK(h, scratch.doubles(&du, (size_t)n));
    HIPCHK(h, scratch.doubles(&dl, (size_t)n));
    HIPCHK(h, scratch.doubles(&de, (size_t)n));
    if (x_in) HIPCHK(h, hipMemcpy(dx, x_in, (size_t)n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_selftest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, n, seed,
                       x_in ? (const double*)dx : (const double*)nullptr, du, dl, de);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (out_uniform) HIPCHK(h, hipMemcpy(out_uniform, du, (size_t)n * 8, hipMemcpyDeviceToHost));
    if (out_log && x_in) HIPCHK(h, hipMemcpy(out_log, dl, (size_t)n * 8, hipMemcpyDeviceToHost));
    if (out_exp && x_in) HIPCHK(h, hipMemcpy(out_exp, de, (size_t)n * 8, hipMemcpyDeviceToHost));
    return DPOMP_OK;
}

int dpomp_selftest_division(dpomp_handle* h, int64_t n, const double* a_in, const double* b_in, double* out_fast,
                            double* out_plain) {
    if (!h || !a_in || !b_in || !out_fast || !out_plain) return fail(h, DPOMP_EINVAL, "null argument");
    if (n < 1) return fail(h, DPOMP_EINVAL, "n < 1");
    HIPCHK(h, hipSetDevice(h->device));
    DeviceScratch scratch;
    double* d[4] = {nullptr, nullptr, nullptr, nullptr};
    for (auto& p : d) HIPCHK(h, scratch.doubles(&p, (size_t)n));
    HIPCHK(h, hipMemcpy(d[0], a_in, (size_t)n * 8, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(d[1], b_in, (size_t)n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_selftest_div, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, n, d[0], d[1], d[2], d[3]);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out_fast, d[2], (size_t)n * 8, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(out_plain, d[3], (size_t)n * 8, hipMemcpyDeviceToHost));
    return DPOMP_OK;
}

int dpomp_set_profiling(dpomp_handle* h, int on) {
    if (!h) return DPOMP_EINVAL;
    if (!on && !h->pending.empty()) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        collect_timings(h);
    }
    h->profiling_every = on > 0 ? on : 0;
    h->profiling = false;
    h->call_counter = 0;
    if (on && h->null_bracket_ms == 0.0) {  // calibrate the bracket itself once
        HIPCHK(h, hipSetDevice(h->device));
        // 64 empty brackets enqueued back to back behind a kernel (a busy queue, as in a filter call), one
        // synchronisation at the end: per-bracket elapsed = what the two markers themselves cost.
        std::vector<float> v;
        const int n = 64;
        std::vector<hipEvent_t> ev(2 * n);
        for (auto& e : ev) HIPCHK(h, counted_event_create(&e));
        int32_t* scratch = nullptr;
        HIPCHK(h, counted_malloc((void**)&scratch, (size_t)kTile * 64 * 4));
        for (int i = 0; i < n; ++i) {
            // filler: S rows of 8 tiles each (S <= 8 -> at most 64 tiles = the scratch size), 16-B aligned rows
            DevModel filler = h->dm;
            filler.S = std::min(filler.S, 8);  // (DevModel holds 8 components; the scratch block 8 rows)
            hipLaunchKernelGGL(k_init, dim3(8, 1), dim3(kBlock), 0, h->stream, filler, scratch, (int64_t)kTile * 8);
            (void)hipEventRecord(ev[2 * i], h->stream);
            (void)hipEventRecord(ev[2 * i + 1], h->stream);
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        for (int i = 0; i < n; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) == hipSuccess) v.push_back(ms);
        }
        for (auto& e : ev) (void)hipEventDestroy(e);
        (void)hipFree(scratch);
        if (!v.empty()) {
            std::sort(v.begin(), v.end());
            h->null_bracket_ms = v[v.size() / 2];
        }
    }
    return DPOMP_OK;
}

int dpomp_get_null_bracket_ms(dpomp_handle* h, double* ms) {
    if (!h || !ms) return DPOMP_EINVAL;
    *ms = h->null_bracket_ms;
    return DPOMP_OK;
}

int dpomp_kernel_count(void) { return KID_COUNT; }

const char* dpomp_kernel_name(int kid) { return (kid >= 0 && kid < KID_COUNT) ? kKernelNames[kid] : ""; }

int dpomp_get_kernel_stat(dpomp_handle* h, int kid, int64_t* launches, double* total_ms) {
    if (!h || kid < 0 || kid >= KID_COUNT) return DPOMP_EINVAL;
    if (!h->pending.empty()) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        collect_timings(h);
    }
    if (launches) *launches = h->stats[kid].launches;
    if (total_ms) *total_ms = h->stats[kid].total_ms;
    return DPOMP_OK;
}

int dpomp_get_lane_utilisation(dpomp_handle* h, int obs_index, double* util) {
    if (!h || !util) return fail(h, DPOMP_EINVAL, "null argument");
    if (obs_index < 1 || obs_index > h->T) return fail(h, DPOMP_EINVAL, "obs_index %d out of 1..%d", obs_index, h->T);
    *util = h->lane_util[(size_t)obs_index - 1];
    return DPOMP_OK;
}

int64_t dpomp_static_max_bytes(void) { return (int64_t)kStaticScratchBytes; }

long long dpomp_debug_alloc_count(void) { return g_alloc_calls.load(std::memory_order_relaxed); }

int dpomp_last_event_count(dpomp_handle* h, uint64_t* events) {
    if (!h || !events) return DPOMP_EINVAL;
    *events = h->last_events;
    return DPOMP_OK;
}

}  // extern "C"
