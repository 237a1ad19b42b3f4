// dpomp_kernels.h -- gfx950 kernels of the bootstrap particle filter.
//
//   k_propagate / k_propagate_c / k_propagate_q
//                 iterate_particles!   src/hmm_particle_filter.jl:9-33  (+ choose_event src/hmm_cmn.jl:4-10,
//                 rate functions src/hmm_examples.jl:103-168, transition :11-16, obs model :59-67);
//                 three schedules of the same arithmetic: one particle per lane, + workgroup compaction
//                 (default for chip-filling launches), LDS queue with lane refill
//   k_wsum, k_scan_blocks, k_offsets, k_gather
//                 log mean weight  src/hmm_particle_filter.jl:60  and rsp_systematic  src/hmm_pf_resample.jl:24-42
//   k_gather_pos  rsp_stratified / rsp_multinomial (documented intent)  src/hmm_pf_resample.jl:46-63, :5-20
//   k_simulate_rec  gillespie_sim with the Event history  src/hmm_sim.jl:55-102
//   k_init, k_copy_segments, k_take_segments   population bookkeeping (hmm_particle_filter.jl:42-44, hmm_ibis.jl:71-75, :109)
//
// The file also compiles under hipRTC (__HIPCC_RTC__) together with a user-supplied rate function.
//
// Layout in HBM (per population handle): state[b][s][p] int32, p fastest, row stride Pp
// (P rounded up to the 1024-slot tile) -- the reference's column-major P x S matrix per
// segment, i.e. structure-of-arrays, so a wave reads 64 consecutive ints per component;
// logw[b][p] double; offs[b][p] int32 (offspring boundaries N_j).
#pragma once
#if !defined(__HIPCC_RTC__)  // hipRTC (run-time compiled user rate functions) has these built in
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "dpomp_numerics.h"

namespace dpomp {

template <bool C, class A, class B> struct cond_t { using type = A; };
template <class A, class B> struct cond_t<false, A, B> { using type = B; };

constexpr int kBlock = 256;        // threads per workgroup (4 waves of 64)
constexpr int kItems = 4;          // slots per thread in the streaming kernels
constexpr int kTile = kBlock * kItems;  // 1024 slots per workgroup

struct DevRateTerm {
    int32_t theta_index, form, n_factors;
    uint32_t fmask0, fmask1, dmask;
};

struct DevModel {
    int32_t S, E, K;
    int32_t t0_index;  // 0 = none, else 1-based theta index (DPOMPModel.t0_index)
    uint32_t obs_state_mask;
    int32_t cap;       // max events per particle and interval
    double obs_tmp1, obs_tmp2;  // log(1/(sqrt(2 pi) sigma)), 2 sigma^2
    uint64_t trans_packed[8];   // m_transition row e as one int8 per state component
    DevRateTerm rate[8];
    int32_t x0[8];
};

struct PropArgs {
    DevModel m;
    int32_t* state;              // current population buffer
    double* logw;
    const double* theta;         // [B][K]
    unsigned long long* segmax;  // [B] ordered key of max log-weight (this observation)
    unsigned long long* events;  // [1]
    uint32_t* status;            // [B]
    int64_t Pp;
    int32_t P;
    uint32_t obs_k, seg_base, domain;
    uint64_t seed;
    double t_prev, t_max;
    long long ysum;
    int32_t use_theta_t0;  // first interval of a ymin == 1 call with t0_index > 0
    int32_t from_x0;       // first interval of a ymin == 1 call: every particle starts at the initial condition
                           // (pop[i,:] .= initial_condition, hmm_particle_filter.jl:42-44) instead of being loaded
    int32_t want_weight;
    const int32_t* carry_in;  // [B] ESS-adaptive mode: 1 = add the carried log-weight (still in logw) to the new one
    const long long* yval;    // [V] Observation.val of this observation (user observation models)
    int32_t n_yval;
    unsigned long long* sched;  // this interval's counters, see sched_add()
    // Resampling fused into the load (systematic resampling of the previous observation): when `parents` is set, slot p
    // starts from the state of its ancestor in `parents` -- pop[i,:] = old_p[j_i,:] (hmm_pf_resample.jl:38) without a
    // separate gather pass -- and the propagated state goes to `state` (the other buffer).
    const int32_t* parents;       // [B][S][Pp] population before resampling, or null
    const int32_t* offs;          // [B][Pp]    offspring boundaries N_j of that resampling
    const int32_t* first_parent;  // [B][Pp/256] parent block (of 256) of the first slot of every 256-slot block
    const int32_t* noresample;    // [B] or null: 1 = the population was kept (ESS-adaptive mode): slot p descends from p
    // Head start (compaction kernel): the workgroups of the launch from linear index `boost_from` on -- those that are
    // dispatched last -- run at raised wave priority until their particles have executed `boost_events` events
    // (0 = no head start).  Scheduling only: results do not depend on it.
    int32_t boost_from, boost_events;
    int32_t B;  // filters of the launch (kModeFlat lays them end to end)
#ifdef DPOMP_TRACE_WG
    unsigned long long* trace;    // [workgroups][8] timeline of every workgroup (experimental builds: tools/wg_timeline.py)
#endif
};

constexpr int kParentBlock = 256;                       // granularity of the first_parent map (= slots per propagate workgroup)
constexpr int kStageBlocks = 6;                         // parent blocks a workgroup stages in LDS for the ancestor search
constexpr int kStageSlots = kStageBlocks * kParentBlock;  // 1536 boundaries = 6 KB
// The workgroups of a launch publish their maximum log-weight into one word per filter.  Thousands of atomics on one
// address serialise at the L2 (~9 ns each: 37 us for the 4096 workgroups of a 2^20-particle filter, most of a launch
// in which little happens), but the maximum is established by the first few workgroups: read it first and only the
// record breakers issue the atomic.  (A stale, smaller value read from a cache merely causes a redundant atomic.)
__device__ __forceinline__ void publish_max(unsigned long long* slot, unsigned long long key) {
    if (key > __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMax(slot, key);
}
constexpr int kQueueChunk = 16;  // events between two refill points of k_propagate_q
// Per-interval counters: executed events, and the sum over waves of 64 * (largest event count in the wave) -- what a
// one-particle-per-lane schedule makes of the interval (dpomp_get_lane_utilisation).  Every workgroup adds to both,
// and atomics that share a 128-byte line serialise at the L2 (~9 ns each; two counters side by side cost a
// 4096-workgroup launch 70 us).  Each counter is therefore kSchedSlots partial sums on lines of their own, the slot
// chosen by the workgroup index; the host adds them up.
constexpr int kSchedSlots = 8;
constexpr int kSchedLine = 16;                             // u64 per 128-byte line
constexpr int kSchedStride = 2 * kSchedSlots * kSchedLine;  // u64 per observation interval
__device__ __forceinline__ void sched_add(unsigned long long* sched, unsigned long long events, unsigned long long slots) {
    const int s = (int)(blockIdx.x & (kSchedSlots - 1));
    atomicAdd(&sched[s * kSchedLine], events);
    atomicAdd(&sched[(kSchedSlots + s) * kSchedLine], slots);
}

// ---- rate evaluators (the pluggable part) -------------------------------------------------
// Each evaluator restates one reference rate function with the same association order:
// Julia's `θ[1] * pop[1] * pop[2]` is ((θ·Float64(x1))·Float64(x2)).
template <int S_, int E_>
struct RGeneric {  // interprets DevModel::rate (any model that fits the descriptor)
    static constexpr int S = S_, E = E_;
    static constexpr bool kGeneric = true;
    static constexpr bool kUserObs = false;
    static constexpr bool kBig = false;
    static __device__ __forceinline__ int msum(uint32_t mask, const int (&x)[S]) {
        int s = 0;
#pragma unroll
        for (int i = 0; i < S; ++i)
            if (mask & (1u << i)) s += x[i];
        return s;
    }
    static __device__ __forceinline__ void rates(const DevModel& m, const double* th,
                                                 const int (&x)[S], double (&r)[E]) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (e < m.E) {
                const DevRateTerm& t = m.rate[e];
                const double theta = th[t.theta_index];
                if (t.form == 0) {
                    double v = theta;
                    if (t.n_factors > 0) v = v * (double)msum(t.fmask0, x);
                    if (t.n_factors > 1) v = v * (double)msum(t.fmask1, x);
                    if (t.dmask) v = v / (double)msum(t.dmask, x);
                    r[e] = v;
                } else {
                    long long prod = 1;
                    if (t.n_factors > 0) prod *= (long long)msum(t.fmask0, x);
                    if (t.n_factors > 1) prod *= (long long)msum(t.fmask1, x);
                    double q = (double)prod;
                    if (t.dmask) q = q / (double)msum(t.dmask, x);
                    r[e] = theta * q;
                }
            } else {
                r[e] = 0.0;
            }
        }
    }
};

template <int S_, int E_>
struct RBase {
    static constexpr int S = S_, E = E_;
    static constexpr bool kGeneric = false;
    static constexpr bool kUserObs = false;
    // kBig (run-time compiled models only): more than 8 state components, events or parameters.  Such a model does
    // not fit DevModel's fixed arrays: its initial condition and transition rows are constants of the generated
    // evaluator (R::x0(s), R::apply_event(e, x)), and its parameters are read from memory, not held in registers.
    static constexpr bool kBig = false;
    static __device__ __forceinline__ double total(const int (&x)[S_]) {
        int s = 0;
#pragma unroll
        for (int i = 0; i < S_; ++i) s += x[i];
        return (double)s;
    }
};
// si_rf / si_rf_fd   hmm_examples.jl:103-105, :126-128
template <bool FD>
struct RSi : RBase<2, 1> {
    static __device__ __forceinline__ void rates(const DevModel&, const double* th,
                                                 const int (&x)[2], double (&r)[1]) {
        r[0] = th[0] * (double)x[0] * (double)x[1];
        if (FD) r[0] = r[0] / total(x);
    }
};
// sir_rf / sir_rf_fd   :107-110, :130-133   (SIR: S=3, SIS: S=2)
template <int S_, bool FD>
struct RSir : RBase<S_, 2> {
    static __device__ __forceinline__ void rates(const DevModel&, const double* th,
                                                 const int (&x)[S_], double (&r)[2]) {
        r[0] = th[0] * (double)x[0] * (double)x[1];
        if (FD) r[0] = r[0] / RBase<S_, 2>::total(x);
        r[1] = th[1] * (double)x[1];
    }
};
// sei_rf / sei_rf_fd   :112-115, :135-138
template <bool FD>
struct RSei : RBase<3, 2> {
    static __device__ __forceinline__ void rates(const DevModel&, const double* th,
                                                 const int (&x)[3], double (&r)[2]) {
        r[0] = th[0] * (double)x[0] * (double)x[2];
        if (FD) r[0] = r[0] / total(x);
        r[1] = th[1] * (double)x[1];
    }
};
// seir_rf / seir_rf_fd   :117-121, :140-144   (SEIR: S=4, SEIS: S=3)
template <int S_, bool FD>
struct RSeir : RBase<S_, 3> {
    static __device__ __forceinline__ void rates(const DevModel&, const double* th,
                                                 const int (&x)[S_], double (&r)[3]) {
        r[0] = th[0] * (double)x[0] * (double)x[2];
        if (FD) r[0] = r[0] / RBase<S_, 3>::total(x);
        r[1] = th[1] * (double)x[1];
        r[2] = th[2] * (double)x[2];
    }
};
// lotka_rf   :149-154
struct RLotka : RBase<2, 3> {
    static __device__ __forceinline__ void rates(const DevModel&, const double* th,
                                                 const int (&x)[2], double (&r)[3]) {
        r[0] = th[0] * (double)x[1];
        r[1] = th[1] * (double)x[0] * (double)x[1];
        r[2] = th[2] * (double)x[0];
    }
};
// rossmac_rf   :159-168
struct RRossmac : RBase<4, 6> {
    static __device__ __forceinline__ void rates(const DevModel&, const double* th,
                                                 const int (&x)[4], double (&r)[6]) {
        r[0] = th[0] * (double)(x[2] + x[3]);
        r[1] = th[0] * (double)x[2];
        r[2] = th[0] * (double)x[3];
        r[3] = th[1] * ((double)((long long)x[0] * (long long)x[3]) / (double)(x[0] + x[1]));
        r[4] = th[2] * ((double)((long long)x[1] * (long long)x[2]) / (double)(x[2] + x[3]));
        r[5] = th[3] * (double)x[1];
    }
};

// The parameter vector of one filter in registers.  Kernels whose loop crosses a barrier (compaction, LDS queue,
// whole-filter) cannot have the compiler hoist the uniform theta loads out of the event loop, and a scalar load
// per event costs more than the RNG; compiled evaluators index theta with constants, so a register copy costs
// only the entries they use.  The table interpreter indexes theta at run time and keeps reading memory.
template <class R>
struct ThetaRegs {
    double v[8];
    __device__ __forceinline__ ThetaRegs(const double* th, int K) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (!R::kGeneric && !R::kBig && i < K) ? th[i] : 0.0;
    }
    __device__ __forceinline__ const double* get(const double* th) const { return (R::kGeneric || R::kBig) ? th : v; }
};

// Observation log-density of one particle: partial_gaussian_obs_model (hmm_examples.jl:59-67),
//   tmp1 - (sum(y.val[y_seq]) - sum(pop[seq]))^2 / tmp2   with the integer square taken before the promotion,
// or, for run-time compiled models that define it, the user's dpomp_user_obs(theta, x, y.val, length(y.val)).
template <class R>
__device__ __forceinline__ double obs_log_weight(const DevModel& m, const double* th, const int (&x)[R::S], long long ysum,
                                                 const long long* yval, int n_yval) {
    if constexpr (R::kUserObs) {
        return R::user_obs(th, x, yval, n_yval);
    } else {
        int xs = 0;
#pragma unroll
        for (int s = 0; s < R::S; ++s)
            if (m.obs_state_mask & (1u << s)) xs += x[s];
        const long long d = ysum - (long long)xs;
        return m.obs_tmp1 - ((double)(d * d)) / m.obs_tmp2;
    }
}

// A LONE wave (a launch that does not fill the chip: one wave per SIMD) also waits for its own instruction fetches:
// after the taken branch at the end of every trip of the event loop the fetch starts over at the loop's first 32-byte
// block, and how that block and the next ones fall decides how well the fetches hide behind the dependent chain.
// Shifting the loop of the whole-filter kernel in steps of 4 bytes moved a 200-particle SIS filter between 333 and
// 367 us (period 32 bytes, smooth in between), the static kernel's between 413 and 441 us at 4096 particles; aligning
// every loop to 32 or 64 bytes (-falign-loops) gives the worst case.  The kernels of that regime therefore (a) run two
// events per trip -- the same calls, each with its own exits: one taken branch per two events (200 particles 334 ->
// 325 us, 4096 particles 414 -> 397 us; four per trip gain nothing more) -- and (b) pin where the loop starts: aligned
// to 32 bytes and NOPS s_nop (0xBF800000) further, the compiler's own loop preheader following.  (The compaction
// kernel, eight waves per SIMD, does not care: +-0.3 % over the same sweep.)
#define DPOMP_LOOP_AT_(NOPS) asm volatile(".p2align 5\n.fill " #NOPS ", 4, 0xBF800000")
#define DPOMP_LOOP_AT(NOPS) DPOMP_LOOP_AT_(NOPS)
#ifndef DPOMP_PAD_SMALL   // (overridden by tools/loop_offset_sweep.py)
#define DPOMP_PAD_SMALL 4
#endif
#ifndef DPOMP_PAD_STATIC
#define DPOMP_PAD_STATIC 0
#endif

// ---- wave / block helpers --------------------------------------------------------------
// Wave votes straight from the lane mask (HIP's __any/__ballot go through an integer compare in VGPRs: two VALU
// instructions per call, which matters inside the event loop).
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// Reductions and the scan go through DPP row shifts and row broadcasts (one v_mov_dpp per 32-bit half and step) rather
// than ds_bpermute shuffles, whose lane index costs four VALU instructions a step to compute.  Six steps leave the
// inclusive scan in every lane and the total in lane 63, which goes to all lanes through an SGPR.  Integer sums and
// max are order-free: same bits as any other reduction order.  All 64 lanes must be active at the call.
// dpp_ctrl: 0x110 + n = row_shr:n (within rows of 16 lanes), 0x142 = row_bcast:15 (lane 15 of each row to the next
// row), 0x143 = row_bcast:31 (lane 31 to rows 2 and 3).
template <int CTRL, int ROWS>
__device__ __forceinline__ unsigned long long dpp_or_zero(unsigned long long v) {  // lanes without a source read 0
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, ROWS, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, ROWS, 0xf, true);
    return ((unsigned long long)hi << 32) | lo;
}
template <int CTRL, int ROWS>
__device__ __forceinline__ int dpp_or_self(int v) {  // lanes without a source read their own value
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROWS, 0xf, false);
}
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_or_self(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)dpp_or_self<CTRL, ROWS>((int)(unsigned)u);
    const unsigned hi = (unsigned)dpp_or_self<CTRL, ROWS>((int)(unsigned)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ unsigned long long lane63(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_inclusive_scan(unsigned long long v, int /*lane*/) {
    v += dpp_or_zero<0x111, 0xf>(v);
    v += dpp_or_zero<0x112, 0xf>(v);
    v += dpp_or_zero<0x114, 0xf>(v);
    v += dpp_or_zero<0x118, 0xf>(v);
    v += dpp_or_zero<0x142, 0xa>(v);
    v += dpp_or_zero<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) { return lane63(wave_inclusive_scan(v, 0)); }
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_or_self<0x111, 0xf>(v));
    v = fmax(v, dpp_or_self<0x112, 0xf>(v));
    v = fmax(v, dpp_or_self<0x114, 0xf>(v));
    v = fmax(v, dpp_or_self<0x118, 0xf>(v));
    v = fmax(v, dpp_or_self<0x142, 0xa>(v));
    v = fmax(v, dpp_or_self<0x143, 0xc>(v));
    return __longlong_as_double((long long)lane63((unsigned long long)__double_as_longlong(v)));
}
__device__ __forceinline__ int wave_max_i32(int v) {
    v = max(v, dpp_or_self<0x111, 0xf>(v));
    v = max(v, dpp_or_self<0x112, 0xf>(v));
    v = max(v, dpp_or_self<0x114, 0xf>(v));
    v = max(v, dpp_or_self<0x118, 0xf>(v));
    v = max(v, dpp_or_self<0x142, 0xa>(v));
    v = max(v, dpp_or_self<0x143, 0xc>(v));
    return __builtin_amdgcn_readlane(v, 63);
}

// Component s of the initial condition (pop[i,:] .= initial_condition, hmm_particle_filter.jl:42-44).
template <class R>
__device__ __forceinline__ int model_x0(const DevModel& m, int s) {
    if constexpr (R::kBig) return R::x0(s);
    else return m.x0[s];
}

// The packed transition rows, held in VGPRs for the whole event loop: selecting between two SGPR operands costs two
// v_mov per event on top of the v_cndmask (one constant-bus read per VALU instruction on gfx9).  v[E-1] is the row
// of the last real event (the inverse-CDF search falls through to it).
template <class R>
struct TransRows {
    using TW = typename cond_t<(R::S <= 4), uint32_t, uint64_t>::type;
    TW v[R::kBig ? 1 : R::E];  // (big models apply their transition rows as constants: R::apply_event)
    __device__ __forceinline__ TransRows(const DevModel& m, int e_real) {
        if constexpr (R::kBig) {
            v[0] = 0;
            return;
        }
#pragma unroll
        for (int i = 0; i < (R::kBig ? 1 : R::E); ++i) v[i] = (TW)m.trans_packed[i];
        if constexpr (R::kGeneric) {  // padded evaluator: the last real row (selected with constant indices only)
#pragma unroll
            for (int i = 0; i < R::E - 1; ++i)
                if (i == e_real - 1) v[R::E - 1] = (TW)m.trans_packed[i];
        }
        if constexpr (!R::kGeneric && !R::kBig) {
#pragma unroll
            for (int i = 0; i < R::E; ++i) asm volatile("" : "+v"(v[i]));
        }
    }
};

// ---- one Gillespie event --------------------------------------------------------------------------------
// The body of iterate_particles! (hmm_particle_filter.jl:14-31): rates -> cumsum -> total == 0 ? -> time -= log(u)/total
// -> time > tmax ? -> event by inverse CDF (strict >, last event falls through; hmm_cmn.jl:4-10) -> state +=
// transition row -> event cap.  Returns true while the particle goes on.  FAST = the division may skip its range
// handling (dpomp_numerics.h: div_in_range); a total outside that range -- zero included, i.e. an extinct epidemic --
// stops the particle before anything is drawn, and the caller continues in the FAST = false loop, which is the same
// arithmetic with the plain division (finish_slow).  Why a particle stopped is read off its state afterwards
// (stopped_on_range): a reason code or flag carried through the loop costs VALU instructions on every event.
// The two uniforms of an event do not depend on the state, only the division by the rate total does.  They are
// therefore drawn -- and the logarithm of the first taken -- one event AHEAD: log(u) with its LDS table read then
// runs beside the rates -> reciprocal -> Newton chain of the current event instead of in front of the division, which
// shortens the dependent chain a lone wave walks per event (small filters, and the stragglers at the end of every
// launch).  Event n still consumes draws 2n and 2n+1 of the particle's stream; the pair drawn ahead when a particle
// stops is simply never used.
struct Draws {
    double lu, u2;  // log of the waiting-time uniform and the event-choice uniform of the NEXT event
};
__device__ __forceinline__ Draws draw_ahead(Stream& rng, const double (*s_log)[2], const LogRegs& lr) {
    Draws d;
    d.lu = dlog_tab(u01(stream_next(rng)), s_log, lr);
    d.u2 = u01(stream_next(rng));
    return d;
}

// COUNT: the event also bumps the caller's per-lane counter and the return value folds in the event cap (n < cap).
// The kernels that run where a launch does not fill the chip use it: there a wave's own latency per event counts, and
// the counter as two VALU instructions inside the one exit condition is 45 cycles per event shorter than the uniform
// trip counter's scalar bookkeeping on a further loop exit (which the chip-filling compaction kernel hides).
template <class R, bool FAST, bool COUNT = false>
__device__ __forceinline__ bool gillespie_event(const DevModel& m, const TransRows<R> rows, const double* thr,
                                                int (&x)[R::S], double& time, const double tmax, Stream& rng, Draws& d,
                                                const int e_real, const double (*s_log)[2], const LogRegs& lr,
                                                int* n = nullptr, const int cap = 0) {
    constexpr int S = R::S, E = R::E;
    using TW = typename TransRows<R>::TW;
    double cum[E];
    R::rates(m, thr, x, cum);
#pragma unroll
    for (int e = 1; e < E; ++e) cum[e] = cum[e - 1] + cum[e];
    const double tot = cum[E - 1];
    if (FAST) {
        if (!rate_in_fast_range(tot)) return false;
    } else {
        if (tot == 0.0) return false;
    }
    const Draws now = d;
    d = draw_ahead(rng, s_log, lr);
    time = time - (FAST ? div_in_range(now.lu, tot) : now.lu / tot);
    if (time > tmax) return false;
    const double etc = now.u2 * tot;
    if constexpr (R::kBig) {  // the event's index by the same inverse-CDF chain, its row as constants of the evaluator
        int ev = E - 1;
#pragma unroll
        for (int i = E - 2; i >= 0; --i)
            if (cum[i] > etc) ev = i;
        R::apply_event(ev, x);
    } else {
        TW tw = rows.v[E - 1];
#pragma unroll
        for (int i = E - 2; i >= 0; --i) {
            const bool in_range = R::kGeneric ? (i < e_real - 1) : true;
            if (in_range && cum[i] > etc) tw = rows.v[i];
        }
#pragma unroll
        for (int s = 0; s < S; ++s) x[s] += (int)(int8_t)(tw >> (8 * s));
    }
    if constexpr (COUNT) return ++*n < cap;
    return true;
}
// Counting events costs nothing in the loop: the lanes of a wave start an interval together and every lane that is
// still in the fast loop has executed exactly one event per iteration, so the event cap is a comparison on the
// wave-uniform trip counter (scalar unit), and a particle's own count is read off its stream afterwards -- the stream
// counts its draws: one pair drawn ahead at the start and one more by every call that got past the rate test, and the
// call that overshot the observation time drew without executing.
__device__ __forceinline__ int events_executed(const Stream& rng, double time, double tmax) {
    const int pairs = (int)(((uint32_t)rng.n - (uint32_t)(1 + kStreamWarmup)) >> 1);
    return pairs - 1 - (time > tmax ? 1 : 0);
}
// after a FAST loop returned false: the clock did not stop the particle, so its rate total did
__device__ __forceinline__ bool stopped_on_range(double time, double tmax) { return !(time > tmax); }
// The plain-division loop for a particle whose rate total left the fast range (zero total = extinct: returns at once):
// runs the particle to the end of the interval with its own event count.  Returns true if the event cap stopped it.
template <class R>
__device__ __forceinline__ bool finish_slow(const DevModel& m, const TransRows<R> rows, const double* thr, int (&x)[R::S],
                                            double& time, const double tmax, Stream& rng, Draws& d, const int cap,
                                            const int e_real, const double (*s_log)[2], const LogRegs& lr) {
    int n = events_executed(rng, time, tmax);
    if (n >= cap) return true;
    while (gillespie_event<R, false>(m, rows, thr, x, time, tmax, rng, d, e_real, s_log, lr))
        if (++n >= cap) return true;
    return false;
}

// ---- k_zero: clears counters, flags and the control block (in place of hipMemsetAsync) -------------------
__global__ __launch_bounds__(kBlock) void k_zero(uint32_t* dst, int64_t words) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < words) dst[i] = 0u;
}

// ---- k_begin: the first kernel of a filter call -- clears the control block and, for small batches, stores the
// parameter vectors that arrived as kernel arguments (no host-to-device copy in front of the call) -----------
constexpr int kThetaInline = 32;  // doubles
struct ThetaInline {
    double v[kThetaInline];
};
__global__ __launch_bounds__(kBlock) void k_begin(uint32_t* ctl, int64_t words, double* theta_dst, ThetaInline th, int n_theta) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < words) ctl[i] = 0u;
    if (blockIdx.x == 0 && (int)threadIdx.x < n_theta) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < kThetaInline; ++k)  // (constant indices: the argument stays in SGPRs)
            if (k == (int)threadIdx.x) v = th.v[k];
        theta_dst[threadIdx.x] = v;
    }
}

template <class Ptr>
__device__ __forceinline__ int first_greater(Ptr N, int n, int i) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (N[mid] > i) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// Ancestor of slot p under systematic resampling = first j with N_j > p (rsp_systematic's `while u[i] > cw[j]`,
// hmm_pf_resample.jl:33-37, solved per slot).  N is monotone, so the 256 slots of a workgroup descend from a contiguous
// window of parent blocks [first_parent[w], first_parent[w+1]] (written by k_offsets).  The workgroup stages that window
// of N in LDS (`stage`, kStageSlots ints, coalesced) and every lane searches there (11 LDS steps); a wider window (very
// uneven weights) is searched in L2 with a bounded binary search.  All threads of the workgroup must call this
// (barriers); returns the ancestor of thread's slot (0 for lanes past P).
__device__ __forceinline__ int find_ancestor(const PropArgs& a, int b, int p, bool valid, int32_t* stage) {
    if (a.noresample && a.noresample[b]) return valid ? p : 0;  // (workgroup-uniform)
    const int nblk = (int)(a.Pp / kParentBlock);
    const int32_t* N = a.offs + (int64_t)b * a.Pp;
    const int32_t* fp = a.first_parent + (int64_t)b * nblk;
    const int w = blockIdx.x;  // this workgroup's 256-slot block
    const int last = (a.P - 1) / kParentBlock;
    const int tlo = fp[w];
    const int thi = (w + 1 <= last) ? fp[w + 1] : last;  // ancestors are monotone in the slot index
    const int nb = thi - tlo + 1;
    int j;
    if (nb <= kStageBlocks) {
        const int M = nb * kParentBlock;
        for (int t = threadIdx.x; t < M; t += kBlock) stage[t] = N[(int64_t)tlo * kParentBlock + t];
        __syncthreads();
        int lo = 0;  // number of staged boundaries <= p
#pragma unroll
        for (int step = 1024; step > 0; step >>= 1) {
            const int mid = lo + step;
            if (mid <= M && stage[mid - 1] <= p) lo = mid;
        }
        j = tlo * kParentBlock + lo;
        __syncthreads();  // the staging area is reused by the caller
    } else {
        const int base = tlo * kParentBlock;
        j = valid ? base + first_greater(N + base, min(nb * kParentBlock, a.P - base), p) : 0;
    }
    return valid ? min(j, a.P - 1) : 0;
}

// ---- k_init: pop[i,:] .= initial_condition   (hmm_particle_filter.jl:42-44) -----------------
__global__ __launch_bounds__(kBlock) void k_init(DevModel m, int32_t* state, int64_t Pp) {
    const int64_t b = blockIdx.y;
    const int64_t p0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kItems;
    if (p0 >= Pp) return;
    for (int s = 0; s < m.S; ++s) {
        const int v = m.x0[s];
        *reinterpret_cast<int4*>(state + (b * m.S + s) * Pp + p0) = make_int4(v, v, v, v);
    }
}

// ---- k_propagate: one particle per lane ------------------------------------------------------
// Gillespie direct method from t_prev to t_max with state and clock in VGPRs, then the
// observation log-weight.  Per event: rates -> cumsum -> total == 0 ? -> time -= log(u)/total
// -> time > tmax ? -> event by inverse CDF (strict >, last event falls through) -> state +=
// transition row.  Epilogue: state and log-weight back to HBM, workgroup max of the
// log-weights into segmax (one 64-bit atomicMax per workgroup), event count.
template <class R>
__global__ __launch_bounds__(kBlock) void k_propagate(PropArgs a) {
    constexpr int S = R::S, E = R::E;
    using TW = typename cond_t<(S <= 4), uint32_t, uint64_t>::type;
    const int b = blockIdx.y;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = p < a.P;
    const int s_real = R::kGeneric ? a.m.S : S;  // RGeneric pads S up to 2/4/8: rows >= s_real do not exist
    int32_t* st = a.state + (int64_t)b * s_real * a.Pp;
    const double* th = a.theta + (int64_t)b * a.m.K;
    const ThetaRegs<R> th_regs(th, a.m.K);
    const double* thr = th_regs.get(th);  // for the rate evaluator inside the event loop

    __shared__ double s_log[kLogTableSize][2];  // dlog table: one ds_read_b128 per event
    __shared__ int32_t s_stage[kStageSlots];    // offspring boundaries of the parents' window (find_ancestor)
    for (int i = threadIdx.x; i < 2 * kLogTableSize; i += kBlock) (&s_log[0][0])[i] = (&kLogTable[0][0])[i];
    const int32_t* ld = st;  // where this slot's starting state is read from
    int pj = p;
    if (a.parents) {         // resampling fused into the load
        ld = a.parents + (int64_t)b * s_real * a.Pp;
        pj = find_ancestor(a, b, p, valid, s_stage);
    }
    __syncthreads();

    int x[S];
#pragma unroll
    for (int s = 0; s < S; ++s) x[s] = (valid && s < s_real) ? (a.from_x0 ? model_x0<R>(a.m, s) : ld[(int64_t)s * a.Pp + pj]) : 0;

    double time = a.use_theta_t0 ? th[a.m.t0_index - 1] : a.t_prev;
    const double tmax = a.t_max;
    Stream rng = stream_init(a.seed, a.seg_base + (uint32_t)b, (uint32_t)p, a.obs_k, a.domain);
    const int e_real = R::kGeneric ? a.m.E : E;
    const TransRows<R> rows(a.m, e_real);
    const LogRegs lr = log_regs();  // (a polynomial constant of the logarithm, in a VGPR pair for the whole kernel)
    const int cap = a.m.cap;
    int n = 0;
    bool capped = false;

    if (valid) {
        Draws ahead = draw_ahead(rng, s_log, lr);
        DPOMP_LOOP_AT(DPOMP_PAD_STATIC);
        for (;;) {  // (two events per trip: see DPOMP_LOOP_AT)
            if (!gillespie_event<R, true, true>(a.m, rows, thr, x, time, tmax, rng, ahead, e_real, s_log, lr, &n, cap)) break;
            if (!gillespie_event<R, true, true>(a.m, rows, thr, x, time, tmax, rng, ahead, e_real, s_log, lr, &n, cap)) break;
        }
        capped = n >= cap;
        if (!capped && stopped_on_range(time, tmax))
            capped = finish_slow<R>(a.m, rows, thr, x, time, tmax, rng, ahead, cap, e_real, s_log, lr);
        n = events_executed(rng, time, tmax);
    }

    double g = -__builtin_inf();
    if (valid) {
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (s < s_real) st[(int64_t)s * a.Pp + p] = x[s];
        if (a.want_weight) {
            g = obs_log_weight<R>(a.m, th, x, a.ysum, a.yval, a.n_yval);
            if (a.carry_in && a.carry_in[b]) g = g + a.logw[(int64_t)b * a.Pp + p];  // t_p = g_p + a_p
            if (capped || !(g > -__builtin_inf())) g = -__builtin_inf();
            a.logw[(int64_t)b * a.Pp + p] = g;
        }
    }

    __shared__ double s_max[kBlock / 64];
    __shared__ unsigned long long s_ev[kBlock / 64];
    __shared__ int s_cap[kBlock / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __shared__ int s_nmax[kBlock / 64];
    const double wm = wave_max(g);
    const unsigned long long wev = wave_sum((unsigned long long)n);
    const int wnmax = wave_max_i32(n);
    const int wcap = wave_any(capped);
    if (lane == 0) {
        s_max[wid] = wm;
        s_ev[wid] = wev;
        s_cap[wid] = wcap;
        s_nmax[wid] = wnmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double bm = s_max[0];
        unsigned long long be = s_ev[0], bs = (unsigned long long)s_nmax[0];
        int bc = s_cap[0];
#pragma unroll
        for (int w = 1; w < kBlock / 64; ++w) {
            bm = fmax(bm, s_max[w]);
            be += s_ev[w];
            bs += (unsigned long long)s_nmax[w];
            bc |= s_cap[w];
        }
        if (a.want_weight && bm > -__builtin_inf()) publish_max(&a.segmax[b], ordered_key(bm));
        if (bc) atomicOr(&a.status[b], 1u /* DPOMP_ST_EVENT_CAP */);
        if (be) {  // events of this interval, and what a one-particle-per-lane schedule makes of it (dpomp_get_lane_utilisation)
            sched_add(a.sched, be, 64ull * bs);
        }
    }
}

// ---- k_propagate_q: the same computation with dynamic lane refill --------------------------------
// One particle per lane wastes lanes: event counts per interval vary (Pooley SIS: 52 +- 50 in the
// first interval, 269 +- 16 later), so a wave runs until its slowest lane is done (measured lane
// utilisation 79 %).  Here a workgroup owns a chunk of C = 256*RPL particles.  Phase 0 stages the
// chunk in LDS with coalesced loads: state tile [S][C] and the Philox-seeded stream state of every
// particle (full lane occupancy for the 10 Philox rounds).  Phase 1: every kQueueChunk events the lanes of a wave
// that have finished their particle pop the next ones from the workgroup's LDS queue (one LDS atomic per wave, 2 LDS
// reads per lane, no global memory, no Philox; no barrier), and write the final state back to the LDS tile.  Phase 2: weights, coalesced write-back and
// the reductions, again at full occupancy.  Which lane runs which particle is irrelevant to the
// result because every draw is keyed by the particle index: still bit-identical to the oracle.
template <class R, int RPL>
__global__ __launch_bounds__(kBlock) void k_propagate_q(PropArgs a) {
    constexpr int S = R::S, E = R::E;
    constexpr int C = kBlock * RPL;
    using TW = typename cond_t<(S <= 4), uint32_t, uint64_t>::type;
    __shared__ double s_log[kLogTableSize][2];
    __shared__ int32_t s_x[S][C];
    __shared__ uint4 s_seed[C];
    __shared__ int32_t s_cnt[C];  // executed events, bit 31 = hit the event cap
    __shared__ int s_next;
    __shared__ double s_max[kBlock / 64];
    __shared__ unsigned long long s_ev[kBlock / 64];
    __shared__ int s_cap[kBlock / 64];
    __shared__ int s_nmax[kBlock / 64];

    const int b = blockIdx.y;
    const int base = blockIdx.x * C;
    const int cnt = min(C, a.P - base);
    const int s_real = R::kGeneric ? a.m.S : S;
    int32_t* st = a.state + (int64_t)b * s_real * a.Pp;
    const double* th = a.theta + (int64_t)b * a.m.K;
    const ThetaRegs<R> th_regs(th, a.m.K);
    const double* thr = th_regs.get(th);  // for the rate evaluator inside the event loop

    for (int i = threadIdx.x; i < 2 * kLogTableSize; i += kBlock) (&s_log[0][0])[i] = (&kLogTable[0][0])[i];
#pragma unroll
    for (int jj = 0; jj < RPL; ++jj) {
        const int idx = jj * kBlock + threadIdx.x;
        if (idx < cnt) {
            const int p = base + idx;
#pragma unroll
            for (int s = 0; s < S; ++s) s_x[s][idx] = (s < s_real) ? (a.from_x0 ? model_x0<R>(a.m, s) : st[(int64_t)s * a.Pp + p]) : 0;
            const StreamSeed r = stream_seed(a.seed, a.seg_base + (uint32_t)b, (uint32_t)p, a.obs_k, a.domain);
            s_seed[idx] = make_uint4(r.w0, r.w1, r.w2, r.w3);
        }
    }
    if (threadIdx.x == 0) s_next = 0;
    __syncthreads();

    const double t_start = a.use_theta_t0 ? th[a.m.t0_index - 1] : a.t_prev;
    const double tmax = a.t_max;
    const int e_real = R::kGeneric ? a.m.E : E;
    const TransRows<R> rows(a.m, e_real);
    const LogRegs lr = log_regs();  // (a polynomial constant of the logarithm, in a VGPR pair for the whole kernel)
    const int cap = a.m.cap;
    int x[S];
#pragma unroll
    for (int s = 0; s < S; ++s) x[s] = 0;
    Stream rng{0, 0, 0, 0};
    Draws ahead{0.0, 0.0};
    double time = 0.0;
    int n = 0, cur = 0;
    bool active = false, exhausted = false;
    const int lane_q = threadIdx.x & 63;
    for (;;) {
        // refill: the lanes of a wave that are out of work take the next particles of the workgroup's queue with one
        // LDS atomic per wave
        const unsigned long long need = wave_ballot(!active && !exhausted);
        if (need) {
            int first = 0;
            if (lane_q == (int)__builtin_ctzll(need)) first = atomicAdd(&s_next, (int)__popcll(need));
            first = __shfl(first, (int)__builtin_ctzll(need), 64);
            if (!active && !exhausted) {
                const int idx = first + (int)__popcll(need & ((1ull << lane_q) - 1ull));
                if (idx < cnt) {
                    cur = idx;
#pragma unroll
                    for (int s = 0; s < S; ++s) x[s] = s_x[s][idx];
                    const uint4 sd = s_seed[idx];
                    rng = stream_start(StreamSeed{sd.x, sd.y, sd.z, sd.w});
                    ahead = draw_ahead(rng, s_log, lr);
                    time = t_start;
                    n = 0;
                    active = true;
                } else {
                    exhausted = true;
                }
            }
        }
        if (!wave_any(active)) break;
        if (active) {  // up to QCHK events, then back to the refill point (a finished lane idles for the rest of the chunk)
            int it = 0;
            bool go;
            bool capped_now = false;  // (the lanes of a wave start their particles at different times: counted per lane)
            do {
                go = gillespie_event<R, true>(a.m, rows, thr, x, time, tmax, rng, ahead, e_real, s_log, lr);
                if (go && ++n >= cap) {
                    capped_now = true;
                    go = false;
                }
            } while (go && ++it < kQueueChunk);
            if (!go && !capped_now && stopped_on_range(time, tmax))
                capped_now = finish_slow<R>(a.m, rows, thr, x, time, tmax, rng, ahead, cap, e_real, s_log, lr);
            if (!go) {
#pragma unroll
                for (int s = 0; s < S; ++s) s_x[s][cur] = x[s];
                s_cnt[cur] = events_executed(rng, time, tmax) | (capped_now ? (int)0x80000000 : 0);
                active = false;
            }
        }
    }
    __syncthreads();

    double gmax = -__builtin_inf();
    unsigned long long ev = 0;
    bool any_cap = false;
    int nmax_sum = 0;  // sum over this wave's RPL slot groups of the largest event count: the static schedule's waves
#pragma unroll
    for (int jj = 0; jj < RPL; ++jj) {
        const int idx = jj * kBlock + threadIdx.x;
        int pn = 0;
        if (idx < cnt) {
            const int p = base + idx;
            int xv[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                xv[s] = s_x[s][idx];
                if (s < s_real) st[(int64_t)s * a.Pp + p] = xv[s];
            }
            const int c = s_cnt[idx];
            ev += (unsigned long long)(c & 0x7FFFFFFF);
            pn = c & 0x7FFFFFFF;
            const bool capped = c < 0;
            any_cap |= capped;
            if (a.want_weight) {
                double g = obs_log_weight<R>(a.m, th, xv, a.ysum, a.yval, a.n_yval);
                if (a.carry_in && a.carry_in[b]) g = g + a.logw[(int64_t)b * a.Pp + p];  // t_p = g_p + a_p
                if (capped || !(g > -__builtin_inf())) g = -__builtin_inf();
                a.logw[(int64_t)b * a.Pp + p] = g;
                gmax = fmax(gmax, g);
            }
        }
        nmax_sum += wave_max_i32(pn);
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const double wm = wave_max(gmax);
    const unsigned long long wev = wave_sum(ev);
    const int wcap = wave_any(any_cap);
    if (lane == 0) {
        s_max[wid] = wm;
        s_ev[wid] = wev;
        s_cap[wid] = wcap;
        s_nmax[wid] = nmax_sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double bm = s_max[0];
        unsigned long long be = s_ev[0], bs = (unsigned long long)s_nmax[0];
        int bc = s_cap[0];
#pragma unroll
        for (int w = 1; w < kBlock / 64; ++w) {
            bm = fmax(bm, s_max[w]);
            be += s_ev[w];
            bs += (unsigned long long)s_nmax[w];
            bc |= s_cap[w];
        }
        if (a.want_weight && bm > -__builtin_inf()) publish_max(&a.segmax[b], ordered_key(bm));
        if (bc) atomicOr(&a.status[b], 1u /* DPOMP_ST_EVENT_CAP */);
        if (be) {
            sched_add(a.sched, be, 64ull * bs);
        }
    }
}

// MODE of k_propagate_c: which particles the 256 slots of a workgroup hold.
//   kModeBlock  slots blockIdx.x * 256 .. of filter blockIdx.y (filters of many particles);
//   kModeFlat   the filters of a call laid end to end: flat index f = blockIdx.x * 256 + slot, filter f / P, particle
//               f % P -- filters of a few hundred particles (the reference's default is 200, C_DF_PF_P) then fill every
//               lane of every workgroup instead of leaving the last wave of each filter mostly empty.  A workgroup
//               then holds particles of several filters: the parameter vector is per lane (re-read when a compaction
//               moves a context into the lane) and the per-filter reductions of the epilogue are segmented.
enum : int { kModeBlock = 0, kModeFlat = 1 };
constexpr int kFlatMaxP = 512;  // kModeFlat: particles per filter (ancestor windows of 256/P + 2 filters fit the stage)

// number of set bits of a wave mask below the calling lane
__device__ __forceinline__ int lanes_below(unsigned long long mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// kModeFlat ancestor search: the workgroup's slots span the filters b_lo..b_hi; their offspring boundaries (P each) are
// staged in LDS and every lane searches its own filter's segment (first j with N_j > p, as find_ancestor).  All
// threads must call (barriers).
__device__ __forceinline__ int find_ancestor_flat(const PropArgs& a, int b, int p, bool valid, int32_t* stage) {
    const uint32_t f0 = blockIdx.x * kBlock, total = (uint32_t)a.B * (uint32_t)a.P;
    const int b_lo = (int)(f0 / (uint32_t)a.P);
    const int b_hi = (int)((min(f0 + kBlock, total) - 1u) / (uint32_t)a.P);
    const int M = (b_hi - b_lo + 1) * a.P;  // <= 255 + 2 P <= kStageSlots
    for (int t = threadIdx.x; t < M; t += kBlock) {
        const int fb = t / a.P;
        stage[t] = a.offs[(int64_t)(b_lo + fb) * a.Pp + (t - fb * a.P)];
    }
    __syncthreads();
    int j = 0;
    if (valid) j = first_greater(stage + (b - b_lo) * a.P, a.P, p);
    __syncthreads();  // the staging area is reused by the caller
    if (valid && a.noresample && a.noresample[b]) j = p;
    return valid ? min(j, a.P - 1) : 0;
}

// ---- k_propagate_c: one particle per lane + workgroup-level compaction ------------------------------
// Lanes whose particle has reached t_max idle until the slowest lane of the wave is done, and a
// wave with 5 live lanes costs the VALU as much as one with 64.  Every CHK events the workgroup
// counts its live lanes (ballot + 4 LDS words + one barrier); when they fit in fewer waves than
// are still running, the live contexts (state, stream, clock, event count, particle slot) are
// packed through LDS into the lowest lanes of the workgroup and the emptied waves only keep
// meeting the barrier.  Finished particles park their final state in an LDS tile, and the
// epilogue (weights, coalesced write-back, reductions) runs at full occupancy.  The arithmetic of
// a particle does not depend on the lane it runs in, so results stay bit-identical.
template <class R, int CHK, int MODE = kModeBlock>
__global__ __launch_bounds__(kBlock, 8) void k_propagate_c(PropArgs a) {  // 8 waves per SIMD: at most 64 VGPRs
    constexpr int S = R::S, E = R::E;
    constexpr bool kFlat = MODE == kModeFlat;
    using TW = typename cond_t<(S <= 4), uint32_t, uint64_t>::type;
    __shared__ double s_log[kLogTableSize][2];
    __shared__ int32_t s_res[S][kBlock];  // final state by particle slot
    __shared__ int32_t s_cnt[kBlock];     // executed events, bit 31 = event cap
    // Compaction staging.  A compaction frees at least one wave, so at most kBlock - 64 contexts move; they go
    // through one 6 KB buffer in three passes of 32 bytes per context (stream words; clock, counters and the draws made
    // ahead; state) to keep the workgroup's LDS small enough for 8 workgroups per CU whatever the model.
    constexpr int kLive = kBlock - 64;
    constexpr int kCtxBytes = 32;  // per pass: stream words (32), clock + counters + draws ahead (32), state (4 S <= 32)
    __shared__ __attribute__((aligned(16))) unsigned char c_raw[kCtxBytes * kLive];
    uint4* const c_rng0 = reinterpret_cast<uint4*>(c_raw);
    uint4* const c_rng1 = c_rng0 + kLive;
    double* const c_time = reinterpret_cast<double*>(c_raw);
    int2* const c_nc = reinterpret_cast<int2*>(c_raw + 8 * kLive);
    double2* const c_ahead = reinterpret_cast<double2*>(c_raw + 16 * kLive);  // the draws made one event ahead
    int32_t* const c_x = reinterpret_cast<int32_t*>(c_raw);                    // [S][kLive], third pass
    __shared__ int s_wcnt[2][kBlock / 64];
    __shared__ double s_max[kBlock / 64];
    __shared__ unsigned long long s_ev[kBlock / 64];
    __shared__ int s_cap[kBlock / 64];

#ifdef DPOMP_TRACE_WG
    const unsigned long long tr_t0 = wall_clock64();
    unsigned long long tr_t1 = 0, tr_x1 = 0, tr_x2 = 0, tr_x3 = 0;  // loop start; when the 2nd / 3rd / 4th wave emptied
#endif
    // (filter, particle) of the particle that STARTS in local slot `slot`; false past the end
    auto locate = [&](int slot, int& fb, int& fp) -> bool {
        if constexpr (kFlat) {
            const uint32_t f = blockIdx.x * kBlock + (uint32_t)slot;
            const bool ok = f < (uint32_t)a.B * (uint32_t)a.P;
            fb = ok ? (int)(f / (uint32_t)a.P) : 0;
            fp = ok ? (int)(f - (uint32_t)fb * (uint32_t)a.P) : 0;
            return ok;
        } else {
            fb = blockIdx.y;
            fp = blockIdx.x * kBlock + slot;
            return fp < a.P;
        }
    };
    int b, p;
    const bool valid = locate(threadIdx.x, b, p);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int s_real = R::kGeneric ? a.m.S : S;
    int32_t* st = a.state + (int64_t)b * s_real * a.Pp;
    const double* th = a.theta + (int64_t)b * a.m.K;
    typename cond_t<kFlat, ThetaRegs<R>, const ThetaRegs<R>>::type th_regs(th, a.m.K);
    const double* thr = th_regs.get(th);  // for the rate evaluator inside the event loop

    for (int i = threadIdx.x; i < 2 * kLogTableSize; i += kBlock) (&s_log[0][0])[i] = (&kLogTable[0][0])[i];
    static_assert(kStageSlots * 4 <= kCtxBytes * kLive, "the ancestor search stages its window in the compaction buffer");
    const int32_t* ld = st;  // where this slot's starting state is read from
    int pj = p;
    if (a.parents) {         // resampling fused into the load (the compaction buffer is free until the first compaction)
        ld = a.parents + (int64_t)b * s_real * a.Pp;
        pj = kFlat ? find_ancestor_flat(a, b, p, valid, reinterpret_cast<int32_t*>(c_raw))
                   : find_ancestor(a, b, p, valid, reinterpret_cast<int32_t*>(c_raw));
    }

    int x[S];
#pragma unroll
    for (int s = 0; s < S; ++s) x[s] = (valid && s < s_real) ? (a.from_x0 ? model_x0<R>(a.m, s) : ld[(int64_t)s * a.Pp + pj]) : 0;
    double time = a.use_theta_t0 ? th[a.m.t0_index - 1] : a.t_prev;
    const double tmax = a.t_max;
    Stream rng = stream_init(a.seed, a.seg_base + (uint32_t)b, (uint32_t)p, a.obs_k, a.domain);
    const int e_real = R::kGeneric ? a.m.E : E;
    const TransRows<R> rows(a.m, e_real);
    const LogRegs lr = log_regs();  // (a polynomial constant of the logarithm, in a VGPR pair for the whole kernel)
    const int cap = a.m.cap;
    int cur = threadIdx.x;
    // particles this workgroup starts with
    const int n_start = kFlat ? (int)min((uint32_t)kBlock, (uint32_t)a.B * (uint32_t)a.P - blockIdx.x * kBlock)
                              : min(kBlock, a.P - (int)blockIdx.x * kBlock);
    // Events until the next count of live lanes (workgroup-uniform).  A count is a barrier for four waves, and it can
    // only pay once particles have finished: CHK events at first, twice that while every particle of the workgroup still
    // runs, 3/8 CHK once they have started to finish (the finer the packing, the fewer lanes idle in a live wave; below
    // that the barriers cost more than the lanes they free), and no further count when one wave holds all that is left.
    constexpr int kChunkMax = 2 * CHK, kChunkDie = 3 * CHK / 8, kChunkTail = 1 << 24;
    int chunk = CHK;
    int ev_base = 0;      // wave-uniform: events every still-running particle of the workgroup has executed
    bool active = valid;  // particle still running
    bool parked = !valid; // final state already parked in s_res
    bool capped = false;
    __syncthreads();
    Draws ahead = draw_ahead(rng, s_log, lr);  // (after the barrier: reads the log table in LDS)
#ifdef DPOMP_TRACE_WG
    tr_t1 = wall_clock64();
#endif

    // Head start for the workgroups that are dispatched last.  The hardware serves the oldest waves of a SIMD first, so
    // without it a launch ends with every CU working off its youngest workgroups, which have most of their event
    // chains left, at one or two live waves per SIMD (a lone wave is bound by instruction latency, not issue).  At
    // raised priority for the first ~70 % of its chain, the youngest half of the last round catches up with the older
    // half and the workgroups of a CU finish together.
    bool boosted = a.boost_events > 0 && (int)(blockIdx.y * gridDim.x + blockIdx.x) >= a.boost_from;
    if (boosted) __builtin_amdgcn_s_setprio(3);
    for (int round = 0;; ++round) {
        if (boosted && ev_base >= a.boost_events) {  // (wave-uniform)
            boosted = false;
            __builtin_amdgcn_s_setprio(0);
        }
        if (active) {  // up to CHK events per live lane: the static kernel's divergent loop plus a uniform trip counter
            // every live lane of the workgroup has executed ev_base events so far (compaction moves contexts, not counts)
            capped = false;  // (this lane may hold another particle's context than last round)
            // (>= 1: a lane that reaches the cap stops being active; read as a scalar so that the trip test stays off the VALU)
            const int lim = __builtin_amdgcn_readfirstlane(min(chunk, cap - ev_base));
            int it = 0;
            bool go;
            do go = gillespie_event<R, true>(a.m, rows, thr, x, time, tmax, rng, ahead, e_real, s_log, lr);
            while (go && ++it < lim);
            if (go && ev_base + lim >= cap) {  // (uniform) still going after `cap` events
                capped = true;
                go = false;
            } else if (!go && stopped_on_range(time, tmax)) {
                capped = finish_slow<R>(a.m, rows, thr, x, time, tmax, rng, ahead, cap, e_real, s_log, lr);
            }
            active = go;
        }
        ev_base += min(chunk, cap - ev_base);  // (never past the cap: no overflow whatever the cap)
        if (!active && !parked) {  // once per particle, outside the hot loop
#pragma unroll
            for (int s = 0; s < S; ++s) s_res[s][cur] = x[s];
            s_cnt[cur] = events_executed(rng, time, tmax) | (capped ? (int)0x80000000 : 0);
            parked = true;
        }
        const unsigned long long bal = wave_ballot(active);
        if (lane == 0) s_wcnt[round & 1][wid] = __popcll(bal);
        __syncthreads();
        const int c0 = s_wcnt[round & 1][0], c1 = s_wcnt[round & 1][1], c2 = s_wcnt[round & 1][2],
                  c3 = s_wcnt[round & 1][3];
        const int total = __builtin_amdgcn_readfirstlane((c0 + c1) + (c2 + c3));  // (uniform: keeps the chunk length scalar)
        if (total == 0) break;
        chunk = total < n_start ? (total <= 64 ? kChunkTail : kChunkDie) : kChunkMax;
        const int waves_now = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
        if (((total + 63) >> 6) < waves_now) {  // workgroup-uniform
#ifdef DPOMP_TRACE_WG
            {
                const unsigned long long now = wall_clock64();
                const int need = (total + 63) >> 6;
                if (need < 4 && !tr_x3) tr_x3 = now;
                if (need < 3 && !tr_x2) tr_x2 = now;
                if (need < 2 && !tr_x1) tr_x1 = now;
            }
#endif
            const int rank = (wid > 0 ? c0 : 0) + (wid > 1 ? c1 : 0) + (wid > 2 ? c2 : 0) +
                             __popcll(bal & ((1ull << lane) - 1ull));
            const bool take = (int)threadIdx.x < total;
            if (active) {
                c_rng0[rank] = make_uint4((uint32_t)rng.a, (uint32_t)(rng.a >> 32), (uint32_t)rng.b, (uint32_t)(rng.b >> 32));
                c_rng1[rank] = make_uint4((uint32_t)rng.c, (uint32_t)(rng.c >> 32), (uint32_t)rng.n, (uint32_t)(rng.n >> 32));
            }
            __syncthreads();
            Stream moved = rng;
            if (take) {
                const uint4 r4 = c_rng0[threadIdx.x], r5 = c_rng1[threadIdx.x];
                moved.a = (uint64_t)r4.x | ((uint64_t)r4.y << 32);
                moved.b = (uint64_t)r4.z | ((uint64_t)r4.w << 32);
                moved.c = (uint64_t)r5.x | ((uint64_t)r5.y << 32);
                moved.n = (uint64_t)r5.z | ((uint64_t)r5.w << 32);
            }
            __syncthreads();
            if (active) {  // pass 2: clock, counters, the draws made ahead
                c_time[rank] = time;
                c_nc[rank] = make_int2(0, cur);
                c_ahead[rank] = make_double2(ahead.lu, ahead.u2);
            }
            __syncthreads();
            double moved_time = time;
            Draws moved_ahead = ahead;
            int2 moved_nc = make_int2(0, cur);
            if (take) {
                moved_time = c_time[threadIdx.x];
                const double2 ah = c_ahead[threadIdx.x];
                moved_ahead.lu = ah.x;
                moved_ahead.u2 = ah.y;
                moved_nc = c_nc[threadIdx.x];
            }
            __syncthreads();
            if (active) {  // pass 3: the state
#pragma unroll
                for (int s = 0; s < S; ++s) c_x[s * kLive + rank] = x[s];
            }
            __syncthreads();
            active = take;
            parked = !active;  // every lane without a live context has already parked (or never had) its particle
            if (active) {
#pragma unroll
                for (int s = 0; s < S; ++s) x[s] = c_x[s * kLive + threadIdx.x];
                rng = moved;
                time = moved_time;
                ahead = moved_ahead;
                cur = moved_nc.y;
                if constexpr (kFlat) {  // the parameters of the filter the moved-in particle belongs to
                    int mb, mp;
                    locate(cur, mb, mp);
                    th = a.theta + (int64_t)mb * a.m.K;
                    th_regs = ThetaRegs<R>(th, a.m.K);
                    thr = th_regs.get(th);
                }
            }
        }
    }
    __syncthreads();

    // thread t closes the particle that STARTED in slot t
    double g = -__builtin_inf();
    unsigned long long ev = 0;
    capped = false;
    if constexpr (kFlat) {  // (re-derived here rather than kept in registers across the event loop; the lane may also
                            // have run particles of other filters in between)
        locate(threadIdx.x, b, p);
        st = a.state + (int64_t)b * s_real * a.Pp;
        th = a.theta + (int64_t)b * a.m.K;
    }
    if (valid) {
        int xv[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            xv[s] = s_res[s][threadIdx.x];
            if (s < s_real) st[(int64_t)s * a.Pp + p] = xv[s];
        }
        const int c = s_cnt[threadIdx.x];
        ev = (unsigned long long)(c & 0x7FFFFFFF);
        capped = c < 0;
        if (a.want_weight) {
            g = obs_log_weight<R>(a.m, th, xv, a.ysum, a.yval, a.n_yval);
            if (a.carry_in && a.carry_in[b]) g = g + a.logw[(int64_t)b * a.Pp + p];  // t_p = g_p + a_p
            if (capped || !(g > -__builtin_inf())) g = -__builtin_inf();
            a.logw[(int64_t)b * a.Pp + p] = g;
        }
    }
    double wm = -__builtin_inf();
    int wcap = 0;
    if constexpr (kFlat) {
        // the particles of a wave belong to several filters: one maximum per filter present (ballot loop), flags per lane
        if (capped) atomicOr(&a.status[b], 1u /* DPOMP_ST_EVENT_CAP */);
        unsigned long long todo = wave_ballot(a.want_weight && g > -__builtin_inf());
        while (todo) {  // (wave-uniform)
            const int leader = (int)__builtin_ctzll(todo);
            const int bb = __builtin_amdgcn_readlane(b, leader);
            const bool same = b == bb;
            const double m = wave_max(same ? g : -__builtin_inf());
            if (lane == leader) publish_max(&a.segmax[bb], ordered_key(m));
            todo &= ~wave_ballot(same);
        }
    } else {
        wm = wave_max(g);
        wcap = wave_any(capped);
    }
    const unsigned long long wev = wave_sum(ev);
    const int wnmax = wave_max_i32((int)ev);  // particle slots in launch order: the static schedule's waves
    if (lane == 0) {
        s_max[wid] = wm;
        s_ev[wid] = wev;
        s_cap[wid] = wcap;
        s_wcnt[0][wid] = wnmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double bm = s_max[0];
        unsigned long long be = s_ev[0], bs = (unsigned long long)s_wcnt[0][0];
        int bc = s_cap[0];
#pragma unroll
        for (int w = 1; w < kBlock / 64; ++w) {
            bm = fmax(bm, s_max[w]);
            be += s_ev[w];
            bs += (unsigned long long)s_wcnt[0][w];
            bc |= s_cap[w];
        }
        if (!kFlat && a.want_weight && bm > -__builtin_inf()) publish_max(&a.segmax[b], ordered_key(bm));
        if (!kFlat && bc) atomicOr(&a.status[b], 1u /* DPOMP_ST_EVENT_CAP */);
        if (be) {  // events of this interval, and what a one-particle-per-lane schedule makes of it (dpomp_get_lane_utilisation)
            sched_add(a.sched, be, 64ull * bs);
        }
#ifdef DPOMP_TRACE_WG
        if (a.trace) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* r = a.trace + 8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
            r[0] = tr_t0; r[1] = tr_t1; r[2] = tr_x3; r[3] = tr_x2; r[4] = tr_x1; r[5] = wall_clock64();
            r[6] = ((unsigned long long)xcc << 32) | hw; r[7] = be;
        }
#endif
    }
}

// ---- k_filter_small: a whole filter per workgroup (P <= 256) ------------------------------------------------
// The reference's default is 200 state particles per filter (C_DF_PF_P) under thousands of theta-particles
// (SMC^2) or chains.  Such a filter fits one workgroup: one particle per lane, and everything the multi-kernel
// path sends through HBM between launches stays on chip -- workgroup max and fixed-point sum (wave reductions +
// LDS), inclusive scan of the weights, offspring boundaries N_j in LDS, ancestor search over them (8 LDS steps),
// parents' states read from an LDS tile -- for ALL observations ymin..ymax in ONE launch (no kernel boundaries,
// no global round trips, no wasted 1024-slot tiles).  Same numerical spec, same RNG keys: bit-identical to the
// multi-kernel path and to the oracle.  Systematic resampling, non-adaptive mode.
struct SmallArgs {
    DevModel m;
    int32_t* state;              // [B][S][Pp] current buffer (read when ymin > 1, written at the end)
    double* logw;                // [B][Pp]  log-weights of the last weighted observation (inspection)
    int32_t* offs;               // [B][Pp]  N_j of the last resampling (inspection)
    const double* theta;         // [B][K]
    const double* obs_time;      // [T]
    const long long* ysum;       // [T]
    const int32_t* obs_id;       // [T]
    const long long* yval;       // [T][V] Observation.val (user observation models)
    int32_t n_yval;
    double* ll;                  // [B]
    unsigned long long* events;  // [1]
    uint32_t* status;            // [B]
    int64_t Pp;
    int32_t P, T, ymin, ymax, sh;
    uint32_t seg_base;
    uint64_t seed;
};

template <class R>
__global__ __launch_bounds__(kBlock) void k_filter_small(SmallArgs a) {
    constexpr int S = R::S, E = R::E;
    using TW = typename cond_t<(S <= 4), uint32_t, uint64_t>::type;
    __shared__ double s_log[kLogTableSize][2];
    __shared__ int32_t s_x[S][kBlock];
    __shared__ int32_t s_N[kBlock];
    __shared__ double s_wmax[kBlock / 64];
    __shared__ unsigned long long s_wsum[kBlock / 64], s_ev[kBlock / 64];
    __shared__ int s_cap[kBlock / 64];
    __shared__ double s_r;

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const bool valid = tid < a.P;
    const int s_real = R::kGeneric ? a.m.S : S;
    const int e_real = R::kGeneric ? a.m.E : E;
    const TransRows<R> rows(a.m, e_real);
    const LogRegs lr = log_regs();  // (a polynomial constant of the logarithm, in a VGPR pair for the whole kernel)
    int32_t* st = a.state + (int64_t)b * s_real * a.Pp;
    const double* th = a.theta + (int64_t)b * a.m.K;
    const ThetaRegs<R> th_regs(th, a.m.K);
    const double* thr = th_regs.get(th);  // for the rate evaluator inside the event loop
    for (int i = tid; i < 2 * kLogTableSize; i += kBlock) (&s_log[0][0])[i] = (&kLogTable[0][0])[i];

    int x[S];
#pragma unroll
    for (int s = 0; s < S; ++s)
        x[s] = (!valid || s >= s_real) ? 0 : (a.ymin == 1 ? model_x0<R>(a.m, s) : st[(int64_t)s * a.Pp + tid]);
    double t_prev = (a.ymin == 1) ? (a.m.t0_index > 0 ? th[a.m.t0_index - 1] : 0.0) : a.obs_time[a.ymin - 2];
    const int cap = a.m.cap;
    double ll = 0.0;
    unsigned long long ev = 0;
    bool any_cap = false, zero_w = false;
    __syncthreads();

    for (int obs_i = a.ymin; obs_i <= a.ymax; ++obs_i) {
        const int k = obs_i - 1;
        const double tmax = a.obs_time[k];
        const bool weighted = a.obs_id[k] > 0;
        // ---- iterate_particles!  (hmm_particle_filter.jl:9-33)
        double time = t_prev;
        Stream rng = stream_init(a.seed, a.seg_base + (uint32_t)b, (uint32_t)tid, (uint32_t)k, DOMAIN_PROPAGATE);
        int n = 0;
        bool capped = false;
        if (valid) {
            Draws ahead = draw_ahead(rng, s_log, lr);
            DPOMP_LOOP_AT(DPOMP_PAD_SMALL);
            for (;;) {  // (two events per trip: see DPOMP_LOOP_AT)
                if (!gillespie_event<R, true, true>(a.m, rows, thr, x, time, tmax, rng, ahead, e_real, s_log, lr, &n, cap)) break;
                if (!gillespie_event<R, true, true>(a.m, rows, thr, x, time, tmax, rng, ahead, e_real, s_log, lr, &n, cap)) break;
            }
            capped = n >= cap;
            if (!capped && stopped_on_range(time, tmax))
                capped = finish_slow<R>(a.m, rows, thr, x, time, tmax, rng, ahead, cap, e_real, s_log, lr);
            n = events_executed(rng, time, tmax);
        }
        ev += (unsigned long long)n;
        any_cap |= capped;
        t_prev = tmax;  // :72
        if (!weighted) continue;  // obs_id <= 0: propagate only (:58)

        // ---- weights: log mean weight (:60) with the max factored out, fixed point
        double g = -__builtin_inf();
        if (valid) {
            g = obs_log_weight<R>(a.m, th, x, a.ysum[k], a.yval + (int64_t)k * a.n_yval, a.n_yval);
            if (capped || !(g > -__builtin_inf())) g = -__builtin_inf();
            a.logw[(int64_t)b * a.Pp + tid] = g;
        }
        const double wm = wave_max(g);
        if (lane == 0) s_wmax[wid] = wm;
#pragma unroll
        for (int s = 0; s < S; ++s) s_x[s][tid] = x[s];  // parents' states for the gather below
        __syncthreads();
        const double m = fmax(fmax(s_wmax[0], s_wmax[1]), fmax(s_wmax[2], s_wmax[3]));
        const unsigned long long w = valid ? fixed_weight(g, m, a.sh) : 0ull;
        const unsigned long long incl = wave_inclusive_scan(w, lane);
        if (lane == 63) s_wsum[wid] = incl;
        if (tid == 0) {  // rand() of hmm_pf_resample.jl:27
            Stream rs = stream_init(a.seed, a.seg_base + (uint32_t)b, 0u, (uint32_t)k, DOMAIN_RESAMPLE);
            s_r = u01(stream_next(rs));
        }
        __syncthreads();
        unsigned long long c = incl;
        for (int w2 = 0; w2 < wid; ++w2) c += s_wsum[w2];
        const unsigned long long W = (s_wsum[0] + s_wsum[1]) + (s_wsum[2] + s_wsum[3]);
        if (W == 0) {
            ll += -__builtin_inf();
            zero_w = true;
        } else {
            ll += log_mean_weight(m, W, a.sh, a.P);
        }
        if (obs_i < a.T) {  // :62 -- rsp_systematic (hmm_pf_resample.jl:24-42) through the offspring boundaries
            const int nj = (W == 0) ? a.P : (int)offspring_end(c, W, a.P, s_r);
            s_N[tid] = valid ? nj : a.P;
            if (valid) a.offs[(int64_t)b * a.Pp + tid] = nj;
            __syncthreads();
            if (valid) {
                int lo = 0, hi = a.P - 1;  // first j with N_j > tid
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (s_N[mid] > tid) hi = mid; else lo = mid + 1;
                }
#pragma unroll
                for (int s = 0; s < S; ++s) x[s] = s_x[s][lo];
            }
        }
        __syncthreads();  // s_x / s_N / s_wmax are rewritten at the next observation
    }

#pragma unroll
    for (int s = 0; s < S; ++s)
        if (valid && s < s_real) st[(int64_t)s * a.Pp + tid] = x[s];
    const unsigned long long wev = wave_sum(ev);
    const int wcap = wave_any(any_cap);
    if (lane == 0) {
        s_ev[wid] = wev;
        s_cap[wid] = wcap;
    }
    __syncthreads();
    if (tid == 0) {
        a.ll[b] += ll;
        const unsigned long long be = (s_ev[0] + s_ev[1]) + (s_ev[2] + s_ev[3]);
        if (be) atomicAdd(a.events, be);
        const uint32_t stb = ((s_cap[0] | s_cap[1] | s_cap[2] | s_cap[3]) ? 1u : 0u) | (zero_w ? 2u : 0u);
        if (stb) atomicOr(&a.status[b], stb);
    }
}

// ---- k_simulate_rec: forward simulation with event recording -------------------------------------------
// gillespie_sim (hmm_sim.jl:86-102): one trajectory per lane through all T observation intervals in one
// launch (no resampling between them), every executed event appended as (time, 1-based type) to the
// lane's column of event-major buffers (coalesced across lanes), state recorded at each observation time.
struct SimArgs {
    DevModel m;
    const double* theta;     // [K]
    const double* obs_time;  // [T]
    long long* counts;       // [P] events recorded, -1 = max_events exceeded (truncated)
    double* ev_time;         // [max_events][P]
    int32_t* ev_type;        // [max_events][P]
    long long* states;       // [T][S][P]
    int64_t max_events;
    int32_t P, T;
    uint32_t seg;
    uint64_t seed;
};

template <class R>
__global__ __launch_bounds__(kBlock) void k_simulate_rec(SimArgs a) {
    constexpr int S = R::S, E = R::E;
    using TW = typename cond_t<(S <= 4), uint32_t, uint64_t>::type;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= a.P) return;
    const int s_real = R::kGeneric ? a.m.S : S;
    const int e_real = R::kGeneric ? a.m.E : E;
    const TransRows<R> rows(a.m, e_real);
    int x[S];
#pragma unroll
    for (int s = 0; s < S; ++s) x[s] = (s < s_real) ? model_x0<R>(a.m, s) : 0;
    double time = a.m.t0_index > 0 ? a.theta[a.m.t0_index - 1] : 0.0;
    long long n = 0;
    bool overflow = false;
    for (int k = 0; k < a.T; ++k) {
        Stream rng = stream_init(a.seed, a.seg, (uint32_t)p, (uint32_t)k, DOMAIN_SIMULATE);
        const double tmax = a.obs_time[k];
        while (!overflow) {
            double cum[E];
            R::rates(a.m, a.theta, x, cum);
#pragma unroll
            for (int e = 1; e < E; ++e) cum[e] = cum[e - 1] + cum[e];
            const double tot = cum[E - 1];
            if (tot == 0.0) break;
            time = time - dlog(u01(stream_next(rng))) / tot;
            if (time > tmax) break;
            const double etc = u01(stream_next(rng)) * tot;
            int ev = e_real - 1;
            if constexpr (R::kBig) {
#pragma unroll
                for (int i = E - 2; i >= 0; --i)
                    if (cum[i] > etc) ev = i;
                R::apply_event(ev, x);
            } else {
                TW tw = (TW)a.m.trans_packed[e_real - 1];
#pragma unroll
                for (int i = E - 2; i >= 0; --i) {
                    const bool in_range = R::kGeneric ? (i < e_real - 1) : true;
                    if (in_range && cum[i] > etc) {
                        tw = (TW)a.m.trans_packed[i];
                        ev = i;
                    }
                }
#pragma unroll
                for (int s = 0; s < S; ++s) x[s] += (int)(int8_t)(tw >> (8 * s));
            }
            if (n >= a.max_events) {
                overflow = true;
                break;
            }
            a.ev_time[n * a.P + p] = time;
            a.ev_type[n * a.P + p] = ev + 1;
            ++n;
        }
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (s < s_real) a.states[((int64_t)k * s_real + s) * a.P + p] = x[s];
        time = tmax;
    }
    a.counts[p] = overflow ? -1 : n;
}

// ---- weights -----------------------------------------------------------------------------------
struct WeightArgs {
    const double* logw;                 // [B][Pp]
    unsigned long long* wfix;           // [B][Pp] fixed-point weights: written by k_wsum, read by k_offsets
    const unsigned long long* segmax;   // [B]
    unsigned long long* blocksum;       // [B][nblk]
    unsigned long long* blockprefix;    // [B][nblk] exclusive prefix of blocksum
    unsigned long long* wtot;           // [B]
    double* ll;                         // [B] running log-likelihood
    uint32_t* status;                   // [B]
    int32_t* offs;                      // [B][Pp]
    int32_t* tile_end;                  // [B][Pp/256] first_parent: parent block (of 256) of the first slot of each 256-slot output block
    double* posw;                       // [B][Pp]   pos_j = c_j / W * P (stratified / multinomial resamplers)
    double* tile_end_pos;               // [B][nblk] pos of the last particle of each tile
    double* rsu;                        // [B] the systematic-resampling uniform of this observation (from k_wsum)
    // ESS-adaptive mode (ess_frac > 0; otherwise all unused)
    double ess_frac;
    unsigned long long* blocksum2;      // [B][nblk][2] per-tile sum of w^2 (hi, lo)
    const int32_t* carry_in;            // [B] weights of this observation include carried ones
    const double* lprev_in;             // [B] log sum exp of the carried log-weights
    int32_t* carry_out;                 // [B]
    double* lprev_out;                  // [B]
    int32_t* noresample;                // [B] 1 = keep the population at this observation (gathers copy identically)
    int32_t last_obs;                   // no resampling after the last observation of the data set
    int64_t Pp;
    int32_t rs_type;                    // DPOMP_RS_*
    int32_t P, nblk, sh;
    int32_t tiles_per_wg;               // k_offsets: consecutive tiles of one filter a workgroup walks (>= 1)
    uint32_t obs_k, seg_base;
    uint64_t seed;
};

__device__ __forceinline__ void load_tile_weights(const WeightArgs& a, int b, int blk,
                                                  unsigned long long (&w)[kItems]) {
    const double m = ordered_value(a.segmax[b]);
    const int64_t i0 = (int64_t)blk * kTile + threadIdx.x * kItems;
    const double* src = a.logw + (int64_t)b * a.Pp + i0;
    const double2 g01 = *reinterpret_cast<const double2*>(src);
    const double2 g23 = *reinterpret_cast<const double2*>(src + 2);
    const double g[kItems] = {g01.x, g01.y, g23.x, g23.y};
#pragma unroll
    for (int q = 0; q < kItems; ++q) w[q] = (i0 + q < a.P) ? fixed_weight(g[q], m, a.sh) : 0ull;
    // kept for k_offsets: the exponential is evaluated once per particle and observation
    ulonglong2* dst = reinterpret_cast<ulonglong2*>(a.wfix + (int64_t)b * a.Pp + i0);
    dst[0] = make_ulonglong2(w[0], w[1]);
    dst[1] = make_ulonglong2(w[2], w[3]);
}
__device__ __forceinline__ void load_tile_fixed(const WeightArgs& a, int b, int blk, unsigned long long (&w)[kItems]) {
    const int64_t i0 = (int64_t)blk * kTile + threadIdx.x * kItems;
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(a.wfix + (int64_t)b * a.Pp + i0);
    const ulonglong2 w01 = src[0], w23 = src[1];
    w[0] = w01.x;
    w[1] = w01.y;
    w[2] = w23.x;
    w[3] = w23.y;
}

__device__ __forceinline__ void add128(unsigned long long& hi, unsigned long long& lo, unsigned long long h,
                                       unsigned long long l) {
    const unsigned long long nl = lo + l;
    hi += h + (nl < l ? 1ull : 0ull);
    lo = nl;
}
__device__ __forceinline__ void wave_sum128(unsigned long long& hi, unsigned long long& lo) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long h = __shfl_xor(hi, off, 64), l = __shfl_xor(lo, off, 64);
        add128(hi, lo, h, l);
    }
}

// Closes one weighted observation of one segment: log-likelihood increment (hmm_particle_filter.jl:60, or
// LSE(t) - LSE(a) when weights were carried) and, in the ESS-adaptive mode, the carry state for the next one.
__device__ __forceinline__ void finish_segment(const WeightArgs& a, int b, unsigned long long W, bool resample) {
    a.wtot[b] = W;
    const bool carried = a.ess_frac > 0.0 && a.carry_in[b] != 0;
    double lk = 0.0;
    if (W == 0) {
        a.ll[b] += -__builtin_inf();
        atomicOr(&a.status[b], 2u /* DPOMP_ST_ZERO_WEIGHT */);
    } else {
        const double m = ordered_value(a.segmax[b]);
        lk = log_sum_weight(m, W, a.sh);
        a.ll[b] += carried ? lk - a.lprev_in[b] : log_mean_weight(m, W, a.sh, a.P);
    }
    if (a.ess_frac > 0.0) {
        const bool keep = !a.last_obs && !resample;
        a.carry_out[b] = keep ? 1 : 0;
        a.lprev_out[b] = lk;
        a.noresample[b] = keep ? 1 : 0;
    }
}

// k_wsum: per-tile sum of the fixed-point weights rint(exp(g - max) 2^sh).
__global__ __launch_bounds__(kBlock) void k_wsum(WeightArgs a) {
    const int b = blockIdx.y, blk = blockIdx.x;
    unsigned long long w[kItems];
    load_tile_weights(a, b, blk, w);
    unsigned long long t = (w[0] + w[1]) + (w[2] + w[3]);
    __shared__ unsigned long long s_sum[kBlock / 64], s_hi[kBlock / 64], s_lo[kBlock / 64];
    t = wave_sum(t);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = t;
    if (a.ess_frac > 0.0) {  // sum of squares, exact in 128 bits (w < 2^53)
        unsigned long long hi = 0, lo = 0;
#pragma unroll
        for (int q = 0; q < kItems; ++q) add128(hi, lo, __umul64hi(w[q], w[q]), w[q] * w[q]);
        wave_sum128(hi, lo);
        if ((threadIdx.x & 63) == 0) {
            s_hi[threadIdx.x >> 6] = hi;
            s_lo[threadIdx.x >> 6] = lo;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.blocksum[(int64_t)b * a.nblk + blk] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
        if (a.ess_frac > 0.0) {
            unsigned long long hi = s_hi[0], lo = s_lo[0];
            for (int w2 = 1; w2 < kBlock / 64; ++w2) add128(hi, lo, s_hi[w2], s_lo[w2]);
            a.blocksum2[((int64_t)b * a.nblk + blk) * 2] = hi;
            a.blocksum2[((int64_t)b * a.nblk + blk) * 2 + 1] = lo;
        }
        if (blk == 0) {  // rand() of hmm_pf_resample.jl:27, once per segment instead of once per thread of k_offsets
            Stream rs = stream_init(a.seed, a.seg_base + (uint32_t)b, 0u, a.obs_k, DOMAIN_RESAMPLE);
            a.rsu[b] = u01(stream_next(rs));
        }
    }
}

// k_scan_blocks: one workgroup per segment; exclusive scan of the tile sums, the total W, and
// output += log(cum_weight[end] / P)   (hmm_particle_filter.jl:60).
__global__ __launch_bounds__(kBlock) void k_scan_blocks(WeightArgs a) {
    const int b = blockIdx.x;
    const int per = (a.nblk + kBlock - 1) / kBlock;
    const int lo = threadIdx.x * per;
    const int hi = min(lo + per, a.nblk);
    const unsigned long long* bs = a.blocksum + (int64_t)b * a.nblk;
    unsigned long long* bp = a.blockprefix + (int64_t)b * a.nblk;
    unsigned long long t = 0;
    for (int i = lo; i < hi; ++i) t += bs[i];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned long long incl = wave_inclusive_scan(t, lane);
    __shared__ unsigned long long s_w[kBlock / 64];
    if (lane == 63) s_w[wid] = incl;
    __syncthreads();
    unsigned long long base = 0;
    for (int w = 0; w < wid; ++w) base += s_w[w];
    unsigned long long run = base + (incl - t);
    for (int i = lo; i < hi; ++i) {
        bp[i] = run;
        run += bs[i];
    }
    // ESS-adaptive mode: total of the squared weights (every thread its tiles, then wave and workgroup reduction; 128-bit
    // integer sums are exact in any order) and the decision
    __shared__ unsigned long long s_h2[kBlock / 64], s_l2[kBlock / 64];
    if (a.ess_frac > 0.0) {
        unsigned long long h2 = 0, l2 = 0;
        const unsigned long long* b2 = a.blocksum2 + (int64_t)b * a.nblk * 2;
        for (int i = lo; i < hi; ++i) add128(h2, l2, b2[2 * i], b2[2 * i + 1]);
        wave_sum128(h2, l2);
        if (lane == 0) {
            s_h2[wid] = h2;
            s_l2[wid] = l2;
        }
        __syncthreads();
    }
    if (threadIdx.x == kBlock - 1) {
        const unsigned long long W = run;
        bool resample = true;
        if (a.ess_frac > 0.0) {
            unsigned long long h2 = s_h2[0], l2 = s_l2[0];
            for (int w = 1; w < kBlock / 64; ++w) add128(h2, l2, s_h2[w], s_l2[w]);
            resample = ess_resample(W, h2, l2, a.ess_frac, a.P);
        }
        finish_segment(a, b, W, resample);
    }
}

// offspring_end without branches (selects): the same operations on the same values, so the same result for every
// input -- four calls per thread are straight-line code instead of twelve divergent regions.
__device__ __forceinline__ int offspring_end_lane(unsigned long long c, unsigned long long W, double Wd, int P, double r) {
    const double pos = div_in_range(u64_to_double(c), Wd) * (double)P - r;
    double n = __builtin_ceil(pos);
    n = n > 0.0 ? n : 0.0;
    n = n > (double)P ? (double)P : n;
    const int v = (int)n;
    return c >= W ? P : v;
}

// Systematic resampling of one tile: offspring boundaries N_j of the tile's particles from their inclusive cumulative
// weights base + c[q], and the first_parent map entries they own.  Called by every thread of the workgroup (barrier).
// prev_last: boundary of the particle before the tile when the caller has it (the last N of the tile it did before
// this one), -1 to compute it from tile_prefix.  Returns (to thread 0) the last boundary of this tile.
template <int ITEMS>  // particles per thread: kItems in the tiled kernels; 1 or 2 for filters of at most 256 / 512 particles
__device__ __forceinline__ int systematic_boundaries(const WeightArgs& a, int b, int blk, unsigned long long tile_prefix,
                                                     unsigned long long base, const unsigned long long (&c)[ITEMS],
                                                     unsigned long long W, double r, int prev_last) {
    static_assert(ITEMS == 1 || ITEMS == 2 || ITEMS == 4, "the particles of a thread share a block of 256");
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t i0 = (int64_t)blk * (kBlock * ITEMS) + threadIdx.x * ITEMS;
    const double Wd = u64_to_double(W);
    int n[ITEMS];
#pragma unroll
    for (int q = 0; q < ITEMS; ++q)
        n[q] = (W == 0) ? a.P : offspring_end_lane(base + c[q], W, Wd, a.P, r);
    int32_t* const offs = a.offs + (int64_t)b * a.Pp + i0;
    if constexpr (ITEMS == 4) *reinterpret_cast<int4*>(offs) = make_int4(n[0], n[1], n[2], n[3]);
    else if constexpr (ITEMS == 2) *reinterpret_cast<int2*>(offs) = make_int2(n[0], n[1]);
    else offs[0] = n[0];
    // Inverse map for the gather: every 256-slot OUTPUT block t learns which parent block (of 256) its first slot
    // (t * 256) copies from.  Particle j owns the slots [N_{j-1}, N_j), so the particle whose range contains
    // t * 256 writes first_parent[t] = its block; exactly one writer per output block, no cross-workgroup
    // dependency.  N_{j-1} of a thread's first particle is the last boundary of the thread before it: taken from
    // the neighbouring lane, by the first lane of a wave from the wave before (through LDS), and computed from the
    // tile's exclusive prefix by the first thread of the tile only -- not by every thread (a division each).
    __shared__ int s_last[kBlock / 64];
    int lo = __builtin_amdgcn_update_dpp(0, n[ITEMS - 1], 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
    if (lane == 63) s_last[wid] = n[ITEMS - 1];
    __syncthreads();
    int tile_last = 0;
    if (lane == 0) {
        if (wid > 0) lo = s_last[wid - 1];
        else {
            if (prev_last >= 0) lo = prev_last;
            else lo = (W == 0) ? ((blk == 0) ? 0 : a.P) : (int)offspring_end(tile_prefix, W, a.P, r);
            tile_last = s_last[kBlock / 64 - 1];
        }
    }
    int32_t* fp = a.tile_end + (int64_t)b * (a.Pp / kParentBlock);
    const int my_block = (int)(i0 / kParentBlock);  // (the particles of a thread share a 256-block)
    // the thread's particles own the slots [lo, n[3]): entries t with t * 256 in that range
    const int t0 = (lo + kParentBlock - 1) / kParentBlock, t1 = (n[ITEMS - 1] + kParentBlock - 1) / kParentBlock;
    const bool wide = t1 - t0 > 2;
    if (!wide)
        for (int t = t0; t < t1; ++t) fp[t] = my_block;
    // Degenerate weights give one particle most of the offspring -- thousands of entries from one thread would
    // take longer than the rest of the kernel: wide ranges are filled by the whole wave, 64 entries at a time.
    for (unsigned long long todo = wave_ballot(wide); todo; todo &= todo - 1) {
        const int src = __builtin_ctzll(todo);
        const int w0 = __builtin_amdgcn_readlane(t0, src), w1 = __builtin_amdgcn_readlane(t1, src),
                  wb = __builtin_amdgcn_readlane(my_block, src);
        for (int t = w0 + lane; t < w1; t += 64) fp[t] = wb;
    }
    return tile_last;
}

// k_offsets<FUSED>: inclusive fixed-point cumulative weight c_j of every particle (tile scan + tile
// prefix) -> offspring boundary N_j.  Slots [N_{j-1}, N_j) of the resampled population copy
// particle j: rsp_systematic's u_i = (u1 + (i-1)/P) cw[end], `while u[i] > cw[j]`
// (hmm_pf_resample.jl:27-37) solved for i instead of walked.
// A workgroup walks a.tiles_per_wg consecutive tiles of one filter (the next tile's weights are loaded while it works
// on this one): what it needs once -- the prefix up to its first tile and the total W -- is then paid once for
// several tiles, and the tile sums every workgroup of a filter reads are read by fewer of them.
// FUSED: every workgroup sums the (<= 2048) tile sums itself -- its own exclusive prefix and the
// total W -- so no separate scan launch is needed; workgroup 0 also adds the log mean weight to
// the running log-likelihood (hmm_particle_filter.jl:60).  Otherwise the prefix and W come from
// k_scan_blocks.  Also writes tile_end[blk] = N of the tile's last particle for k_gather.
template <bool FUSED>
__global__ __launch_bounds__(kBlock) void k_offsets(WeightArgs a) {
    const int b = blockIdx.y, blk0 = blockIdx.x * a.tiles_per_wg;
    const int blk1 = min(blk0 + a.tiles_per_wg, a.nblk);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __shared__ unsigned long long s_w[2][kBlock / 64], s_pre[kBlock / 64], s_tot[kBlock / 64], s_hi[kBlock / 64],
        s_lo[kBlock / 64];
    unsigned long long tile_prefix, W;
    const bool adaptive = a.ess_frac > 0.0;
    if (FUSED) {
        const unsigned long long* bs = a.blocksum + (int64_t)b * a.nblk;
        unsigned long long pre = 0, tot = 0, hi = 0, lo = 0;
        for (int i = threadIdx.x; i < a.nblk; i += kBlock) {
            const unsigned long long v = bs[i];
            tot += v;
            if (i < blk0) pre += v;
            if (adaptive)  // (also needs every tile's sum of squares)
                add128(hi, lo, a.blocksum2[((int64_t)b * a.nblk + i) * 2], a.blocksum2[((int64_t)b * a.nblk + i) * 2 + 1]);
        }
        pre = wave_sum(pre);
        tot = wave_sum(tot);
        if (adaptive) wave_sum128(hi, lo);
        if (lane == 0) {
            s_pre[wid] = pre;
            s_tot[wid] = tot;
            s_hi[wid] = hi;
            s_lo[wid] = lo;
        }
    } else if (adaptive && a.noresample[b]) {
        return;  // k_scan_blocks decided to keep this segment's population
    }
    unsigned long long w[kItems];
    load_tile_fixed(a, b, blk0, w);
    if (FUSED) {
        __syncthreads();
        tile_prefix = (s_pre[0] + s_pre[1]) + (s_pre[2] + s_pre[3]);
        W = (s_tot[0] + s_tot[1]) + (s_tot[2] + s_tot[3]);
        bool resample = true;
        if (adaptive) {  // every workgroup reaches the same decision from the same exact sums
            unsigned long long hi = s_hi[0], lo = s_lo[0];
            for (int w2 = 1; w2 < kBlock / 64; ++w2) add128(hi, lo, s_hi[w2], s_lo[w2]);
            resample = ess_resample(W, hi, lo, a.ess_frac, a.P);
        }
        if (blk0 == 0 && threadIdx.x == 0) finish_segment(a, b, W, resample);
        if (!resample) return;  // population kept: the gather copies it identically
    } else {
        tile_prefix = a.blockprefix[(int64_t)b * a.nblk + blk0];
        W = a.wtot[b];
    }
    const double r = a.rsu[b];  // rand() of hmm_pf_resample.jl:27 (drawn in k_wsum)
    int prev_last = -1;
    for (int blk = blk0; blk < blk1; ++blk) {  // (uniform)
        unsigned long long c[kItems];
        c[0] = w[0];
#pragma unroll
        for (int q = 1; q < kItems; ++q) c[q] = c[q - 1] + w[q];
        if (blk + 1 < blk1) load_tile_fixed(a, b, blk + 1, w);
        const unsigned long long incl = wave_inclusive_scan(c[kItems - 1], lane);
        unsigned long long* sw = s_w[(blk - blk0) & 1];  // (two buffers: the next tile writes while slow waves still read)
        if (lane == 63) sw[wid] = incl;
        __syncthreads();
        unsigned long long base = tile_prefix + (incl - c[kItems - 1]);
        for (int ww = 0; ww < wid; ++ww) base += sw[ww];
        const int64_t i0 = (int64_t)blk * kTile + threadIdx.x * kItems;
        if (a.rs_type == 1) {
            prev_last = systematic_boundaries<kItems>(a, b, blk, tile_prefix, base, c, W, r, prev_last);
        } else {
            // stratified / multinomial: scaled cumulative weights pos_j = c_j / W * P; slot i later takes the
            // first j with pos_j > t_i.  W == 0: pos = P for every j, so every slot copies particle 0.
            const double Wd = u64_to_double(W);
            double ps[kItems];
#pragma unroll
            for (int q = 0; q < kItems; ++q)
                ps[q] = (W == 0) ? (double)a.P : div_in_range(u64_to_double(base + c[q]), Wd) * (double)a.P;
            double* dst = a.posw + (int64_t)b * a.Pp + i0;
            *reinterpret_cast<double2*>(dst) = make_double2(ps[0], ps[1]);
            *reinterpret_cast<double2*>(dst + 2) = make_double2(ps[2], ps[3]);
            if (threadIdx.x == kBlock - 1) a.tile_end_pos[(int64_t)b * a.nblk + blk] = ps[kItems - 1];
        }
        tile_prefix += (sw[0] + sw[1]) + (sw[2] + sw[3]);
    }
}

// ---- k_weights_tile: k_wsum + k_offsets<true> in one launch for filters of at most one tile (P <= 1024) -------
// With a single tile per filter nothing crosses a workgroup: the tile's sum IS the filter's total, so the fixed-point
// weights, their sum, the log mean weight (hmm_particle_filter.jl:60) and the offspring boundaries of rsp_systematic
// (hmm_pf_resample.jl:24-42) come from one workgroup per filter -- one launch per observation instead of two, and the
// weights never go through memory.  Same arithmetic, operation for operation, as the two-kernel path (integer sums are
// order-free).  Systematic resampling, non-adaptive mode.
template <int ITEMS>  // particles per thread: 1 / 2 / 4 for filters of at most 256 / 512 / 1024 particles (all 256 threads busy)
__global__ __launch_bounds__(kBlock) void k_weights_tile(WeightArgs a, int resample) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __shared__ unsigned long long s_w[kBlock / 64];
    __shared__ double s_r;
    const double m = ordered_value(a.segmax[b]);
    const int64_t i0 = threadIdx.x * ITEMS;
    unsigned long long c[ITEMS];
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) c[q] = 0ull;
    if (i0 < a.P) {  // (threads past the filter hold zero weights: no load)
        const double* src = a.logw + (int64_t)b * a.Pp + i0;
        double g[ITEMS];
        if constexpr (ITEMS == 4) {
            const double2 g01 = *reinterpret_cast<const double2*>(src);
            const double2 g23 = *reinterpret_cast<const double2*>(src + 2);
            g[0] = g01.x, g[1] = g01.y, g[2] = g23.x, g[3] = g23.y;
        } else if constexpr (ITEMS == 2) {
            const double2 g01 = *reinterpret_cast<const double2*>(src);
            g[0] = g01.x, g[1] = g01.y;
        } else {
            g[0] = src[0];
        }
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) c[q] = (i0 + q < a.P) ? fixed_weight(g[q], m, a.sh) : 0ull;
    }
#pragma unroll
    for (int q = 1; q < ITEMS; ++q) c[q] = c[q - 1] + c[q];
    const unsigned long long incl = wave_inclusive_scan(c[ITEMS - 1], lane);
    if (lane == 63) s_w[wid] = incl;
    if (threadIdx.x == 0) {  // rand() of hmm_pf_resample.jl:27
        Stream rs = stream_init(a.seed, a.seg_base + (uint32_t)b, 0u, a.obs_k, DOMAIN_RESAMPLE);
        const double r = u01(stream_next(rs));
        s_r = r;
        a.rsu[b] = r;
    }
    __syncthreads();
    const unsigned long long W = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
    if (threadIdx.x == 0) finish_segment(a, b, W, true);
    if (!resample) return;  // (uniform) last observation of the data set: no resampling (:62)
    unsigned long long base = incl - c[ITEMS - 1];
    for (int ww = 0; ww < wid; ++ww) base += s_w[ww];
    systematic_boundaries<ITEMS>(a, b, 0, 0ull, base, c, W, s_r, -1);
}

// ---- k_gather: m_pop[i,:] .= old_p[j_i,:]   (hmm_pf_resample.jl:38) -----------------------------
// Ancestor of slot i = first j with N_j > i.  N is monotone, so the 1024 output slots of a workgroup draw from
// a contiguous window of parent tiles [first_parent[t], first_parent[t+1]] (written by k_offsets).  The
// workgroup stages that window of N in LDS (up to kStageTiles tiles, coalesced 16-B loads) and every lane
// searches there (12 LDS steps, its 4 searches interleaved); wider windows (very uneven weights) fall back to
// a bounded binary search in L2.  Writes are 16 B per lane per component; parent reads are monotone in i.
struct GatherArgs {
    const int32_t* src;  // [B][S][Pp]
    int32_t* dst;
    const int32_t* offs;          // [B][Pp]  N_j
    const int32_t* first_parent;  // [B][nblk]
    const int32_t* noresample;    // [B] or null: 1 = identity copy (ESS-adaptive mode kept the population)
    int64_t Pp;
    int32_t P, S, nblk;
};

constexpr int kStageTiles = 4;

__global__ __launch_bounds__(kBlock) void k_gather(GatherArgs a) {
    __shared__ __attribute__((aligned(16))) int32_t s_n[kStageTiles * kTile];
    const int b = blockIdx.y, t_out = blockIdx.x;
    const int64_t i0 = (int64_t)t_out * kTile + threadIdx.x * kItems;
    const int32_t* N = a.offs + (int64_t)b * a.Pp;
    const int32_t* fp = a.first_parent + (int64_t)b * (a.Pp / kParentBlock);  // per 256-slot block, in parent blocks of 256
    if (a.noresample && a.noresample[b]) {
        for (int s = 0; s < a.S; ++s) {
            const int64_t o = ((int64_t)b * a.S + s) * a.Pp + i0;
            *reinterpret_cast<int4*>(a.dst + o) = *reinterpret_cast<const int4*>(a.src + o);
        }
        return;
    }
    constexpr int kPerTile = kTile / kParentBlock;
    const int tlo = fp[t_out * kPerTile] / kPerTile;  // parent TILE of this output tile's first slot
    const int thi = (t_out + 1 < a.nblk && (int64_t)(t_out + 1) * kTile < a.P) ? fp[(t_out + 1) * kPerTile] / kPerTile
                                                                             : (a.P - 1) / kTile;  // ancestors are monotone in the slot index
    const int ntiles = thi - tlo + 1;
    int j[kItems];
    if (ntiles <= kStageTiles) {
        const int4* gsrc = reinterpret_cast<const int4*>(N + (int64_t)tlo * kTile);
        int4* ldst = reinterpret_cast<int4*>(s_n);
        for (int t = threadIdx.x; t < ntiles * (kTile / 4); t += kBlock) ldst[t] = gsrc[t];
        __syncthreads();
        const int M = min(ntiles * kTile, a.P - tlo * kTile);  // (boundaries past the last particle need not exist)
        int lo[kItems];
#pragma unroll
        for (int q = 0; q < kItems; ++q) lo[q] = 0;
#pragma unroll
        for (int step = kStageTiles * kTile / 2; step > 0; step >>= 1) {
#pragma unroll
            for (int q = 0; q < kItems; ++q) {  // lo = number of staged boundaries <= slot
                const int mid = lo[q] + step;
                if (mid <= M && s_n[mid - 1] <= (int)(i0 + q)) lo[q] = mid;
            }
        }
#pragma unroll
        for (int q = 0; q < kItems; ++q) j[q] = (i0 + q < a.P) ? min(tlo * kTile + lo[q], a.P - 1) : 0;
    } else {
#pragma unroll
        for (int q = 0; q < kItems; ++q)
            j[q] = (i0 + q < a.P) ? tlo * kTile + first_greater(N + (int64_t)tlo * kTile, min(ntiles * kTile, a.P - tlo * kTile), (int)(i0 + q)) : 0;
    }
    for (int s = 0; s < a.S; ++s) {
        const int32_t* src = a.src + ((int64_t)b * a.S + s) * a.Pp;
        const int4 v = make_int4(src[j[0]], src[j[1]], src[j[2]], src[j[3]]);
        *reinterpret_cast<int4*>(a.dst + ((int64_t)b * a.S + s) * a.Pp + i0) = v;
    }
}

// ---- k_gather_pos: stratified (rsp_stratified hmm_pf_resample.jl:46-63) and multinomial (:5-20) --------
// Slot i draws one uniform from its own stream (seed, segment, i, observation, DOMAIN_RESAMPLE_PARTICLE);
// threshold t_i = i + u_i (stratified: one point per stratum) or P u_i (multinomial: i.i.d. points);
// ancestor = first j with pos_j > t_i: tile through the tile_end_pos array in LDS, then 10 steps in L2.
constexpr int kGatherPosStageTiles = 4096;  // tile_end_pos (8 B per tile) staged in LDS up to this many tiles
struct GatherPosArgs {
    const int32_t* src;
    int32_t* dst;
    const double* posw;
    const double* tile_end_pos;
    const int32_t* noresample;  // as in GatherArgs
    int64_t Pp;
    int32_t P, S, nblk, rs_type;
    uint32_t obs_k, seg_base;
    uint64_t seed;
};

template <class Ptr>
__device__ __forceinline__ int first_greater_d(Ptr v, int n, double t) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (v[mid] > t) hi = mid; else lo = mid + 1;
    }
    return lo;
}

__global__ __launch_bounds__(kBlock) void k_gather_pos(GatherPosArgs a) {
    extern __shared__ __attribute__((aligned(16))) double s_tep[];  // [nblk]
    const int b = blockIdx.y;
    const int64_t i0 = (int64_t)blockIdx.x * kTile + threadIdx.x * kItems;
    const double* pos = a.posw + (int64_t)b * a.Pp;
    const double* TE = a.tile_end_pos + (int64_t)b * a.nblk;
    if (a.noresample && a.noresample[b]) {
        for (int s = 0; s < a.S; ++s) {
            const int64_t o = ((int64_t)b * a.S + s) * a.Pp + i0;
            *reinterpret_cast<int4*>(a.dst + o) = *reinterpret_cast<const int4*>(a.src + o);
        }
        return;
    }
    const bool staged = a.nblk <= kGatherPosStageTiles;  // (uniform) larger filters search the tile ends in L2
    if (staged) {
        for (int t = threadIdx.x; t < a.nblk; t += kBlock) s_tep[t] = TE[t];
        __syncthreads();
    }
    int j[kItems];
#pragma unroll
    for (int q = 0; q < kItems; ++q) {
        const int64_t i = i0 + q;
        if (i < a.P) {
            Stream ps = stream_init(a.seed, a.seg_base + (uint32_t)b, (uint32_t)i, a.obs_k, DOMAIN_RESAMPLE_PARTICLE);
            const double u = u01(stream_next(ps));
            const double thr = (a.rs_type == 2) ? (double)i + u : (double)a.P * u;
            const int t = staged ? first_greater_d(s_tep, a.nblk, thr) : first_greater_d(TE, a.nblk, thr);
            const int within = first_greater_d(pos + (int64_t)t * kTile, kTile, thr);
            j[q] = min(t * kTile + within, a.P - 1);
        } else {
            j[q] = 0;
        }
    }
    for (int s = 0; s < a.S; ++s) {
        const int32_t* src = a.src + ((int64_t)b * a.S + s) * a.Pp;
        const int4 v = make_int4(src[j[0]], src[j[1]], src[j[2]], src[j[3]]);
        *reinterpret_cast<int4*>(a.dst + ((int64_t)b * a.S + s) * a.Pp + i0) = v;
    }
}

// ---- k_copy_segments: pop[b] <- pop[ancestors[b]]   (hmm_ibis.jl:71-75) ---------------------------
__global__ __launch_bounds__(kBlock) void k_copy_segments(const int32_t* src, int32_t* dst,
                                                          const int64_t* ancestors, int64_t seg_ints) {
    const int64_t b = blockIdx.y;
    const int64_t i0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kItems;
    if (i0 >= seg_ints) return;
    const int64_t a = ancestors[b];
    *reinterpret_cast<int4*>(dst + b * seg_ints + i0) =
        *reinterpret_cast<const int4*>(src + a * seg_ints + i0);
}

// ---- k_take_segments: dst[b] <- src[take[b]] where take[b] >= 0   (hmm_ibis.jl:109) -----------------
__global__ __launch_bounds__(kBlock) void k_take_segments(const int32_t* src, int32_t* dst, const int64_t* take,
                                                          int64_t seg_ints) {
    const int64_t b = blockIdx.y;
    const int64_t i0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kItems;
    const int64_t t = take[b];
    if (t < 0 || i0 >= seg_ints) return;
    *reinterpret_cast<int4*>(dst + b * seg_ints + i0) = *reinterpret_cast<const int4*>(src + t * seg_ints + i0);
}

// ---- k_selftest: device evaluation of the shared numerics ----------------------------------------
__global__ void k_selftest(int64_t n, uint64_t seed, const double* x, double* out_u, double* out_log,
                           double* out_exp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (out_u) {
        Stream s = stream_init(seed, 0u, (uint32_t)i, 0u, DOMAIN_PROPAGATE);
        out_u[i] = u01(stream_next(s));
    }
    if (x && out_log) out_log[i] = dlog(x[i]);
    if (x && out_exp) out_exp[i] = dexp(-37.0 * x[i]);
}

// Device quotients a[i] / b[i] by the event loop's in-range division (b in its range) and by the plain `/`.
__global__ void k_selftest_div(int64_t n, const double* a, const double* b, double* out_fast, double* out_plain) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_fast[i] = rate_in_fast_range(b[i]) ? div_in_range(a[i], b[i]) : __builtin_nan("");
    out_plain[i] = a[i] / b[i];
}

}  // namespace dpomp
