/*
 * dpomp_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A serial CPU restatement of the DiscretePOMP.jl bootstrap particle filter
 * (reference paths relative to the reference tree, `src/...`), used only as
 * the checker for the HIP path (tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py).  Nothing under discretepomp.jl_amd/ may
 * import, link or call this file.
 *
 * Parity status: the reference is Julia-only and Julia is not installed in
 * the build container, so the reference itself cannot be run here and its
 * tests hold no golden vectors for this path (they pin seed-1 values of whole
 * analyses that depend on Julia's RNG stream).  At BIT level this oracle is
 * therefore "parity unpinned" against the reference; at DISTRIBUTION level it
 * is pinned by the exact CTMC forward-algorithm likelihood of the Pooley SIS
 * data set (oracle/exact_sis.py, tests/test_oracle_exact.py).
 *
 * Two modes:
 *   ORC_MODE_FAITHFUL  follows the reference operation for operation in
 *       Float64 with libm log/exp: linear weights `total_weight += exp(g)`
 *       as a serial left fold (hmm_particle_filter.jl:29-30), `log(cw[end]/P)`
 *       (:60), and rsp_systematic verbatim (hmm_pf_resample.jl:24-42).
 *   ORC_MODE_SPEC      the numerical spec the HIP kernels implement
 *       (discretepomp.jl_amd/csrc/dpomp_numerics.h): same propagation, but elementary
 *       functions dlog/dexp restated below (so CPU and GPU agree to the bit),
 *       max-shifted weights held as 64-bit fixed point (so the sum and scan
 *       are associative and every partition gives the same bits), offspring
 *       boundaries N_j = ceil(c_j/W*P - r).  The device result must equal
 *       this mode exactly; FAITHFUL vs SPEC differences are bounded in tests.
 *
 * The only thing the two modes share with the product is the POD model
 * descriptor from include/dpomp.h (the public ABI header).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/dpomp.h"
#include "orc_log_table.h"

#define ORC_MODE_SPEC 0
#define ORC_MODE_FAITHFUL 1

#define ORC_DOMAIN_PROPAGATE 0u
#define ORC_DOMAIN_RESAMPLE 1u
#define ORC_DOMAIN_SIMULATE 2u
#define ORC_DOMAIN_RESAMPLE_PARTICLE 3u

/* ------------------------------------------------------------------------- */
/* Counter-based RNG.  The reference draws from Julia's global RNG           */
/* (`rand()` at hmm_particle_filter.jl:23, hmm_cmn.jl:5, hmm_pf_resample.jl:27); */
/* a serial global stream cannot be reproduced by 2^20 concurrent lanes, so  */
/* the spec keys a private stream per (seed, segment, particle, observation).*/
/* ------------------------------------------------------------------------- */

/* Philox4x32-10 (Salmon et al., SC'11), restated from the published algorithm. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* sfc64 (C. Doty-Humphrey's "small fast chaotic" generator as published with PractRand; shifts 11/3, rotate 24),
 * restated from the published algorithm.  State: a, b, c and a counter. */
uint64_t orc_stream_next(uint64_t s[4]) {
    uint64_t t = s[0] + s[1] + s[3];
    s[3] += 1;
    s[0] = s[1] ^ (s[1] >> 11);
    s[1] = s[2] + (s[2] << 3);
    s[2] = ((s[2] << 24) | (s[2] >> 40)) + t;
    return t;
}

#define ORC_STREAM_WARMUP 12 /* outputs discarded after seeding, as PractRand's sfc64 seeding does */

/* Stream for (seed, segment b, particle p, observation k, domain): the 128 Philox bits give a and b, c is a fixed
 * odd constant, the counter starts at 1, and the first 12 outputs are discarded. */
void orc_stream_init(uint64_t seed, uint32_t b, uint32_t p, uint32_t k, uint32_t domain,
                     uint64_t s[4]) {
    uint32_t ctr[4] = {p, b, k, domain};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t o[4];
    orc_philox4x32_10(ctr, key, o);
    s[0] = (uint64_t)o[0] | ((uint64_t)o[1] << 32);
    s[1] = (uint64_t)o[2] | ((uint64_t)o[3] << 32);
    s[2] = 0x9E3779B97F4A7C15ull;
    s[3] = 1;
    for (int i = 0; i < ORC_STREAM_WARMUP; ++i) (void)orc_stream_next(s);
}

/* 52 random bits -> (0,1): u = (2k+1) * 2^-53, never 0 or 1, exact in double. */
double orc_u01(uint64_t x) { return (double)(2 * (x >> 12) + 1) * 0x1p-53; }

void orc_uniforms(uint64_t seed, uint32_t b, uint32_t p, uint32_t k, uint32_t domain, int64_t n,
                  double* out) {
    uint64_t s[4];
    orc_stream_init(seed, b, p, k, domain, s);
    for (int64_t i = 0; i < n; ++i) out[i] = orc_u01(orc_stream_next(s));
}

/* ------------------------------------------------------------------------- */
/* Elementary functions of the SPEC mode.  libm and the GPU's ocml differ by */
/* up to an ulp, which would break bit parity; both sides therefore evaluate */
/* the same fixed sequence of IEEE operations (fma = one rounding).          */
/* ------------------------------------------------------------------------- */

/* log(x) for positive normal x, table-driven (structure of glibc's log; table generated by
 * tools/gen_log_table.py, re-derived entry by entry in tests/test_oracle_numerics.py):
 * x = 2^k z, z in [0.6875, 1.375); T[i] = (invc, logc) for the i-th of 512 subintervals (|r| <= 2^-9.5);
 * r = fma(z, invc, -1); log(x) = (k ln2hi + logc) + (k ln2lo + r + r^2 (A0 + ... + A4 r^4)).  Max error 1.5 ulp. */
double orc_dlog(double x) {
    static const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
    static const double A0 = -0.5, A1 = 0x1.5555555555555p-2, A2 = -0.25, A3 = 0x1.999999999999ap-3,
                        A4 = -0x1.5555555555555p-3;
    uint64_t ix;
    memcpy(&ix, &x, 8);
    uint64_t tmp = ix - 0x3FE6000000000000ull;
    int i = (int)((tmp >> 43) & 511);
    int32_t k = (int32_t)((int64_t)tmp >> 52);
    uint64_t iz = ix - (tmp & 0xFFF0000000000000ull);
    double z;
    memcpy(&z, &iz, 8);
    double invc = kLogTable[i][0], logc = kLogTable[i][1];
    double r = fma(z, invc, -1.0);
    double kd = (double)k;
    double w = fma(kd, Ln2hi, logc);
    double s = fma(kd, Ln2lo, r);
    double r2 = r * r;
    double p = fma(r, fma(r, fma(r, fma(r, A4, A3), A2), A1), A0);
    return w + fma(r2, p, s);
}

/* exp(x) for -37 <= x <= 0 (callers cut off below -37, where the fixed-point weight
 * is zero anyway).  x = k ln2 + r, exp(r) = 1 + r + r c/(2-c); error < 1 ulp. */
double orc_dexp(double x) {
    static const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                        invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                        P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                        P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    int32_t k = (int32_t)(invln2 * x - 0.5);
    double t = (double)k;
    double hi = x - t * ln2HI;
    double lo = t * ln2LO;
    double r = hi - lo;
    double tt = r * r;
    double c = r - tt * (P1 + tt * (P2 + tt * (P3 + tt * (P4 + tt * P5))));
    double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    uint64_t bits;
    memcpy(&bits, &y, 8);
    bits += (uint64_t)(int64_t)k << 52; /* y * 2^k, result stays normal for k >= -54 */
    memcpy(&y, &bits, 8);
    return y;
}

/* u64 -> double as hi*2^32 + lo: both halves exact, one rounding. */
static inline double u64_to_double(uint64_t v) {
    return (double)(uint32_t)(v >> 32) * 4294967296.0 + (double)(uint32_t)v;
}

/* ------------------------------------------------------------------------- */
/* Model pieces                                                               */
/* ------------------------------------------------------------------------- */

static inline int64_t mask_sum(uint32_t mask, const int64_t* x, int S) {
    int64_t s = 0;
    for (int i = 0; i < S; ++i)
        if (mask & (1u << i)) s += x[i];
    return s;
}

/* Event rates (not yet cumulated) from the generic rate table.  Restates the
 * catalogue rate functions hmm_examples.jl:103-168: Julia promotes each integer
 * operand to Float64 as the left-associated product proceeds (DPOMP_RATE_LEFT);
 * rossmac_rf :165-166 multiplies two integers first and divides Int/Int -> Float64
 * (DPOMP_RATE_RATIO). */
typedef void (*orc_user_rates_fn)(const double* theta, const int* x, double* r);
static orc_user_rates_fn g_user_rates = 0;
/* Host twin of a run-time compiled rate source (DPOMP_BUILTIN_USER): tests compile the SAME source text for
 * the CPU (g++ -ffp-contract=off) and register it here, so user rate code is checked like the built-ins. */
void orc_set_user_rates(orc_user_rates_fn fn) { g_user_rates = fn; }

void orc_rates(const dpomp_model_desc* d, const double* theta, const int64_t* x, double* rates) {
    if (d->builtin_id == DPOMP_BUILTIN_USER && g_user_rates) {
        int xi[DPOMP_MAX_STATES];
        for (int i = 0; i < d->n_states; ++i) xi[i] = (int)x[i];
        g_user_rates(theta, xi, rates);
        return;
    }
    for (int e = 0; e < d->n_events; ++e) {
        const dpomp_rate_term* t = &d->rate[e];
        double th = theta[t->theta_index];
        if (t->form == DPOMP_RATE_LEFT) {
            double r = th;
            for (int f = 0; f < t->n_factors; ++f)
                r = r * (double)mask_sum(t->factor_mask[f], x, d->n_states);
            if (t->denom_mask) r = r / (double)mask_sum(t->denom_mask, x, d->n_states);
            rates[e] = r;
        } else {
            int64_t prod = 1;
            for (int f = 0; f < t->n_factors; ++f)
                prod *= mask_sum(t->factor_mask[f], x, d->n_states);
            double q = (double)prod;
            if (t->denom_mask) q = q / (double)mask_sum(t->denom_mask, x, d->n_states);
            rates[e] = th * q;
        }
    }
}

/* choose_event, hmm_cmn.jl:4-10: etc = rand()*cum[end]; first i < E with cum[i] > etc,
 * else E.  Returns a 0-based event index. */
int orc_choose_event(const double* cum, int E, double u) {
    double etc = u * cum[E - 1];
    for (int i = 0; i < E - 1; ++i)
        if (cum[i] > etc) return i;
    return E - 1;
}

/* Sum of the observed value columns, `sum(y.val[y_seq])` hmm_examples.jl:64. */
static int64_t obs_value_sum(const dpomp_model_desc* d, int k) {
    int64_t s = 0;
    for (int v = 0; v < d->n_obs_values; ++v)
        if (d->obs.value_mask & (1u << v)) s += d->obs_value[(int64_t)k * d->n_obs_values + v];
    return s;
}

/* partial_gaussian_obs_model, hmm_examples.jl:59-67:
 *   tmp1 = log(1/(sqrt(2 pi) sigma)); tmp2 = 2 sigma^2;
 *   tmp1 - ((sum(y.val[y_seq]) - sum(pop[seq]))^2) / tmp2      (integer square, then Float64) */
typedef double (*orc_user_obs_fn)(const double* theta, const int* x, const long long* y, int ny);
static orc_user_obs_fn g_user_obs = 0;
static const double* g_obs_theta = 0; /* set per call by the filter (the obs model receives theta) */
void orc_set_user_obs(orc_user_obs_fn fn) { g_user_obs = fn; }

double orc_obs_loglik_theta(const dpomp_model_desc* d, int k, const int64_t* x, const double* theta);

double orc_obs_loglik(const dpomp_model_desc* d, int k, const int64_t* x) {
    return orc_obs_loglik_theta(d, k, x, g_obs_theta);
}

double orc_obs_loglik_theta(const dpomp_model_desc* d, int k, const int64_t* x, const double* theta) {
    if (d->obs.kind == DPOMP_OBS_USER && g_user_obs) { /* host twin of dpomp_user_obs (obs_model(y, pop, theta)) */
        int xi[DPOMP_MAX_STATES];
        long long yv[32];
        for (int i = 0; i < d->n_states; ++i) xi[i] = (int)x[i];
        for (int v = 0; v < d->n_obs_values; ++v) yv[v] = d->obs_value[(int64_t)k * d->n_obs_values + v];
        return g_user_obs(theta, xi, yv, d->n_obs_values);
    }
    double sigma = d->obs.sigma;
    double tmp1 = log(1.0 / (sqrt(2.0 * M_PI) * sigma));
    double tmp2 = 2.0 * sigma * sigma;
    int64_t diff = obs_value_sum(d, k) - mask_sum(d->obs.state_mask, x, d->n_states);
    return tmp1 - ((double)(diff * diff)) / tmp2;
}

static int64_t default_cap(const dpomp_model_desc* d) {
    return d->max_events_per_interval > 0 ? d->max_events_per_interval : ((int64_t)1 << 22);
}

/* The Gillespie inner loop shared by iterate_particles! (hmm_particle_filter.jl:19-27)
 * and iterate_particle! (hmm_sim.jl:8-16, :57-64):
 *     rate_function -> cumsum! -> total == 0 ? break -> time -= log(rand())/total ->
 *     time > tmax ? break -> pop += transition[choose_event]
 * x is advanced in place from time t to tmax; returns the number of executed events,
 * or -1 if the event cap was hit (a guard the reference lacks). */
static int64_t gillespie_advance(const dpomp_model_desc* d, const double* theta, int64_t* x,
                                 double t, double tmax, uint64_t s[4], int mode, int64_t cap) {
    const int E = d->n_events, S = d->n_states;
    double cum[DPOMP_MAX_EVENTS];
    int64_t n = 0;
    double time = t;
    for (;;) {
        orc_rates(d, theta, x, cum);
        for (int e = 1; e < E; ++e) cum[e] = cum[e - 1] + cum[e]; /* cumsum! :21 */
        if (cum[E - 1] == 0.0) break;                             /* :22 */
        double u = orc_u01(orc_stream_next(s));
        double lg = (mode == ORC_MODE_FAITHFUL) ? log(u) : orc_dlog(u);
        time -= lg / cum[E - 1];                                  /* :23 */
        if (time > tmax) break;                                   /* :24 */
        int ev = orc_choose_event(cum, E, orc_u01(orc_stream_next(s))); /* :26 */
        for (int i = 0; i < S; ++i) x[i] += d->transition[ev][i];
        if (++n >= cap) return -1;
    }
    return n;
}

/* iterate_particles!, hmm_particle_filter.jl:9-33, for observation k (0-based):
 * advance every particle from t to tmax and evaluate its log-weight.  pop is P x S
 * column-major like the reference's Array{Int64,2}.  The reference's serial running sum
 * `total_weight += exp(...)` (:29-30) is done by the caller from logw. */
void orc_iterate_particles(const dpomp_model_desc* d, int64_t* pop, int64_t P, const double* theta,
                           int k, double t, double tmax, uint64_t seed, uint32_t seg, int mode,
                           int nthreads, double* logw, uint64_t* events, uint32_t* status,
                           int32_t* per_particle_events /* [P] or NULL */) {
    const int S = d->n_states;
    const int64_t cap = default_cap(d);
    uint64_t ev_total = 0;
    uint32_t st = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads > 0 ? nthreads : 1) \
    reduction(+ : ev_total) reduction(| : st)
#endif
    for (int64_t p = 0; p < P; ++p) {
        int64_t x[DPOMP_MAX_STATES];
        for (int i = 0; i < S; ++i) x[i] = pop[p + P * i]; /* ptemp .= pop[p,:] :17 */
        uint64_t s[4];
        orc_stream_init(seed, seg, (uint32_t)p, (uint32_t)k, ORC_DOMAIN_PROPAGATE, s);
        int64_t n = gillespie_advance(d, theta, x, t, tmax, s, mode, cap);
        double g = orc_obs_loglik_theta(d, k, x, theta); /* :29 */
        if (n < 0) {
            st |= DPOMP_ST_EVENT_CAP;
            g = -INFINITY;
            n = cap;
        }
        logw[p] = g;
        if (per_particle_events) per_particle_events[p] = (int32_t)n;
        ev_total += (uint64_t)n;
        for (int i = 0; i < S; ++i) pop[p + P * i] = x[i]; /* :31 */
    }
    (void)nthreads;
    *events += ev_total;
    *status |= st;
}

static int ceil_log2_i64(int64_t v) {
    int n = 0;
    while (((int64_t)1 << n) < v) ++n;
    return n;
}

/* SPEC weights: m = max g, w_p = rint(dexp(g_p - m) * 2^sh) as u64, sh = min(52, 62 - ceil(log2 P)),
 * so that W = sum w_p < 2^63 and integer addition makes sum/scan order-independent. */
int orc_weight_shift(int64_t P) {
    int sh = 62 - ceil_log2_i64(P);
    return sh > 52 ? 52 : sh;
}

uint64_t orc_fixed_weight(double g, double m, int sh) {
    if (!(g > -INFINITY)) return 0; /* -inf and NaN */
    double x = g - m;
    if (x < -37.0) return 0;
    return (uint64_t)rint(orc_dexp(x) * (double)((uint64_t)1 << sh));
}

/* SPEC offspring boundary: number of output slots i with ancestor <= j, given the inclusive
 * fixed-point cumulative weight c of particle j:  N = clamp(ceil(c/W * P - r), 0, P).
 * Monotone in c, N(0) = 0, N(W) = P (pinned).  Slot i takes the first j with N_j > i, i.e. the first j
 * with c_j > (i + r)/P * W  -- rsp_systematic's `while u[i] > cw[j]` (hmm_pf_resample.jl:36). */
int64_t orc_offspring_end(uint64_t c, uint64_t W, int64_t P, double r) {
    if (c >= W) return P; /* N(W) = P exactly: P - r can round down to P - 1 when r is within 2^-53 of 1 */
    double Wd = u64_to_double(W);
    double pos = (u64_to_double(c) / Wd) * (double)P - r;
    double n = ceil(pos);
    if (!(n > 0.0)) return 0;
    if (n > (double)P) return P;
    return (int64_t)n;
}

/* ------------------------------------------------------------------------- */
/* partial_log_likelihood!, hmm_particle_filter.jl:39-76                      */
/* ------------------------------------------------------------------------- */
typedef struct orc_trace {
    double* per_obs_ll;     /* [T] log mean weight per observation (0 where obs_id <= 0) */
    double* logw;           /* [P] log-weights of the last weighted observation          */
    int64_t* offspring_end; /* [P] SPEC N_j of the last resampling                       */
    int64_t* ancestors;     /* [P] ancestor of every slot at the last resampling         */
    int64_t* states;        /* [T][nkeep][S] state of the first nkeep particles after
                               propagation to each observation (before resampling)       */
    int64_t nkeep;
    int32_t* particle_events; /* [T][P] executed events per particle and interval (work-balance analysis) */
} orc_trace;

/* Carried state of the ESS-adaptive mode between calls (the reference has no such state: its criterion is
 * commented out, hmm_particle_filter.jl:63-68, and it resamples at every weighted observation). */
typedef struct orc_carry {
    int active;      /* weights were carried from the previous weighted observation */
    double l_prev;   /* log sum exp of the carried log-weights */
    double* a;       /* [P] carried log-weights */
} orc_carry;

static double u128_to_double(unsigned __int128 v) {
    return u64_to_double((uint64_t)(v >> 64)) * 18446744073709551616.0 + u64_to_double((uint64_t)v);
}

/* SPEC resampling decision of the adaptive mode: ESS = W^2 / sum w^2 from the exact integer sums. */
int orc_ess_resample(uint64_t W, uint64_t s2_hi, uint64_t s2_lo, double ess_frac, int64_t P) {
    if (W == 0) return 1;
    double Wd = u64_to_double(W);
    double ess = (Wd * Wd) / u128_to_double(((unsigned __int128)s2_hi << 64) | s2_lo);
    return ess < ess_frac * (double)P;
}

int orc_pf_partial_adaptive(const dpomp_model_desc* d, int64_t* pop, int64_t P, const double* theta,
                            int ymin, int ymax, uint64_t seed, uint32_t seg, int mode, int rs_type,
                            int nthreads, double ess_frac, orc_carry* carry, double* out_ll,
                            uint32_t* out_status, uint64_t* out_events, const orc_trace* tr);

int orc_pf_partial(const dpomp_model_desc* d, int64_t* pop, int64_t P, const double* theta,
                   int ymin, int ymax, uint64_t seed, uint32_t seg, int mode, int rs_type,
                   int nthreads, double* out_ll, uint32_t* out_status, uint64_t* out_events,
                   const orc_trace* tr) {
    return orc_pf_partial_adaptive(d, pop, P, theta, ymin, ymax, seed, seg, mode, rs_type, nthreads, 0.0, 0,
                                   out_ll, out_status, out_events, tr);
}

/* ess_frac <= 0: the reference's behaviour (resample at every weighted observation but the last).
 * ess_frac > 0 (SPEC mode only): resample only when ESS < ess_frac * P, otherwise carry the log-weights:
 * t_p = a_p + g_p, log-likelihood increment = LSE(t) - LSE(a). */
int orc_pf_partial_adaptive(const dpomp_model_desc* d, int64_t* pop, int64_t P, const double* theta,
                            int ymin, int ymax, uint64_t seed, uint32_t seg, int mode, int rs_type,
                            int nthreads, double ess_frac, orc_carry* carry, double* out_ll,
                            uint32_t* out_status, uint64_t* out_events, const orc_trace* tr) {
    const int S = d->n_states, T = d->n_obs;
    if (ymin < 1 || ymax > T || ymin > ymax || P < 1) return DPOMP_EINVAL;
    if (rs_type < DPOMP_RS_SYSTEMATIC || rs_type > DPOMP_RS_MULTINOMIAL) return DPOMP_EINVAL;
    if (ess_frac > 0.0 && (mode != ORC_MODE_SPEC || !carry || !carry->a)) return DPOMP_EINVAL;
    if (ess_frac > 0.0 && ymin == 1) carry->active = 0;
    double t_prev;
    if (ymin == 1) { /* :41-45 */
        for (int64_t p = 0; p < P; ++p)
            for (int i = 0; i < S; ++i) pop[p + P * i] = d->initial_condition[i];
        t_prev = d->t0_index == 0 ? 0.0 : theta[d->t0_index - 1];
    } else {
        t_prev = d->obs_time[ymin - 2]; /* :47 */
    }
    int64_t* old_p = (int64_t*)malloc(sizeof(int64_t) * P * S); /* :50 */
    double* logw = (double*)malloc(sizeof(double) * P);
    double* cw = (double*)malloc(sizeof(double) * P);       /* FAITHFUL cum_weight :53 */
    uint64_t* cint = (uint64_t*)malloc(sizeof(uint64_t) * P); /* SPEC cumulative fixed point */
    if (!old_p || !logw || !cw || !cint) return DPOMP_ENOMEM;
    double output = 0.0; /* :52 */
    uint32_t status = 0;
    uint64_t events = 0;
    const int sh = orc_weight_shift(P);

    for (int obs_i = ymin; obs_i <= ymax; ++obs_i) { /* :54 */
        const int k = obs_i - 1;
        orc_iterate_particles(d, pop, P, theta, k, t_prev, d->obs_time[k], seed, seg, mode,
                              nthreads, logw, &events, &status,
                              (tr && tr->particle_events) ? tr->particle_events + (int64_t)k * P : NULL); /* :56 */
        if (tr && tr->states)
            for (int64_t p = 0; p < tr->nkeep && p < P; ++p)
                for (int i = 0; i < S; ++i)
                    tr->states[((int64_t)k * tr->nkeep + p) * S + i] = pop[p + P * i];
        if (d->obs_id[k] > 0) { /* :58 */
            if (tr && tr->logw) memcpy(tr->logw, logw, sizeof(double) * P);
            double inc, lk = 0.0;
            uint64_t W = 0;
            unsigned __int128 s2 = 0;
            if (mode == ORC_MODE_FAITHFUL) {
                double total_weight = 0.0; /* :13 */
                for (int64_t p = 0; p < P; ++p) {
                    total_weight += exp(logw[p]); /* :29 */
                    cw[p] = total_weight;         /* :30 */
                }
                inc = log(cw[P - 1] / (double)P); /* :60 */
            } else {
                const int carried = ess_frac > 0.0 && carry->active;
                if (carried)
                    for (int64_t p = 0; p < P; ++p) logw[p] = logw[p] + carry->a[p]; /* t_p = g_p + a_p */
                double m = -INFINITY;
                for (int64_t p = 0; p < P; ++p)
                    if (logw[p] > m) m = logw[p];
                uint64_t c = 0;
                for (int64_t p = 0; p < P; ++p) {
                    uint64_t wv = orc_fixed_weight(logw[p], m, sh);
                    c += wv;
                    s2 += (unsigned __int128)wv * wv;
                    cint[p] = c;
                }
                if (c == 0) {
                    inc = -INFINITY;
                } else {
                    double Wd = u64_to_double(c);
                    W = c;
                    lk = m + orc_dlog(Wd * (1.0 / (double)((uint64_t)1 << sh)));
                    inc = carried ? lk - carry->l_prev
                                  : m + orc_dlog((Wd * (1.0 / (double)((uint64_t)1 << sh))) / (double)P);
                }
            }
            if (!(inc > -INFINITY)) status |= DPOMP_ST_ZERO_WEIGHT;
            output += inc; /* :60 */
            if (tr && tr->per_obs_ll) tr->per_obs_ll[k] = inc;
            if (obs_i < T && ess_frac > 0.0 &&
                !orc_ess_resample(W, (uint64_t)(s2 >> 64), (uint64_t)s2, ess_frac, P)) {
                /* ESS above the threshold: keep the population, carry the log-weights */
                memcpy(carry->a, logw, sizeof(double) * P);
                carry->active = 1;
                carry->l_prev = lk;
            } else if (obs_i < T) { /* :62 */
                if (ess_frac > 0.0) carry->active = 0;
                memcpy(old_p, pop, sizeof(int64_t) * P * S); /* old_p .= pop :66 */
                uint64_t rs[4];
                orc_stream_init(seed, seg, 0u, (uint32_t)k, ORC_DOMAIN_RESAMPLE, rs);
                double r = orc_u01(orc_stream_next(rs)); /* rand() hmm_pf_resample.jl:27 */
                if (rs_type != DPOMP_RS_SYSTEMATIC) {
                    /* rsp_stratified (Kitagawa 1996, hmm_pf_resample.jl:46-63) and rsp_multinomial
                     * (inverse CDF, :5-20).  Both throw as shipped (undefined `output` :55 / `weights` :10);
                     * this follows their documented intent: stratified u_i = (rand_i + (i-1))/N * cw[end],
                     * multinomial chs_i = rand_i * cw[end]; ancestor = first j whose cumulative weight
                     * exceeds it.  One uniform per slot from its own stream. */
                    for (int64_t i = 0; i < P; ++i) {
                        uint64_t ps[4];
                        orc_stream_init(seed, seg, (uint32_t)i, (uint32_t)k, ORC_DOMAIN_RESAMPLE_PARTICLE, ps);
                        double ui = orc_u01(orc_stream_next(ps));
                        int64_t j = 0;
                        if (mode == ORC_MODE_FAITHFUL) {
                            double thr = (rs_type == DPOMP_RS_STRATIFIED) ? (ui / (double)P + (double)i / (double)P) * cw[P - 1]
                                                                          : ui * cw[P - 1];
                            if (rs_type == DPOMP_RS_STRATIFIED) {
                                while (thr > cw[j] && j < P - 1) ++j; /* :57 */
                            } else {
                                j = P - 1; /* new_p = length(weights) :11 */
                                for (int64_t p2 = 0; p2 < P - 1; ++p2)
                                    if (thr < cw[p2]) { j = p2; break; } /* :14 */
                            }
                        } else if (W > 0) {
                            /* SPEC: pos_j = c_j/W * P; first j with pos_j > t_i, t_i = i + u_i or P u_i */
                            double Wd = u64_to_double(W);
                            double thr = (rs_type == DPOMP_RS_STRATIFIED) ? (double)i + ui : (double)P * ui;
                            int64_t lo = 0, hi = P - 1;
                            while (lo < hi) {
                                int64_t mid = (lo + hi) >> 1;
                                double pos = (u64_to_double(cint[mid]) / Wd) * (double)P;
                                if (pos > thr) hi = mid; else lo = mid + 1;
                            }
                            j = lo;
                        }
                        for (int s2 = 0; s2 < S; ++s2) pop[i + P * s2] = old_p[j + P * s2];
                        if (tr && tr->ancestors) tr->ancestors[i] = j;
                    }
                } else if (mode == ORC_MODE_FAITHFUL) {
                    /* rsp_systematic hmm_pf_resample.jl:24-42, verbatim in Float64 */
                    double u1 = r / (double)P; /* :27 */
                    int64_t j = 0;
                    for (int64_t i = 0; i < P; ++i) {
                        double ui = (i == 0) ? u1 : u1 + ((double)i / (double)P); /* :29 */
                        ui *= cw[P - 1];                                          /* :31 */
                        while (ui > cw[j] && j < P - 1) ++j; /* :36 (clamped; the reference
                                                                 would throw on overshoot) */
                        for (int s = 0; s < S; ++s) pop[i + P * s] = old_p[j + P * s]; /* :38 */
                        if (tr && tr->ancestors) tr->ancestors[i] = j;
                    }
                } else {
                    /* SPEC: offspring boundaries from the fixed-point cumulative weights.
                     * W == 0 (all weights zero): every slot copies particle 0, which is what
                     * the reference degenerates to (u = 0, `0 > cw[1]` false). */
                    int64_t start = 0;
                    for (int64_t j = 0; j < P; ++j) {
                        int64_t end = (W > 0) ? orc_offspring_end(cint[j], W, P, r) : P;
                        if (tr && tr->offspring_end) tr->offspring_end[j] = end;
                        for (int64_t i = start; i < end; ++i) {
                            for (int s = 0; s < S; ++s) pop[i + P * s] = old_p[j + P * s];
                            if (tr && tr->ancestors) tr->ancestors[i] = j;
                        }
                        if (end > start) start = end;
                    }
                }
            }
        } else if (tr && tr->per_obs_ll) {
            tr->per_obs_ll[k] = 0.0;
        }
        t_prev = d->obs_time[k]; /* :72 */
    }
    free(old_p);
    free(logw);
    free(cw);
    free(cint);
    *out_ll = output; /* :75 */
    if (out_status) *out_status = status;
    if (out_events) *out_events = events;
    return DPOMP_OK;
}

/* estimate_likelihood, hmm_particle_filter.jl:79-84. */
int orc_pf_loglik(const dpomp_model_desc* d, int64_t P, const double* theta, uint64_t seed,
                  uint32_t seg, int mode, int rs_type, int nthreads, double* out_ll,
                  uint32_t* out_status, uint64_t* out_events, const orc_trace* tr) {
    int64_t* pop = (int64_t*)calloc((size_t)P * d->n_states, sizeof(int64_t)); /* zeros(Int64,P,S) :82 */
    if (!pop) return DPOMP_ENOMEM;
    int rc = orc_pf_partial(d, pop, P, theta, 1, d->n_obs, seed, seg, mode, rs_type, nthreads,
                            out_ll, out_status, out_events, tr);
    free(pop);
    return rc;
}

/* Forward simulation of independent trajectories, states recorded at each observation
 * time only (gillespie_sim hmm_sim.jl:86-102 with iterate_particle! :55-70, minus the
 * Event/pop_v recording).  out[(t*S + s)*P + p]. */
int orc_simulate(const dpomp_model_desc* d, int64_t P, const double* theta, uint64_t seed,
                 uint32_t seg, int64_t* out) {
    const int S = d->n_states, T = d->n_obs;
    const int64_t cap = default_cap(d);
    for (int64_t p = 0; p < P; ++p) {
        int64_t x[DPOMP_MAX_STATES];
        for (int i = 0; i < S; ++i) x[i] = d->initial_condition[i];
        double t = d->t0_index == 0 ? 0.0 : theta[d->t0_index - 1]; /* hmm_sim.jl:92 */
        for (int k = 0; k < T; ++k) {
            uint64_t s[4];
            orc_stream_init(seed, seg, (uint32_t)p, (uint32_t)k, ORC_DOMAIN_SIMULATE, s);
            gillespie_advance(d, theta, x, t, d->obs_time[k], s, ORC_MODE_SPEC, cap);
            for (int i = 0; i < S; ++i) out[((int64_t)k * S + i) * P + p] = x[i];
            t = d->obs_time[k];
        }
    }
    return DPOMP_OK;
}

/* One forward simulation with its event history: gillespie_sim (hmm_sim.jl:86-102) with
 * iterate_particle! (:55-70) -- `push!(p.trajectory, Event(time, et))` :65 for every executed event.
 * Event types are reported 1-based as in the reference.  Returns the number of events, or -1 when
 * `max_events` would be exceeded (the trajectory is then truncated). */
int64_t orc_simulate_events(const dpomp_model_desc* d, const double* theta, uint64_t seed, uint32_t seg,
                            uint32_t p, int64_t max_events, double* out_times, int32_t* out_types,
                            int64_t* out_states /* [T][S] */) {
    const int S = d->n_states, T = d->n_obs, E = d->n_events;
    int64_t x[DPOMP_MAX_STATES];
    for (int i = 0; i < S; ++i) x[i] = d->initial_condition[i];
    double time = d->t0_index == 0 ? 0.0 : theta[d->t0_index - 1];
    int64_t n = 0;
    int overflow = 0;
    for (int k = 0; k < T; ++k) {
        uint64_t s[4];
        orc_stream_init(seed, seg, p, (uint32_t)k, ORC_DOMAIN_SIMULATE, s);
        double cum[DPOMP_MAX_EVENTS];
        while (!overflow) {
            orc_rates(d, theta, x, cum);
            for (int e = 1; e < E; ++e) cum[e] = cum[e - 1] + cum[e];
            if (cum[E - 1] == 0.0) break;
            time -= orc_dlog(orc_u01(orc_stream_next(s))) / cum[E - 1];
            if (time > d->obs_time[k]) break;
            int ev = orc_choose_event(cum, E, orc_u01(orc_stream_next(s)));
            for (int i = 0; i < S; ++i) x[i] += d->transition[ev][i];
            if (n >= max_events) { overflow = 1; break; }
            out_times[n] = time;
            out_types[n] = ev + 1;
            ++n;
        }
        for (int i = 0; i < S; ++i) out_states[(int64_t)k * S + i] = x[i];
        time = d->obs_time[k]; /* t = y[i].time  hmm_sim.jl:98 */
    }
    return overflow ? -1 : n;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
