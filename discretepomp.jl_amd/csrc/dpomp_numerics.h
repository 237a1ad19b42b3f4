// dpomp_numerics.h -- the numerical spec shared by every kernel: counter-based
// RNG streams, (0,1) uniforms, dlog/dexp, fixed-point weights and offspring
// boundaries.  Every routine is a fixed sequence
// of IEEE-754 double operations (fma = one rounding; the translation unit is
// built with -ffp-contract=off so nothing else fuses), so a CPU evaluation of
// the same spec gives the same bits.  Compiles for gfx950 (hipcc) and, for the
// host-side bit check in tests/test_numerics_host.py, as plain C++.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <stdint.h>
#else  // hipRTC has no <stdint.h>; same widths as the host's types, so kernel-argument structs keep their layout
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
#endif

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
#define DPOMP_HD __host__ __device__ __forceinline__
#define DPOMP_D __device__ __forceinline__
#define DPOMP_TABLE_QUAL static __device__ const
#else
#include <cmath>
#include <cstring>
#define DPOMP_HD static inline
#define DPOMP_D static inline
#define DPOMP_TABLE_QUAL static const
#endif

#include "dpomp_log_table.h"

namespace dpomp {

enum : uint32_t {
    DOMAIN_PROPAGATE = 0u,          // per-particle Gillespie stream of one interval
    DOMAIN_RESAMPLE = 1u,           // the single systematic-resampling uniform of one observation
    DOMAIN_SIMULATE = 2u,           // forward simulation (dpomp_simulate)
    DOMAIN_RESAMPLE_PARTICLE = 3u,  // per-slot uniforms of the stratified / multinomial resamplers
};

DPOMP_HD double bits_to_double(uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)b);
#else
    double d;
    __builtin_memcpy(&d, &b, 8);
    return d;
#endif
}
DPOMP_HD uint64_t double_to_bits(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t)__double_as_longlong(d);
#else
    uint64_t b;
    __builtin_memcpy(&b, &d, 8);
    return b;
#endif
}

// sfc64 state (Doty-Humphrey's "small fast chaotic" generator, PractRand): three 64-bit words and a counter.
struct Stream {
    uint64_t a, b, c, n;
};

// Philox4x32-10 keyed by the call seed; counter = (particle, segment, observation, domain).
DPOMP_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                            uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// sfc64 step: additions, one shift-xor, one shift-add and one rotate -- on gfx950 the 64-bit adds and the
// shift-add are single v_lshl_add_u64 instructions, 10 VALU ops per 64 random bits (xoroshiro128++ needs ~19).
#if defined(__HIP_DEVICE_COMPILE__)
// Device form of the same step.  c + (c << 3) is pinned to one v_lshl_add_u64 (the compiler otherwise rewrites it
// as c * 9 and emits quarter-rate v_mad_u64_u32 pairs).  The 64-bit rotate is two v_alignbit_b32 (the compiler's own
// form is a 64-bit shift, a 32-bit shift and an or), and the add that follows is pinned to v_lshl_add_u64 as well:
// left to the compiler, the halves of the rotate cost a register copy and a second add.  10 VALU instructions.
__device__ __forceinline__ uint64_t stream_next(Stream& s) {
    const uint64_t t = s.a + s.b + s.n;
    s.n += 1;
    s.a = s.b ^ (s.b >> 11);
    uint64_t nb;
    asm("v_lshl_add_u64 %0, %1, 3, %1" : "=v"(nb) : "v"(s.c));
    s.b = nb;
    const uint32_t lo = (uint32_t)s.c, hi = (uint32_t)(s.c >> 32);
    const uint64_t rot = ((uint64_t)__builtin_amdgcn_alignbit(hi, lo, 8) << 32) | __builtin_amdgcn_alignbit(lo, hi, 8);  // rotl(c, 24)
    uint64_t nc;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(nc) : "v"(rot), "v"(t));
    s.c = nc;
    return t;
}
#else
DPOMP_HD uint64_t stream_next(Stream& s) {
    const uint64_t t = s.a + s.b + s.n;
    s.n += 1;
    s.a = s.b ^ (s.b >> 11);
    s.b = s.c + (s.c << 3);
    s.c = ((s.c << 24) | (s.c >> 40)) + t;
    return t;
}
#endif

constexpr int kStreamWarmup = 12;  // rounds discarded after seeding (PractRand's sfc64 seeding discards 12)

// The 128 Philox bits that name a stream; stream_start() expands them into a running generator.  The two are
// separate so that a kernel can queue the 16-byte name in LDS and start the stream where it is consumed.
struct StreamSeed {
    uint32_t w0, w1, w2, w3;
};

DPOMP_HD StreamSeed stream_seed(uint64_t seed, uint32_t seg, uint32_t particle, uint32_t obs, uint32_t domain) {
    uint32_t o[4];
    philox4x32_10(particle, seg, obs, domain, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    return StreamSeed{o[0], o[1], o[2], o[3]};
}

DPOMP_HD Stream stream_start(const StreamSeed& w) {
    Stream s;
    s.a = (uint64_t)w.w0 | ((uint64_t)w.w1 << 32);
    s.b = (uint64_t)w.w2 | ((uint64_t)w.w3 << 32);
    s.c = 0x9E3779B97F4A7C15ull;
    s.n = 1;
#pragma unroll
    for (int i = 0; i < kStreamWarmup; ++i) (void)stream_next(s);
    return s;
}

DPOMP_HD Stream stream_init(uint64_t seed, uint32_t seg, uint32_t particle, uint32_t obs, uint32_t domain) {
    return stream_start(stream_seed(seed, seg, particle, obs, domain));
}

// Top 52 bits -> u = (2k+1) 2^-53 in (0,1).  [1,2) mantissa trick: (d - 1) = k 2^-52 exactly,
// and k 2^-52 + 2^-53 is representable, so this equals (double)(2k+1) * 2^-53 bit for bit.
DPOMP_HD double u01(uint64_t x) {
    const double d = bits_to_double(0x3FF0000000000000ull | (x >> 12));
    // (d - 1) + 2^-53 in ONE add: the constant -(1 - 2^-53) is a double and the exact sum (2k+1) 2^-53 is
    // representable, so the correctly rounded add returns it exactly.
    return d + (-0x1.fffffffffffffp-1);
}

// log(x), positive normal x, table-driven and division-free (the structure of glibc's log):
//   bits(x) - OFF splits x = 2^k z with z in [0.6875, 1.375); subinterval i = next 9 bits;
//   T[i] = (invc, logc) with invc = RN(1/c_i), logc = RN(-log(invc))  (tools/gen_log_table.py);
//   r = fma(z, invc, -1)  (|r| <= 2^-9.5),  log(x) = (k ln2hi + logc) + (k ln2lo + log1p(r)),
//   log1p(r) = r + r^2 (A0 + A1 r + ... + A4 r^4),  A_n = (-1)^(n+1)/(n+2).
// The two subintervals touching 1 have c = 1 (invc = 1, logc = 0), so log(x) keeps full relative
// accuracy as x -> 1.  Max error 1.5 ulp (tests/test_oracle_numerics.py).  17 VALU instructions on gfx950;
// the first version (128 entries, terms to r^8, compensated sum: 1.3 ulp) took 24.  In k_propagate the
// 8 KB table lives in LDS (one ds_read_b128 per event).
constexpr int kLogTableSize = 512;
// On the device the polynomial constants are pinned to scalar registers: a wave-uniform fma addend then comes
// straight from an SGPR pair, where the compiler otherwise parks the literals in VGPRs and spends a v_mov_b64 per
// Horner step re-materialising the accumulator.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double scalar_const(double v) {
    asm("" : "+s"(v));
    return v;
}
#else
inline double scalar_const(double v) { return v; }
#endif

// The innermost Horner step fma(r, A4, A3) has two non-inline constants; a gfx9 VALU instruction reads at most one
// scalar operand, so with both in SGPRs the compiler copies one into a VGPR pair on every evaluation.  A loop that
// evaluates the logarithm per iteration therefore keeps A3 in a VGPR pair of its own, set up once outside the loop.
struct LogRegs {
    double a3;
};
DPOMP_HD LogRegs log_regs() {
    double v = 0x1.999999999999ap-3;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return LogRegs{v};
}

DPOMP_HD double dlog_tab(double x, const double (*T)[2], const LogRegs& lr) {
    const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
    const double A0 = -0.5, A1 = scalar_const(0x1.5555555555555p-2), A2 = scalar_const(-0.25),
                 A3 = lr.a3, A4 = -0x1.5555555555555p-3;
    // only the high word of bits(x) takes part in the range reduction (OFF and the exponent mask have a zero low word)
    const uint64_t ix = double_to_bits(x);
    const uint32_t hx = (uint32_t)(ix >> 32);
    const uint32_t tmp = hx - 0x3FE60000u;
    const int i = (int)((tmp >> 11) & 511u);
    const int32_t k = (int32_t)tmp >> 20;
    const double z = bits_to_double(((uint64_t)(hx - (tmp & 0xFFF00000u)) << 32) | (uint32_t)ix);
    const double invc = T[i][0], logc = T[i][1];
    const double r = __builtin_fma(z, invc, -1.0);
    const double kd = (double)k;
    const double w = __builtin_fma(kd, Ln2hi, logc);
    const double s = __builtin_fma(kd, Ln2lo, r);
    const double r2 = r * r;
    const double p = __builtin_fma(r, __builtin_fma(r, __builtin_fma(r, __builtin_fma(r, A4, A3), A2), A1), A0);
    return w + __builtin_fma(r2, p, s);
}
DPOMP_HD double dlog_tab(double x, const double (*T)[2]) { return dlog_tab(x, T, LogRegs{0x1.999999999999ap-3}); }
DPOMP_D double dlog(double x) { return dlog_tab(x, kLogTable); }

#if defined(__HIP_DEVICE_COMPILE__)
// Waiting time log(u) / total with the IEEE quotient but without the range machinery of the compiler's division.
// The compiler expands a / b into v_div_scale x2, v_rcp_f64, two Newton steps, q0 = a y, r = a - b q0,
// v_div_fmas (q0 + r y) and v_div_fixup; the scale / fmas / fixup instructions only act when an operand or the
// quotient is zero, subnormal, infinite, NaN or needs pre-scaling against overflow.  For 2^-200 <= b <= 2^200 and
// a = 0 or 2^-200 <= |a| <= 2^200 they are identities (a = 0 gives 0 either way), and the remaining eight instructions
// are the same Newton-Markstein sequence, so the result is the correctly rounded quotient -- what `/` gives on the
// host (checked against `/` by dpomp_selftest_numerics and, implicitly, by every parity test).  The other users --
// the quotient inside dexp and the scaled cumulative weight c / W -- are in that range by construction.
// |log(u)| is in [2^-53, 37] by construction;
// the rate total is tested (one integer compare on its high word) and anything else -- a zero total, i.e. a
// particle that can no longer move, included -- leaves the fast event loop for one with the plain division.
__device__ __forceinline__ bool rate_in_fast_range(double total) {
    return (uint32_t)(__double2hiint(total) - 0x33700000) < 0x19000000u;  // 2^-200 <= total < 2^200, positive, finite
}
__device__ __forceinline__ double div_in_range(double a, double b) {
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    const double q = a * y;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}
#else  // host pass of the HIP compilation and the host twin library: same predicate, the plain quotient
DPOMP_HD bool rate_in_fast_range(double total) {
    return (uint32_t)((uint32_t)(double_to_bits(total) >> 32) - 0x33700000u) < 0x19000000u;
}
DPOMP_HD double div_in_range(double a, double b) { return a / b; }
#endif

// exp(x), -37 <= x <= 0.
DPOMP_HD double dexp(double x) {
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                 P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    const int32_t k = (int32_t)(invln2 * x - 0.5);
    const double t = (double)k;
    const double hi = x - t * ln2HI;
    const double lo = t * ln2LO;
    const double r = hi - lo;
    const double tt = r * r;
    const double c = r - tt * (P1 + tt * (P2 + tt * (P3 + tt * (P4 + tt * P5))));
    // (|r c| is 0 or in [2^-140, 0.2] and 2 - c in [1.6, 2.4]: the short division gives the IEEE quotient)
    const double y = 1.0 - ((lo - div_in_range(r * c, 2.0 - c)) - hi);
    return bits_to_double(double_to_bits(y) + ((uint64_t)(int64_t)k << 52));
}

DPOMP_HD double u64_to_double(uint64_t v) {
    return (double)(uint32_t)(v >> 32) * 4294967296.0 + (double)(uint32_t)v;
}

DPOMP_HD int weight_shift(int64_t P) {
    int n = 0;
    while (((int64_t)1 << n) < P) ++n;
    const int sh = 62 - n;
    return sh > 52 ? 52 : sh;
}

// Max-shifted weight as fixed point: rint(exp(g - m) 2^sh); 0 for -inf/NaN and below e^-37.
DPOMP_HD uint64_t fixed_weight(double g, double m, int sh) {
    if (!(g > -__builtin_inf())) return 0;
    const double x = g - m;
    if (x < -37.0) return 0;
    return (uint64_t)__builtin_rint(dexp(x) * (double)((uint64_t)1 << sh));
}

// Offspring boundary N(c) = clamp(ceil(c/W P - r), 0, P), N(W) = P: slots [N(c_{j-1}), N(c_j)) copy
// particle j.  (c == W is pinned to P because P - r can round down to P - 1 for r within 2^-53 of 1.)
DPOMP_HD int64_t offspring_end(uint64_t c, uint64_t W, int64_t P, double r) {
    if (c >= W) return P;
    const double Wd = u64_to_double(W);
    const double pos = div_in_range(u64_to_double(c), Wd) * (double)P - r;  // (0 <= c < W, 1 <= W < 2^63: in range)
    const double n = __builtin_ceil(pos);
    if (!(n > 0.0)) return 0;
    if (n > (double)P) return P;
    return (int64_t)n;
}

// log mean weight of one observation: m + log((W 2^-sh) / P), i.e. log(cum_weight[end]/P)
// of hmm_particle_filter.jl:60 with the max factored out.
DPOMP_D double log_mean_weight(double m, uint64_t W, int sh, int64_t P) {
    const double Wd = u64_to_double(W);
    return m + dlog((Wd * (1.0 / (double)((uint64_t)1 << sh))) / (double)P);
}

// ESS-adaptive mode: resample iff W^2 / sum w^2 < ess_frac * P, from the exact integer sums (sum w^2 as 128 bits).
DPOMP_HD bool ess_resample(uint64_t W, uint64_t s2_hi, uint64_t s2_lo, double ess_frac, int64_t P) {
    if (W == 0) return true;
    const double Wd = u64_to_double(W);
    const double s2 = u64_to_double(s2_hi) * 18446744073709551616.0 + u64_to_double(s2_lo);
    return (Wd * Wd) / s2 < ess_frac * (double)P;
}

// log sum exp of the (possibly carried) log-weights: m + log(W 2^-sh).
DPOMP_D double log_sum_weight(double m, uint64_t W, int sh) {
    return m + dlog(u64_to_double(W) * (1.0 / (double)((uint64_t)1 << sh)));
}

// Order-preserving u64 key of a double, for integer atomicMax; key 0 sorts below every value.
DPOMP_HD uint64_t ordered_key(double d) {
    const uint64_t b = double_to_bits(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
DPOMP_HD double ordered_value(uint64_t key) {
    if (key == 0) return -__builtin_inf();
    return bits_to_double((key >> 63) ? (key & 0x7FFFFFFFFFFFFFFFull) : ~key);
}

}  // namespace dpomp
