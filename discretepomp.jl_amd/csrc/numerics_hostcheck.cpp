// Host compilation of dpomp_numerics.h behind a C ABI, so a CPU-only test can compare the
// product's numerical spec with the oracle's restatement bit for bit.  Not a filter path.
#include "dpomp_numerics.h"

extern "C" {
void hc_uniforms(uint64_t seed, uint32_t seg, uint32_t p, uint32_t k, uint32_t domain, int64_t n, double* out) {
    dpomp::Stream s = dpomp::stream_init(seed, seg, p, k, domain);
    for (int64_t i = 0; i < n; ++i) out[i] = dpomp::u01(dpomp::stream_next(s));
}
void hc_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
    uint32_t o[4];
    dpomp::philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], o);
    for (int i = 0; i < 4; ++i) out[i] = o[i];
}
double hc_dlog(double x) { return dpomp::dlog(x); }
double hc_dexp(double x) { return dpomp::dexp(x); }
int hc_weight_shift(int64_t P) { return dpomp::weight_shift(P); }
uint64_t hc_fixed_weight(double g, double m, int sh) { return dpomp::fixed_weight(g, m, sh); }
int64_t hc_offspring_end(uint64_t c, uint64_t W, int64_t P, double r) { return dpomp::offspring_end(c, W, P, r); }
double hc_log_mean_weight(double m, uint64_t W, int sh, int64_t P) { return dpomp::log_mean_weight(m, W, sh, P); }
double hc_ordered_roundtrip(double d) { return dpomp::ordered_value(dpomp::ordered_key(d)); }
uint64_t hc_ordered_key(double d) { return dpomp::ordered_key(d); }
int hc_ess_resample(uint64_t W, uint64_t hi, uint64_t lo, double frac, int64_t P) { return dpomp::ess_resample(W, hi, lo, frac, P) ? 1 : 0; }
double hc_log_sum_weight(double m, uint64_t W, int sh) { return dpomp::log_sum_weight(m, W, sh); }
}
