// Issue-rate probe for the VALU instructions of k_propagate's inner loop (gfx950).
// Each kernel runs ITER x 16 copies of one instruction on 8 independent register sets with 8 waves per SIMD
// resident, so the figure is throughput, not latency.  Output: ns per wave-instruction per SIMD and the ratio
// to v_xor_b32 (one full-rate issue = 4 cycles for a wave64 on a 16-lane SIMD).
//   hipcc -O2 --offload-arch=gfx950 tools/valu_rate_probe.hip -o gpurun_out/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define ITER 4096

#define REP8(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)

// 32-bit destination, two 32-bit sources
#define K32(NAME, INS)                                                                          \
    __global__ void NAME(unsigned* out) {                                                         \
        unsigned r[8];                                                                            \
        for (int j = 0; j < 8; ++j) r[j] = threadIdx.x * 2654435761u + j;                          \
        unsigned b = threadIdx.x | 1u;                                                             \
        for (int i = 0; i < ITER; ++i) {                                                           \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile(INS : "+v"(r[j]) : "v"(b)); \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile(INS : "+v"(r[j]) : "v"(b)); \
        }                                                                                          \
        unsigned s = 0;                                                                            \
        for (int j = 0; j < 8; ++j) s ^= r[j];                                                      \
        if (s == 0x12345678u) out[0] = s;                                                          \
    }
// 64-bit destination (u64), 64-bit source
#define K64(NAME, INS)                                                                          \
    __global__ void NAME(unsigned* out) {                                                         \
        unsigned long long r[8];                                                                  \
        for (int j = 0; j < 8; ++j) r[j] = threadIdx.x * 0x9E3779B97F4A7C15ull + j;                 \
        unsigned long long b = threadIdx.x * 77ull + 12345ull;                                     \
        for (int i = 0; i < ITER; ++i) {                                                           \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile(INS : "+v"(r[j]) : "v"(b)); \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile(INS : "+v"(r[j]) : "v"(b)); \
        }                                                                                          \
        unsigned long long s = 0;                                                                  \
        for (int j = 0; j < 8; ++j) s ^= r[j];                                                      \
        if (s == 0x12345678ull) out[0] = (unsigned)s;                                              \
    }
// double destination, double source
#define KF64(NAME, INS)                                                                         \
    __global__ void NAME(unsigned* out) {                                                         \
        double r[8];                                                                              \
        for (int j = 0; j < 8; ++j) r[j] = 1.0 + 1e-3 * (threadIdx.x + j);                          \
        double b = 1.0 + 1e-9 * threadIdx.x;                                                       \
        for (int i = 0; i < ITER; ++i) {                                                           \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile(INS : "+v"(r[j]) : "v"(b)); \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile(INS : "+v"(r[j]) : "v"(b)); \
        }                                                                                          \
        double s = 0;                                                                              \
        for (int j = 0; j < 8; ++j) s += r[j];                                                      \
        if (s == 0.12345) out[0] = 1;                                                              \
    }

K32(k_xor32, "v_xor_b32 %0, %0, %1")
K32(k_add32, "v_add_u32 %0, %0, %1")
K32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 15")
K32(k_perm, "v_perm_b32 %0, %0, %1, %1")
K32(k_andor, "v_and_or_b32 %0, %0, %1, %1")
K32(k_lshladd32, "v_lshl_add_u32 %0, %0, 3, %1")
K32(k_add3, "v_add3_u32 %0, %0, %1, %1")
K32(k_mullo, "v_mul_lo_u32 %0, %0, %1")
K32(k_mulhi, "v_mul_hi_u32 %0, %0, %1")
K32(k_bfe, "v_bfe_u32 %0, %0, 13, 7")
K32(k_fma32, "v_fma_f32 %0, %0, %1, %1")
K32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
K32(k_addco, "v_add_co_u32 %0, vcc, %0, %1")
K32(k_sdwa, "v_add_u32_sdwa %0, sext(%1), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD")
K64(k_lshladd64, "v_lshl_add_u64 %0, %0, 3, %1")
K64(k_add64, "v_lshl_add_u64 %0, %0, 0, %1")
K64(k_lshr64, "v_lshrrev_b64 %0, 11, %0")
K64(k_lshl64, "v_lshlrev_b64 %0, 3, %0")
K64(k_mov64, "v_mov_b64 %0, %1")
KF64(k_addf64, "v_add_f64 %0, %0, %1")
KF64(k_mulf64, "v_mul_f64 %0, %0, %1")
KF64(k_fmaf64, "v_fma_f64 %0, %0, %1, %1")
KF64(k_fmacf64, "v_fmac_f64 %0, %1, %1")
KF64(k_rcpf64, "v_rcp_f64 %0, %0")
KF64(k_divscale, "v_div_scale_f64 %0, vcc, %0, %1, %0")
KF64(k_divfmas, "v_div_fmas_f64 %0, %0, %1, %1")
KF64(k_divfixup, "v_div_fixup_f64 %0, %0, %1, %1")
KF64(k_cmpf64, "v_cmp_gt_f64 vcc, %0, %1")
KF64(k_rndne, "v_rndne_f64 %0, %0")
KF64(k_ldexp, "v_ldexp_f64 %0, %0, 1")
KF64(k_sqrt, "v_sqrt_f64 %0, %0")

__global__ void k_cvt_f64_i32(unsigned* out) {
    double r[8];
    int b = threadIdx.x + 5;
    for (int j = 0; j < 8; ++j) r[j] = 0;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r[j]) : "v"(b));
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r[j]) : "v"(b));
    }
    double s = 0;
    for (int j = 0; j < 8; ++j) s += r[j];
    if (s == 0.12345) out[0] = 1;
}

__global__ void k_ds_read_b128(unsigned* out) {
    __shared__ uint4 tab[128];
    if (threadIdx.x < 128) tab[threadIdx.x] = make_uint4(threadIdx.x, 1, 2, 3);
    __syncthreads();
    unsigned idx = (threadIdx.x * 2654435761u >> 25) & 127u;
    unsigned acc = 0;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            uint4 v = tab[(idx + j * 37 + acc) & 127u];
            acc += v.x;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

struct Entry { const char* name; void (*fn)(unsigned*); };

int main() {
    unsigned* d;
    hipMalloc(&d, 64);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 8, threads = 256;   // 8 workgroups x 4 waves per CU = 8 waves per SIMD
    std::vector<Entry> es = {
        {"v_xor_b32", k_xor32}, {"v_add_u32", k_add32}, {"v_alignbit_b32", k_alignbit}, {"v_perm_b32", k_perm},
        {"v_and_or_b32", k_andor}, {"v_lshl_add_u32", k_lshladd32}, {"v_add3_u32", k_add3},
        {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi}, {"v_bfe_u32", k_bfe}, {"v_fma_f32", k_fma32},
        {"v_cndmask_b32", k_cndmask}, {"v_add_co_u32", k_addco}, {"v_add_u32_sdwa", k_sdwa},
        {"v_lshl_add_u64 (<<3)", k_lshladd64}, {"v_lshl_add_u64 (<<0)", k_add64}, {"v_lshrrev_b64", k_lshr64},
        {"v_add_f64", k_addf64}, {"v_mul_f64", k_mulf64}, {"v_fma_f64", k_fmaf64}, {"v_fmac_f64", k_fmacf64},
        {"v_rcp_f64", k_rcpf64}, {"v_div_scale_f64", k_divscale}, {"v_div_fmas_f64", k_divfmas},
        {"v_div_fixup_f64", k_divfixup}, {"v_cmp_gt_f64", k_cmpf64}, {"v_rndne_f64", k_rndne}, {"v_ldexp_f64", k_ldexp},
        {"v_sqrt_f64", k_sqrt}, {"v_cvt_f64_i32", k_cvt_f64_i32}, {"ds_read_b128 (dependent)", k_ds_read_b128},
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double base = 0;
    printf("{\"device\": \"%s\", \"cus\": %d, \"waves_per_simd\": 8, \"rows\": [\n", prop.name, cus);
    for (size_t k = 0; k < es.size(); ++k) {
        hipLaunchKernelGGL(es[k].fn, dim3(blocks), dim3(threads), 0, 0, d);   // warm-up
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(es[k].fn, dim3(blocks), dim3(threads), 0, 0, d);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        // wave-instructions per SIMD = 8 waves x ITER x 16
        const double ns_per = (double)best * 1e6 / (8.0 * ITER * 16.0);
        if (k == 0) base = ns_per;
        printf("  {\"op\": \"%s\", \"ms\": %.3f, \"ns_per_wave_instr\": %.3f, \"issue_slots\": %.2f}%s\n", es[k].name, best,
               ns_per, ns_per / base, k + 1 < es.size() ? "," : "");
    }
    printf("]}\n");
    return 0;
}
