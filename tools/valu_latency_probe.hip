// Dependent-chain latency of the instructions on k_propagate's critical path (gfx950): ONE wave, each instruction
// consumes the previous one's result.  Output: ns and cycles (at the clock measured with s_memtime... here: ns only,
// plus the ratio to v_xor_b32).
//   hipcc -O2 --offload-arch=gfx950 tools/valu_latency_probe.hip -o /tmp/vlp && /tmp/vlp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 8192
#define CHAIN32(NAME, INS)                                                            \
    __global__ void NAME(unsigned* out, unsigned long long* cyc) {                      \
        unsigned r = threadIdx.x * 2654435761u + 1u, b = threadIdx.x | 1u;               \
        unsigned long long t0 = __builtin_readcyclecounter();                           \
        for (int i = 0; i < ITER; ++i) {                                                 \
            _Pragma("unroll") for (int j = 0; j < 16; ++j) asm volatile(INS : "+v"(r) : "v"(b)); \
        }                                                                                \
        unsigned long long t1 = __builtin_readcyclecounter();                           \
        if (r == 0x12345678u) out[0] = r;                                                \
        if (threadIdx.x == 0) cyc[0] = t1 - t0;                                          \
    }
#define CHAIN64(NAME, INS)                                                            \
    __global__ void NAME(unsigned* out, unsigned long long* cyc) {                      \
        unsigned long long r = threadIdx.x * 0x9E3779B97F4A7C15ull + 1ull, b = threadIdx.x * 77ull + 5ull; \
        unsigned long long t0 = __builtin_readcyclecounter();                           \
        for (int i = 0; i < ITER; ++i) {                                                 \
            _Pragma("unroll") for (int j = 0; j < 16; ++j) asm volatile(INS : "+v"(r) : "v"(b)); \
        }                                                                                \
        unsigned long long t1 = __builtin_readcyclecounter();                           \
        if (r == 0x12345678ull) out[0] = (unsigned)r;                                    \
        if (threadIdx.x == 0) cyc[0] = t1 - t0;                                          \
    }
#define CHAINF(NAME, INS)                                                             \
    __global__ void NAME(unsigned* out, unsigned long long* cyc) {                      \
        double r = 1.0 + 1e-3 * threadIdx.x, b = 1.0 + 1e-9 * threadIdx.x;               \
        unsigned long long t0 = __builtin_readcyclecounter();                           \
        for (int i = 0; i < ITER; ++i) {                                                 \
            _Pragma("unroll") for (int j = 0; j < 16; ++j) asm volatile(INS : "+v"(r) : "v"(b)); \
        }                                                                                \
        unsigned long long t1 = __builtin_readcyclecounter();                           \
        if (r == 0.12345) out[0] = 1;                                                    \
        if (threadIdx.x == 0) cyc[0] = t1 - t0;                                          \
    }
CHAIN32(l_xor, "v_xor_b32 %0, %0, %1")
CHAIN32(l_add3, "v_add3_u32 %0, %0, %1, %1")
CHAIN32(l_perm, "v_perm_b32 %0, %0, %1, %1")
CHAIN64(l_add64, "v_lshl_add_u64 %0, %0, 0, %1")
CHAIN64(l_shr64, "v_lshrrev_b64 %0, 1, %0")
CHAINF(l_addf64, "v_add_f64 %0, %0, %1")
CHAINF(l_mulf64, "v_mul_f64 %0, %0, %1")
CHAINF(l_fmaf64, "v_fma_f64 %0, %0, %1, %1")
CHAINF(l_rcpf64, "v_rcp_f64 %0, %0")

__global__ void l_cvt(unsigned* out, unsigned long long* cyc) {   // cvt i32->f64->i32 round trip
    int r = threadIdx.x + 3;
    double d = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d) : "v"(r));
            asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(d));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (r == 0x12345678) out[0] = r;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void l_lds(unsigned* out, unsigned long long* cyc) {   // dependent ds_read_b128
    __shared__ uint4 tab[512];
    for (int i = threadIdx.x; i < 512; i += 64) tab[i] = make_uint4((i * 37 + 11) & 511, 1, 2, 3);
    __syncthreads();
    unsigned idx = threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) idx = tab[idx & 511u].x;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (idx == 0x12345678u) out[0] = idx;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void l_branch(unsigned* out, unsigned long long* cyc) {   // divergent-style branch: saveexec + cbranch per step
    unsigned r = threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER * 16; ++i) {
        if (r & 0x80000000u) break;     // never taken, but not provable: v_cmp + s_and_saveexec + s_cbranch each step
        asm volatile("v_add_u32 %0, %0, 1" : "+v"(r));
        r &= 0x7fffffffu;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (r == 0x12345678u) out[0] = r;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
struct E { const char* n; void (*f)(unsigned*, unsigned long long*); double per; };
int main() {
    unsigned* d; unsigned long long* c;
    hipMalloc(&d, 64); hipMalloc(&c, 8);
    std::vector<E> es = {{"v_xor_b32", l_xor, 16}, {"v_add3_u32", l_add3, 16}, {"v_perm_b32", l_perm, 16}, {"v_lshl_add_u64", l_add64, 16},
                         {"v_lshrrev_b64", l_shr64, 16}, {"v_add_f64", l_addf64, 16}, {"v_mul_f64", l_mulf64, 16}, {"v_fma_f64", l_fmaf64, 16},
                         {"v_rcp_f64", l_rcpf64, 16}, {"v_cvt_f64_i32 + v_cvt_i32_f64 (pair)", l_cvt, 8}, {"ds_read_b128 (dependent address)", l_lds, 16},
                         {"loop step with an exit test (v_cmp + exec mask + s_cbranch)", l_branch, 16}};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("{\"rows\": [\n");
    for (size_t k = 0; k < es.size(); ++k) {
        hipLaunchKernelGGL(es[k].f, dim3(1), dim3(64), 0, 0, d, c); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(es[k].f, dim3(1), dim3(64), 0, 0, d, c); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
        const double n = (double)ITER * es[k].per;
        printf("  {\"op\": \"%s\", \"ns_per_step\": %.2f, \"cycles_per_step_s_memtime\": %.2f}%s\n", es[k].n, ms * 1e6 / n, (double)cy / n,
               k + 1 < es.size() ? "," : "");
    }
    printf("]}\n");
    return 0;
}
