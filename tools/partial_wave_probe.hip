// Does a wave64 VALU instruction cost fewer cycles when only the first 32 or 16 lanes are active?  One wave per
// workgroup, one workgroup (a lone wave on its SIMD); 8 independent v_fma_f64 chains (issue-bound, not latency-bound)
// and one dependent chain (latency-bound), with EXEC = 64, 32, 16 and 1 lanes.  Cycles from s_memtime.
//   hipcc -O2 --offload-arch=gfx950 tools/partial_wave_probe.hip -o gpurun_out/partial_wave_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITER 2048

__global__ void k_indep(unsigned long long* cyc, double* sink, int lanes) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = 1.0 + 1e-3 * (threadIdx.x + j);
    const double b = 1.0 + 1e-9 * threadIdx.x;
    unsigned long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < lanes) {
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < ITER; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(r[j]) : "v"(b));
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(r[j]) : "v"(b));
        }
        t1 = __builtin_readcyclecounter();
    }
    double s = 0;
    for (int j = 0; j < 8; ++j) s += r[j];
    sink[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void k_dep(unsigned long long* cyc, double* sink, int lanes) {
    double r = 1.0 + 1e-3 * threadIdx.x;
    const double b = 1.0 + 1e-9 * threadIdx.x;
    unsigned long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < lanes) {
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < ITER; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(r) : "v"(b));
        }
        t1 = __builtin_readcyclecounter();
    }
    sink[threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    unsigned long long* cyc;
    double* sink;
    if (hipMalloc(&cyc, 8) != hipSuccess || hipMalloc(&sink, 64 * 8) != hipSuccess) return 1;
    int mhz = 0;
    hipDeviceGetAttribute(&mhz, hipDeviceAttributeWallClockRate, 0);
    for (int lanes : {64, 48, 32, 16, 1}) {
        unsigned long long a = 0, b = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k_indep, dim3(1), dim3(64), 0, 0, cyc, sink, lanes);
            hipMemcpy(&a, cyc, 8, hipMemcpyDeviceToHost);
            hipLaunchKernelGGL(k_dep, dim3(1), dim3(64), 0, 0, cyc, sink, lanes);
            hipMemcpy(&b, cyc, 8, hipMemcpyDeviceToHost);
        }
        printf("active lanes %2d: independent v_fma_f64 %.2f counter ticks per instruction, dependent %.2f\n", lanes,
               (double)a / (ITER * 16.0), (double)b / (ITER * 16.0));
    }
    printf("(s_memtime ticks at the constant wall-clock rate, %d kHz; the shader clock is higher)\n", mhz);
    return 0;
}
