// Does a wave that ends early give its slot to the next workgroup?  Workgroups of 4 waves; in the "early" runs waves
// 1-3 end at once and wave 0 runs a bounded dependent chain.  Every workgroup records start, end (100 MHz wall clock)
// and where its wave 0 ran (XCC, SE, CU, SIMD); the host prints the largest number of workgroups that overlapped on
// one CU and how wave 0 spread over the SIMDs.  early = 2: the surviving wave then meets barriers alone (a barrier waits
// only for the waves of the workgroup that have not ended); early = 3: waves 1-3 run longer than wave 0 and end while it
// waits at its first barrier.
//   hipcc -O3 --offload-arch=gfx950 tools/wave_exit_probe.hip -o /tmp/wave_exit_probe && /tmp/wave_exit_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

struct Rec { unsigned long long t0, t1; unsigned hw, xcc; };

__global__ __launch_bounds__(256) void k_probe(int iters, int early, Rec* rec, double* sink) {
    extern __shared__ double pad[];
    const int wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) pad[0] = 1.0;
    if (early == 3 && wid > 0) {  // these waves end AFTER wave 0 has reached its barrier
        double z = threadIdx.x * 1e-9;
        for (int i = 0; i < 4 * iters; ++i) z = __builtin_fma(z, 1.0000001, 1e-9);
        if (z == 12345.678) sink[1] = z;
        return;
    }
    if (early && wid > 0) return;
    const unsigned long long t0 = wall_clock64();
    double x = threadIdx.x * 1e-9;
    for (int i = 0; i < (early == 3 ? iters / 4 : iters); ++i) x = __builtin_fma(x, 1.0000001, 1e-9);
    if (early >= 2) {  // the surviving wave meets barriers on its own: s_barrier waits for the waves that have not ended
        for (int r = 0; r < 3; ++r) {
            pad[1 + (threadIdx.x & 63)] = x;
            __syncthreads();
            x += pad[1 + ((threadIdx.x + 1) & 63)];
            __syncthreads();
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        rec[blockIdx.x] = Rec{t0, t1, hw, xcc};
    }
    if (x == 12345.678) sink[0] = x + pad[0];
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

int main() {
    const int nwg = 256 * 48, iters = 20000;
    Rec* d_rec; double* d_sink;
    CK(hipMalloc(&d_rec, nwg * sizeof(Rec)));
    CK(hipMalloc(&d_sink, 16));
    std::vector<Rec> rec(nwg);
    for (int lds_kb : {1, 9}) {
        for (int early : {0, 1, 2, 3}) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_probe, dim3(nwg), dim3(256), (size_t)lds_kb << 10, 0, iters, early, d_rec, d_sink);
            CK(hipGetLastError());
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(rec.data(), d_rec, nwg * sizeof(Rec), hipMemcpyDeviceToHost));
            // CU key: xcc, se (hw[15:13]), sh (12), cu (11:8); simd hw[5:4]
            std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> ev;
            int simd_hist[4] = {0, 0, 0, 0};
            for (auto& r : rec) {
                const unsigned cu = ((r.xcc & 0xF) << 16) | ((r.hw >> 8) & 0xFF);
                ev[cu].push_back({r.t0, +1});
                ev[cu].push_back({r.t1, -1});
                simd_hist[(r.hw >> 4) & 3]++;
            }
            int worst = 0; double avg = 0;
            for (auto& kv : ev) {
                std::sort(kv.second.begin(), kv.second.end());
                int cur = 0, mx = 0;
                for (auto& e : kv.second) { cur += e.second; mx = std::max(mx, cur); }
                worst = std::max(worst, mx); avg += mx;
            }
            printf("lds %2d KB early %d: %8.3f ms, CUs seen %zu, max overlapping workgroups per CU %d (avg of per-CU max %.1f), wave-0 SIMD histogram %d %d %d %d\n",
                   lds_kb, early, ms, ev.size(), worst, avg / ev.size(), simd_hist[0], simd_hist[1], simd_hist[2], simd_hist[3]);
        }
    }
    return 0;
}
