// Throughput of 64-bit integer adds on gfx950: v_lshl_add_u64 (what the compiler emits) vs v_add_co_u32 / v_addc_co_u32 pairs.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1;} } while (0)
template <int MODE>
__global__ __launch_bounds__(256) void k(long long *out, int iters, long long seed)
{
    long long a[8];
    for (int q = 0; q < 8; ++q) a[q] = seed + threadIdx.x * (q + 1);
    const long long inc = seed | 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (MODE == 0) a[q] += inc + q;      // compiler's choice
            else {
                unsigned lo = (unsigned)a[q], hi = (unsigned)((unsigned long long)a[q] >> 32);
                const unsigned ilo = (unsigned)(inc + q), ihi = (unsigned)((unsigned long long)(inc + q) >> 32);
                asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(ilo), "v"(ihi) : "vcc");
                a[q] = (long long)(((unsigned long long)hi << 32) | lo);
            }
        }
    }
    long long s = 0;
    for (int q = 0; q < 8; ++q) s ^= a[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    long long *out; CHK(hipMalloc(&out, 256 * 8 * 256 * 8));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 20000, grid = 256 * 8;
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipEventRecord(e0));
            if (mode == 0) k<0><<<grid, 256>>>(out, iters, 12345); else k<1><<<grid, 256>>>(out, iters, 12345);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double adds = (double)grid * 4 /*waves*/ * iters * 8;   // wave-level 64-bit adds
        const double per_simd = adds / (256.0 * 4);
        printf("mode %d: %.3f ms, %.2f cycles per wave-level 64-bit add per SIMD (at 2.4 GHz)\n", mode, best, best * 1e-3 * 2.4e9 / per_simd);
    }
    return 0;
}
