// Latency of dependent instructions of ONE wave (what bounds batch_sim's entry loop): ns per dependent VALU add, SALU add,
// LDS read (pointer chase), readlane, and s_memrealtime itself.  hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(long long *out, int iters, int blocks_busy)
{
    __shared__ int chain[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) chain[i] = (i * 37 + 11) & 255;
    __syncthreads();
    if (threadIdx.x >= 64) { __syncthreads(); return; }   // other waves wait at the barrier like the resolver's
    long long t0, t1;
    int v = threadIdx.x, s = iters;
    t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(u + 1));
    }
    t1 = __builtin_amdgcn_s_memrealtime();
    long long valu = t1 - t0;
    t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s) : "s"(u + 1) : "scc");
    }
    t1 = __builtin_amdgcn_s_memrealtime();
    long long salu = t1 - t0;
    int p = threadIdx.x & 255;
    t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) p = chain[p];
    }
    t1 = __builtin_amdgcn_s_memrealtime();
    long long lds = t1 - t0;
    int r = v;
    t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { int x = __builtin_amdgcn_readlane(r, (s + u) & 63); asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "s"(x)); }
    }
    t1 = __builtin_amdgcn_s_memrealtime();
    long long rl = t1 - t0;
    t0 = __builtin_amdgcn_s_memrealtime();
    long long acc = 0;
    for (int i = 0; i < 256; ++i) acc += __builtin_amdgcn_s_memrealtime();
    t1 = __builtin_amdgcn_s_memrealtime();
    long long mt = t1 - t0;
    // scalar branch chain: compare + branch on a scalar
    int sb = s;
    t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters * 16; ++i) { if (__builtin_amdgcn_readfirstlane(sb) & 1) sb = sb * 3 + 1; else sb >>= 1; if (sb == 0) sb = i | 1; }
    t1 = __builtin_amdgcn_s_memrealtime();
    long long br = t1 - t0;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = valu; out[1] = salu; out[2] = lds; out[3] = rl; out[4] = mt; out[5] = v + s + p + r + (int)acc + sb; out[6] = br; }
    __syncthreads();
}
int main()
{
    long long *d; hipMalloc(&d, 64); long long h[8];
    for (int blocks : {1, 256}) {
        const int iters = 2000;
        k<<<blocks, 256>>>(d, iters, blocks); hipDeviceSynchronize();
        k<<<blocks, 256>>>(d, iters, blocks); hipDeviceSynchronize();
        hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        const double n = iters * 16.0;
        printf("blocks %3d: dependent VALU add %.1f ns, SALU add %.1f ns, LDS pointer chase %.1f ns, readlane+add %.1f ns, s_memrealtime %.1f ns, scalar branch step %.1f ns\n", blocks,
               h[0] * 10.0 / n, h[1] * 10.0 / n, h[2] * 10.0 / n, h[3] * 10.0 / n, h[4] * 10.0 / 256, h[6] * 10.0 / n);
    }
    return 0;
}
