// bulk_tune.hip — standalone tuning harness for the row-bucket reduction (k_bulk of redclust_hip.hip).
// Not part of the product; results are copied into profiles/ and the winning shape into the library.
//   hipcc --offload-arch=gfx950 -O3 -o bulk_tune bulk_tune.hip && ./bulk_tune [n] [K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef long long ll2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

struct V { int n, ld; const long long *Dq, *Lq; long long *SD, *SL; const int *perm, *pslot; };

__device__ __forceinline__ void flush2(const V &v, int slot, int i, long long d0, long long d1, long long l0, long long l1)
{
    u64 *pd = (u64 *)(v.SD + (size_t)slot * v.ld + i);
    u64 *pl = (u64 *)(v.SL + (size_t)slot * v.ld + i);
    __hip_atomic_fetch_add(pd, (u64)d0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(pd + 1, (u64)d1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(pl, (u64)l0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(pl + 1, (u64)l1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool NT> __device__ __forceinline__ ll2 ld16(const long long *p)
{
    if constexpr (NT) return __builtin_nontemporal_load((const ll2 *)p);
    else return *(const ll2 *)p;
}

// PPL = points per lane (2 or 4); block = TPB threads covering TPB*PPL columns
template <int PPL, int U, bool NT, int TPB>
__global__ __launch_bounds__(TPB) void k_bulk(V v, int rows_per_split)
{
    constexpr int NV = PPL / 2;
    const int i = (blockIdx.x * TPB + threadIdx.x) * 2;        // first pair; further pairs at + TPB*2*q
    const int p0 = blockIdx.y * rows_per_split, p1 = min(v.n, p0 + rows_per_split);
    if (p0 >= p1) return;
    const size_t ld = v.ld;
    long long aD[PPL] = {0}, aL[PPL] = {0};
    int cur = v.pslot[p0];
    const int base = blockIdx.x * TPB * PPL + threadIdx.x * 2;
    auto flush = [&](int slot) {
#pragma unroll
        for (int q = 0; q < NV; ++q) flush2(v, slot, base + q * TPB * 2, aD[2*q], aD[2*q+1], aL[2*q], aL[2*q+1]);
    };
    (void)i;
    int p = p0;
    for (; p + U <= p1; p += U) {
        int j[U], s[U];
        ll2 d[U][NV], l[U][NV];
#pragma unroll
        for (int u = 0; u < U; ++u) { j[u] = v.perm[p + u]; s[u] = v.pslot[p + u]; }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                d[u][q] = ld16<NT>(v.Dq + (size_t)j[u] * ld + base + q * TPB * 2);
                l[u][q] = ld16<NT>(v.Lq + (size_t)j[u] * ld + base + q * TPB * 2);
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (s[u] != cur) {
                flush(cur);
#pragma unroll
                for (int q = 0; q < PPL; ++q) { aD[q] = 0; aL[q] = 0; }
                cur = s[u];
            }
#pragma unroll
            for (int q = 0; q < NV; ++q) { aD[2*q] += d[u][q].x; aD[2*q+1] += d[u][q].y; aL[2*q] += l[u][q].x; aL[2*q+1] += l[u][q].y; }
        }
    }
    for (; p < p1; ++p) {
        const int j = v.perm[p], s = v.pslot[p];
        if (s != cur) { flush(cur);
#pragma unroll
            for (int q = 0; q < PPL; ++q) { aD[q] = 0; aL[q] = 0; }
            cur = s; }
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            ll2 d = ld16<NT>(v.Dq + (size_t)j * ld + base + q * TPB * 2), l = ld16<NT>(v.Lq + (size_t)j * ld + base + q * TPB * 2);
            aD[2*q] += d.x; aD[2*q+1] += d.y; aL[2*q] += l.x; aL[2*q+1] += l.y;
        }
    }
    flush(cur);
}

// ceiling: stream both matrices, sum into one word per thread (no bucket logic, no atomics but one)
__global__ __launch_bounds__(256) void k_stream(const long long *A, const long long *B, size_t total2, u64 *out)
{
    long long acc = 0;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total2; t += (size_t)gridDim.x * 256) {
        ll2 a = ((const ll2 *)A)[t], b = ((const ll2 *)B)[t];
        acc += a.x + a.y + b.x + b.y;
    }
    if (acc == 0x123456789) atomicAdd(out, (u64)acc);
}

template <int PPL, int U, bool NT, int TPB>
float run(const char *name, V v, int kcap, int rows, int iters, std::vector<long long> *check)
{
    dim3 g(v.ld / (TPB * PPL), (v.n + rows - 1) / rows);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float best = 1e9, tot = 0;
    for (int it = 0; it < iters + 2; ++it) {
        CHK(hipMemset(v.SD, 0, (size_t)kcap * v.ld * 8)); CHK(hipMemset(v.SL, 0, (size_t)kcap * v.ld * 8));
        CHK(hipEventRecord(e0));
        k_bulk<PPL, U, NT, TPB><<<g, TPB>>>(v, rows);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 2) { best = std::min(best, ms); tot += ms; }
    }
    std::vector<long long> h((size_t)kcap * v.ld);
    CHK(hipMemcpy(h.data(), v.SD, h.size() * 8, hipMemcpyDeviceToHost));
    bool ok = true;
    if (check->empty()) *check = h; else ok = (h == *check);
    const double gb = 2.0 * v.n * (double)v.n * 8 / 1e9;
    printf("%-34s rows=%4d grid=%4dx%4d  avg %.1f us  best %.1f us  %.0f GB/s (best %.0f)  %s\n", name, rows, g.x, g.y,
           tot / iters * 1e3, best * 1e3, gb / (tot / iters * 1e-3), gb / (best * 1e-3), ok ? "ok" : "MISMATCH");
    return tot / iters;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 8192, K = argc > 2 ? atoi(argv[2]) : 50, kcap = 128;
    const int ld = ((n + 1023) / 1024) * 1024;
    std::vector<long long> hD((size_t)n * ld), hL((size_t)n * ld);
    srand(1);
    for (size_t t = 0; t < hD.size(); ++t) { hD[t] = ((long long)rand() << 20) ^ rand(); hL[t] = -(((long long)rand() << 18) ^ rand()); }
    std::vector<int> perm(n), pslot(n);
    for (int p = 0; p < n; ++p) { perm[p] = p; pslot[p] = (int)((long long)p * K / n); }
    V v; v.n = n; v.ld = ld;
    long long *Dq, *Lq, *SD, *SL; int *dp, *ds; u64 *out;
    CHK(hipMalloc(&Dq, hD.size() * 8)); CHK(hipMalloc(&Lq, hL.size() * 8));
    CHK(hipMalloc(&SD, (size_t)kcap * ld * 8)); CHK(hipMalloc(&SL, (size_t)kcap * ld * 8));
    CHK(hipMalloc(&dp, n * 4)); CHK(hipMalloc(&ds, n * 4)); CHK(hipMalloc(&out, 8));
    CHK(hipMemcpy(Dq, hD.data(), hD.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(Lq, hL.data(), hL.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dp, perm.data(), n * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(ds, pslot.data(), n * 4, hipMemcpyHostToDevice));
    v.Dq = Dq; v.Lq = Lq; v.SD = SD; v.SL = SL; v.perm = dp; v.pslot = ds;
    {   // streaming ceiling
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        for (int grid : {2048, 4096, 8192, 16384}) {
            float tot = 0, best = 1e9;
            for (int it = 0; it < 12; ++it) {
                CHK(hipEventRecord(e0));
                k_stream<<<grid, 256>>>(Dq, Lq, (size_t)n * ld / 2, out);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (it >= 2) { tot += ms; best = std::min(best, ms); }
            }
            printf("stream ceiling grid=%5d: avg %.1f us best %.1f us  %.0f GB/s\n", grid, tot / 10 * 1e3, best * 1e3, 2.0 * n * (double)ld * 8 / 1e9 / (tot / 10 * 1e-3));
        }
    }
    std::vector<long long> check;
    const int IT = 10;
    for (int rows : {64, 128, 256, 512}) run<2, 8, false, 256>("ppl2 U8 tpb256", v, kcap, rows, IT, &check);
    for (int rows : {64, 128, 256}) run<2, 4, false, 256>("ppl2 U4 tpb256", v, kcap, rows, IT, &check);
    for (int rows : {128, 256}) run<2, 16, false, 256>("ppl2 U16 tpb256", v, kcap, rows, IT, &check);
    for (int rows : {64, 128, 256}) run<2, 8, true, 256>("ppl2 U8 NT tpb256", v, kcap, rows, IT, &check);
    for (int rows : {32, 64, 128, 256}) run<4, 4, false, 256>("ppl4 U4 tpb256", v, kcap, rows, IT, &check);
    for (int rows : {32, 64, 128}) run<4, 8, false, 256>("ppl4 U8 tpb256", v, kcap, rows, IT, &check);
    for (int rows : {32, 64, 128}) run<4, 4, true, 256>("ppl4 U4 NT tpb256", v, kcap, rows, IT, &check);
    for (int rows : {64, 128, 256}) run<2, 8, false, 512>("ppl2 U8 tpb512", v, kcap, rows, IT, &check);
    for (int rows : {64, 128, 256}) run<2, 8, false, 128>("ppl2 U8 tpb128", v, kcap, rows, IT, &check);
    for (int rows : {64, 128, 256}) run<2, 8, false, 64>("ppl2 U8 tpb64", v, kcap, rows, IT, &check);
    return 0;
}
