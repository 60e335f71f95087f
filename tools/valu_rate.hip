// Throughput of the VALU instructions the row reduction is made of, per SIMD, at 1 / 2 / 3 / 4 waves per SIMD (every CU busy):
// cycles per wave-instruction of ONE SIMD = elapsed shader cycles / (instructions per wave x waves on the SIMD).
// Independent instruction streams (8 accumulators per wave) — what the pipe sustains, not dependent latency.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void k(long long *out, int iters, double seed)
{
    double d[8]; long long q[8]; unsigned u[8]; int c[8];
    for (int i = 0; i < 8; ++i) { d[i] = seed + i + threadIdx.x; q[i] = (long long)(seed * 1000) + i * 77 + threadIdx.x; u[i] = (unsigned)q[i]; c[i] = i; }
    const double m = 1.0000001, a = 1e-9;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if (OP == 0) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(m), "v"(a));
                REP8(X)
#undef X
            } else if (OP == 1) {
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (OP == 2) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (OP == 3) {
#define X(i) asm volatile("v_add_co_u32_dpp %0, vcc, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : : "vcc");
                REP8(X)
#undef X
            } else if (OP == 4) {
#define X(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(c[i]));
                REP8(X)
#undef X
            } else if (OP == 5) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(u[i]), "v"(u[(i + 3) & 7]) : "vcc");
                REP8(X)
#undef X
            } else if (OP == 6) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(a));
                REP8(X)
#undef X
            } else if (OP == 7) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(m));
                REP8(X)
#undef X
            } else if (OP == 8) {
#define X(i) asm volatile("v_log_f32 %0, %0" : "+v"(u[i]));
                REP8(X)
#undef X
            } else if (OP == 9) {
#define X(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(u[i]), "+v"(c[i]) : "v"(u[(i + 1) & 7]), "v"(c[(i + 1) & 7]) : "vcc");
                REP8(X)
#undef X
            } else if (OP == 10) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(1.0001f), "v"(0.5f));
                REP8(X)
#undef X
            } else if (OP == 11) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(m));
                REP8(X)
#undef X
            } else if (OP == 12) {
#define X(i) asm volatile("v_mov_b64 %0, %1" : "=v"(d[i]) : "v"(d[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (OP == 13) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 7]) : "vcc");
                REP8(X)
#undef X
            } else if (OP == 14) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (OP == 15) {
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (OP == 16) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (OP == 17) {
#define X(i) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(q[i]));
                REP8(X)
#undef X
            } else if (OP == 18) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (OP == 19) {
#define X(i) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d[i]) : "v"(m), "v"(a));
                REP8(X)
#undef X
            } else if (OP == 20) {
#define X(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(c[i]), "v"(c[(i + 3) & 7]) : "vcc");
                REP8(X)
#undef X
            } else if (OP == 21) {
#define X(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c[i]));
                REP8(X)
#undef X
            } else if (OP == 22) {
#define X(i) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(d[i]));
                REP8(X)
#undef X
            } else if (OP == 23) {
#define X(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(c[i]) : "v"(d[i]));
                REP8(X)
#undef X
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; long long sq = 0; unsigned su = 0;
    for (int i = 0; i < 8; ++i) { s += d[i]; sq += q[i]; su += u[i] + (unsigned)c[i]; }
    if (s == 123.456 && sq == 77 && su == 5) out[1] = 1;
    if (threadIdx.x == 0) out[2 + blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char *name, long long *dbuf, int ncu)
{
    const int iters = 2000, per_wave = iters * 32;
    printf("%-28s", name);
    for (int blocks_per_cu = 1; blocks_per_cu <= 4; ++blocks_per_cu) {   // 256-thread blocks: one wave per SIMD each
        const int nb = ncu * blocks_per_cu;
        hipLaunchKernelGGL(k<OP>, dim3(nb), dim3(256), 0, 0, dbuf, 10, 1.5);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k<OP>, dim3(nb), dim3(256), 0, 0, dbuf, iters, 1.5);
        hipDeviceSynchronize();
        std::vector<long long> h(nb + 2);
        hipMemcpy(h.data(), dbuf, (nb + 2) * sizeof(long long), hipMemcpyDeviceToHost);
        std::sort(h.begin() + 2, h.end());
        const double med = (double)h[2 + nb / 2];
        printf("  %dw/SIMD: %5.2f cyc/inst/SIMD (wave %6.2f)", blocks_per_cu, med / (per_wave * blocks_per_cu), med / per_wave);
    }
    printf("\n");
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    long long *d; hipMalloc(&d, (4 * ncu + 8) * sizeof(long long));
    printf("%s, %d CUs; cycles per wave-instruction per SIMD (shader clock, s_memtime)\n", p.name, ncu);
    run<2>("v_add_u32", d, ncu);
    run<10>("v_fma_f32", d, ncu);
    run<13>("v_cndmask_b32", d, ncu);
    run<16>("v_mad_u32_u24", d, ncu);
    run<14>("v_mul_lo_u32", d, ncu);
    run<15>("v_mul_hi_u32", d, ncu);
    run<3>("v_add_co_u32_dpp", d, ncu);
    run<18>("v_mov_b32_dpp", d, ncu);
    run<9>("v_add_co+v_addc_co (pair)", d, ncu);
    run<1>("v_lshl_add_u64", d, ncu);
    run<17>("v_lshlrev_b64", d, ncu);
    run<12>("v_mov_b64", d, ncu);
    run<5>("v_mad_u64_u32", d, ncu);
    run<20>("v_mad_i64_i32", d, ncu);
    run<0>("v_fma_f64", d, ncu);
    run<19>("v_fmac_f64", d, ncu);
    run<6>("v_add_f64", d, ncu);
    run<7>("v_mul_f64", d, ncu);
    run<11>("v_pk_fma_f32", d, ncu);
    run<4>("v_cvt_f64_i32", d, ncu);
    run<23>("v_cvt_i32_f64", d, ncu);
    run<21>("v_ldexp_f64", d, ncu);
    run<22>("v_frexp_mant_f64", d, ncu);
    run<8>("v_log_f32", d, ncu);
    return 0;
}
