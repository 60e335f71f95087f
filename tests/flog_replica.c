/* CPU replica of the sweep kernel's table-driven logarithms (rc_flog / rc_flog1p / rc_gumbel in
 * redclust.jl_amd/csrc/redclust_hip.hip, table built in create_impl) — test infrastructure only: the same operations in the
 * same order (fma where the kernel uses fma), so that the accuracy claim of DESIGN.md section 4 (<= 1.5 ulp against the true
 * value on the domains the sweep feeds it) can be checked without a GPU.  tests/test_flog_cpu.py compiles and runs it;
 * tests/test_gpu_logs.py checks the device routine itself.  The reference evaluates log1p / log of Float64
 * (src/mcmc.jl:223-241, src/utils.jl:4). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double invc[128], logc[128];

static void build_table(void)
{
    for (int i = 0; i < 128; ++i) {
        const uint64_t b0 = ((uint64_t)(0x3fe60000u + ((uint32_t)i << 13))) << 32, b1 = ((uint64_t)(0x3fe60000u + ((uint32_t)(i + 1) << 13))) << 32;
        double z0, z1;
        memcpy(&z0, &b0, 8); memcpy(&z1, &b1, 8);
        if (i == 79 || i == 80) { invc[i] = 1.0; logc[i] = 0.0; continue; }
        const long double cc = 0.5L * ((long double)z0 + (long double)z1);
        invc[i] = (double)(1.0L / cc);
        logc[i] = (double)(-logl((long double)invc[i]));
    }
}

static double flog(double x, double extra)
{
    uint64_t b;
    memcpy(&b, &x, 8);
    const int hi = (int)(b >> 32);
    const int tmp = hi - 0x3fe60000, k = tmp >> 20;
    const uint64_t zb = ((uint64_t)(uint32_t)(hi - (tmp & (int)0xfff00000u)) << 32) | (uint32_t)b;
    double z;
    memcpy(&z, &zb, 8);
    const int i = (tmp >> 13) & 127;
    const double r = fma(z, invc[i], -1.0), kd = (double)k;
    const double h = fma(kd, 0x1.62e42fefa3800p-1, logc[i]);
    const double t1 = h + r;
    const double lo = fma(kd, 0x1.ef35793c76730p-45, (h - t1) + r) + extra;
    double p = fma(r, -1.0 / 8, 1.0 / 7);
    p = fma(r, p, -1.0 / 6); p = fma(r, p, 1.0 / 5); p = fma(r, p, -1.0 / 4); p = fma(r, p, 1.0 / 3); p = fma(r, p, -0.5);
    return fma(r * r, p, lo) + t1;
}
static double flog1p(double x)
{
    const double u = 1.0 + x, v = u - 1.0;
    const double c = (1.0 - (u - v)) + (x - v);
    return flog(u, c * (1.0 / u));      /* (the kernel multiplies by v_rcp_f64(u): c / u is below 2^-53, its last bits do not matter) */
}
static double ulps(double got, long double ref)
{
    const double rd = (double)ref, u = fabs(nextafter(rd, INFINITY) - rd);
    return (double)(fabsl((long double)got - ref) / (long double)u);
}

int main(int argc, char **argv)
{
    const long m = argc > 1 ? atol(argv[1]) : 2000000;
    build_table();
    srand48(7);
    double e_log = 0, e_u = 0, e_1p = 0, e_g = 0, gmax = -1e9;
    for (long t = 0; t < m; ++t) {
        const double x = exp((drand48() - 0.5) * 160.0);
        const double e = ulps(flog(x, 0.0), logl((long double)x));
        if (e > e_log) e_log = e;
        double u = (t & 1) ? 1.0 - ldexp(drand48(), -(int)(lrand48() % 52 + 1)) : drand48();
        if (u > 0.0 && u < 1.0) {
            const double eu = ulps(flog(u, 0.0), logl((long double)u));
            if (eu > e_u) e_u = eu;
            const double g = -flog(-flog(u, 0.0), 0.0);
            const double eg = fabs((double)((long double)g - (-logl(-logl((long double)u)))));
            if (eg > e_g) e_g = eg;
            if (g > gmax) gmax = g;
        }
        const double y = exp((drand48() - 0.67) * 60.0);
        const double e1 = ulps(flog1p(y), log1pl((long double)y));
        if (e1 > e_1p) e_1p = e1;
    }
    const double top = -flog(-flog(1.0 - ldexp(1.0, -53), 0.0), 0.0);    /* the largest noise value: u = 1 - 2^-53 */
    printf("%.4f %.4f %.4f %.3e %.6f %.17g %.17g\n", e_log, e_u, e_1p, e_g, top, flog(1.0, 0.0), flog1p(0.0));
    return 0;
}
