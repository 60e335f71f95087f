// Point estimation on the device: the numsamples × numsamples matrix of pairwise clustering losses of
// getpointestimate(method = "MPEL") (/root/reference/src/pointestimate.jl:49-58) and the pair measures behind
// binderloss / infodist / evaluateclustering (pointestimate.jl:68-99, src/summaries.jl:12-23).
// Included at the end of redclust_hip.hip (same translation unit: shares fail(), HIPCHK, the error buffer).
//
// Every loss the reference offers is a function of the contingency table n_ij of two labelings through just two
// sums, Σ n_ij² and Σ n_ij log n_ij, plus per-sample marginals (Σ a_i², Σ a_i log a_i) that the host prepares
// once.  So the work is: for each pair of samples, histogram n label pairs and reduce the non-zero cells.
//
//   * one WAVE per pair, its contingency table private in LDS (u32[Ka·Kb]); no atomics — per 64 points the wave
//     peels off the distinct (la, lb) keys one at a time (ballot + shuffle) and one lane adds the multiplicity.
//     Samples of a converged chain agree on most points and the host stores all samples in a common
//     cluster-contiguous point order (sorted by the last sample's labels; a contingency table does not care about
//     the order of the points as long as both labelings use the same), so 64 consecutive points usually share one
//     key: ≈1–2 peel iterations per step instead of 64 conflicting atomics.
//   * cells that become non-zero are remembered in a short "touched" list; the final reduction visits (and
//     zeroes) only those, so the table is clean for the wave's next pair without a K² scan.  If the list
//     overflows the wave scans its Ka·Kb cells instead.
//   * a block works through 16×16 tiles of the pair matrix, so its waves keep re-reading the same 32 label rows
//     (u16, L2-resident).
//   * tables too large for LDS (K² · 4 B > ~150 KB) live in global scratch, one per wave, same code.

#include <algorithm>

namespace pe {

constexpr int TILE = 16;          // pair-matrix tile edge
constexpr int TOUCH_CAP = 768;    // touched-list entries per wave
constexpr unsigned short PAD = 0xFFFF;

struct Args {
    const unsigned short *lab;   // m × ld compact labels (0..K_s−1) in the common point order; padding = PAD
    const int *K;                // clusters per sample
    const double *nis;           // Σ a_i² per sample
    const double *ea;            // Σ a_i log a_i per sample
    const double *xlogx;         // c·log c for c = 0..n (host libm, so that terms match a CPU evaluation bit for bit)
    double *L;                   // m × m loss matrix (both triangles written; diagonal stays 0)
    unsigned *gtab;              // global scratch tables (GLOBAL_TAB) or nullptr
    unsigned long long gtab_stride;
    int m, n, ld, kind, tab_cells, ntile;
    double N, logN, t1;
};

// the reference's loss choices (pointestimate.jl:38-47) from the sums; kinds 100 / 101 return the raw sums
__device__ inline double loss_from_sums(const Args &A, double t2, double e_ab, int a, int b)
{
    const double nis = A.nis[a], njs = A.nis[b], e_a = A.ea[a], e_b = A.ea[b];
    const double N = A.N;
    const int kind = A.kind;
    if (kind == 100) return t2;
    if (kind == 101) return e_ab;
    if (kind <= 1) {
        const double t3 = 0.5 * (nis + njs);
        const double Dd = -t2 + t3;
        if (kind == 0) return Dd / A.t1;                                // Mirkin index = Binder loss / C(N,2)
        const double nc = (N * (N * N + 1) - (N + 1) * nis - (N + 1) * njs + 2 * (nis * njs) / N) / (2 * (N - 1));
        const double Aa = A.t1 + t2 - t3;
        const double ari = (A.t1 == nc) ? 0.0 : (Aa - nc) / (A.t1 - nc);
        return 1 - ari;
    }
    const double ha = A.logN - e_a / N, hb = A.logN - e_b / N;
    const double mi = (e_ab - e_a - e_b) / N + A.logN;
    if (kind == 2) return ha + hb - 2 * mi;                             // VI
    return (ha > hb ? ha : hb) - mi;                                    // ID (normalised = false)
}

template <bool GLOBAL_TAB>
__global__ __launch_bounds__(512) void k_pair_losses(Args A)
{
    extern __shared__ unsigned pe_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    unsigned *touched = pe_lds + (size_t)wave * TOUCH_CAP;
    unsigned *tab;
    if (GLOBAL_TAB) {
        tab = A.gtab + ((size_t)blockIdx.x * W + wave) * A.gtab_stride; // arrives zeroed, left zeroed
    } else {
        tab = pe_lds + (size_t)W * TOUCH_CAP + (size_t)wave * A.tab_cells;
        for (int c = lane; c < A.tab_cells; c += 64) tab[c] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    for (int tile = blockIdx.x; tile < A.ntile * A.ntile; tile += gridDim.x) {
        const int ta = tile / A.ntile, tb = tile % A.ntile;
        if (tb < ta) continue;                                          // tile entirely below the diagonal
        for (int q = wave; q < TILE * TILE; q += W) {
            const int a = ta * TILE + q / TILE, b = tb * TILE + q % TILE;
            if (a >= b || b >= A.m) continue;                           // wave-uniform
            const int Kb = A.K[b], cells = A.K[a] * Kb;
            const unsigned short *ra = A.lab + (size_t)a * A.ld, *rb = A.lab + (size_t)b * A.ld;
            int ntouched = 0;                                           // wave-uniform
            for (int base = 0; base < A.ld; base += 256) {
                const ushort4 va = *reinterpret_cast<const ushort4 *>(ra + base + 4 * lane);
                const ushort4 vb = *reinterpret_cast<const ushort4 *>(rb + base + 4 * lane);
                const unsigned short xa[4] = {va.x, va.y, va.z, va.w}, xb[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bool active = xa[j] != PAD;
                    const unsigned key = (unsigned)xa[j] * (unsigned)Kb + (unsigned)xb[j];
                    unsigned long long todo = __ballot(active);
                    while (todo) {                                      // peel one distinct key per turn
                        const int leader = __ffsll((long long)todo) - 1;
                        const unsigned k = (unsigned)__shfl((int)key, leader);
                        const bool same = active && key == k;
                        const unsigned long long mask = __ballot(same);
                        int isnew = 0;
                        if (lane == leader) {
                            const unsigned add = (unsigned)__popcll(mask);
                            unsigned old;
                            if (GLOBAL_TAB) old = atomicAdd(&tab[k], add);
                            else { old = tab[k]; tab[k] = old + add; }
                            isnew = old == 0;
                            if (isnew && ntouched < TOUCH_CAP) touched[ntouched] = k;
                        }
                        ntouched += __shfl(isnew, leader);
                        todo &= ~mask;
                        active = active && !same;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            // reduce the non-zero cells (and leave the table zeroed)
            unsigned long long t2 = 0;
            double e = 0.0;
            if (ntouched <= TOUCH_CAP) {
                for (int t = lane; t < ntouched; t += 64) {
                    const unsigned k = touched[t];
                    unsigned c;
                    if (GLOBAL_TAB) c = atomicExch(&tab[k], 0u);
                    else { c = tab[k]; tab[k] = 0; }
                    t2 += (unsigned long long)c * c;
                    e += A.xlogx[c];
                }
            } else {
                for (int k = lane; k < cells; k += 64) {
                    unsigned c;
                    if (GLOBAL_TAB) c = atomicExch(&tab[k], 0u);
                    else { c = tab[k]; tab[k] = 0; }
                    t2 += (unsigned long long)c * c;
                    e += A.xlogx[c];
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)t2, off), hi = (unsigned)__shfl_xor((int)(unsigned)(t2 >> 32), off);
                t2 += ((unsigned long long)hi << 32) | lo;
                e += __shfl_xor(e, off);
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                const double v = loss_from_sums(A, (double)t2, e, a, b);
                A.L[(size_t)a * A.m + b] = v;
                A.L[(size_t)b * A.m + a] = v;
            }
        }
    }
}

// sum(lossmatrix, dims = 1) (pointestimate.jl:56): one thread per column, rows in ascending order
__global__ void k_colsum(const double *__restrict__ L, int m, double *__restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    double s = 0.0;
    for (int i = 0; i < m; ++i) s += L[(size_t)i * m + j];
    out[j] = s;
}

struct Prep {
    std::vector<unsigned short> lab;
    std::vector<int> K;
    std::vector<double> nis, ea, xlogx;
    int ld = 0, Kmax = 0;
};

// compact labels (dense index by ascending label), marginal sums, common cluster-contiguous point order
static int32_t prepare(const int64_t *samples, int64_t m, int64_t n, Prep &P)
{
    P.ld = (int)((n + 255) / 256 * 256);
    P.lab.assign((size_t)m * P.ld, PAD);
    P.K.resize((size_t)m); P.nis.resize((size_t)m); P.ea.resize((size_t)m);
    P.xlogx.resize((size_t)n + 1);
    P.xlogx[0] = 0.0;
    for (int64_t c = 1; c <= n; ++c) P.xlogx[(size_t)c] = (double)c * std::log((double)c);
    for (int64_t t = 0; t < m * n; ++t)
        if (samples[t] < 1 || samples[t] > n)
            return fail(nullptr, RC_ERR_ARG, "point estimate: label %lld of sample %lld outside 1..n", (long long)samples[t], (long long)(t / n + 1));
    // common point order: stable counting sort by the last sample's labels
    std::vector<int> order((size_t)n), cnt((size_t)n + 2, 0), dense((size_t)n + 1, 0);
    const int64_t *last = samples + (m - 1) * n;
    for (int64_t i = 0; i < n; ++i) cnt[(size_t)last[i] + 1]++;
    for (int64_t l = 1; l <= n + 1; ++l) cnt[(size_t)l] += cnt[(size_t)l - 1];
    for (int64_t i = 0; i < n; ++i) order[(size_t)cnt[(size_t)last[i]]++] = (int)i;
    std::vector<int> c2((size_t)n + 1, 0);
    P.Kmax = 0;
    for (int64_t s = 0; s < m; ++s) {
        const int64_t *x = samples + s * n;
        for (int64_t i = 0; i < n; ++i) c2[(size_t)x[i]]++;
        int K = 0;
        double nis = 0, ea = 0;
        for (int64_t l = 1; l <= n; ++l)
            if (c2[(size_t)l]) {
                dense[(size_t)l] = K++;
                nis += (double)c2[(size_t)l] * (double)c2[(size_t)l];
                ea += P.xlogx[(size_t)c2[(size_t)l]];
                c2[(size_t)l] = 0;
            }
        if (K >= 0xFFFF) return fail(nullptr, RC_ERR_CAPACITY, "point estimate: sample %lld has %d clusters (limit 65534)", (long long)s + 1, K);
        P.K[(size_t)s] = K; P.nis[(size_t)s] = nis; P.ea[(size_t)s] = ea;
        P.Kmax = std::max(P.Kmax, K);
        unsigned short *row = P.lab.data() + (size_t)s * P.ld;
        for (int64_t w = 0; w < n; ++w) row[w] = (unsigned short)dense[(size_t)x[order[(size_t)w]]];
    }
    return RC_OK;
}

struct DevBufs {
    void *p[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    ~DevBufs() { for (void *q : p) if (q) (void)hipFree(q); }
};

#define PECHK(call)                                                                                   \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail(nullptr, (e_ == hipErrorOutOfMemory) ? RC_ERR_OOM : RC_ERR_HIP, "%s failed: %s (%s:%d)", #call, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                   \
    } while (0)

// runs the pair kernel for each kind in `kinds`, copying the m×m result of kind t to outs[t] (host, may be null),
// and the column sums of the LAST kind to colsum (may be null)
static int32_t run(int32_t device, const int64_t *samples, int64_t m, int64_t n, const int *kinds, int nkinds,
                   double **outs, double *colsum, double *kernel_ms, Prep *prep_out)
{
    if (!samples || m < 1 || n < 2) return fail(nullptr, RC_ERR_ARG, "point estimate: need samples, m >= 1 and n >= 2 (got m=%lld n=%lld)", (long long)m, (long long)n);
    if (m > 46340) return fail(nullptr, RC_ERR_ARG, "point estimate: at most 46340 samples (m*m must fit 32 bits), got %lld", (long long)m);
    int ndev = 0;
    PECHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(nullptr, RC_ERR_ARG, "point estimate: device %d not available (%d visible)", device, ndev);
    PECHK(hipSetDevice(device));
    Prep P;
    int32_t rc = prepare(samples, m, n, P);
    if (rc != RC_OK) return rc;
    DevBufs B;
    unsigned short *d_lab; int *d_K; double *d_nis, *d_ea, *d_x, *d_L, *d_cs; unsigned *d_gt = nullptr;
    PECHK(hipMalloc(&B.p[0], P.lab.size() * 2)); d_lab = (unsigned short *)B.p[0];
    PECHK(hipMalloc(&B.p[1], (size_t)m * 4)); d_K = (int *)B.p[1];
    PECHK(hipMalloc(&B.p[2], (size_t)m * 8)); d_nis = (double *)B.p[2];
    PECHK(hipMalloc(&B.p[3], (size_t)m * 8)); d_ea = (double *)B.p[3];
    PECHK(hipMalloc(&B.p[4], P.xlogx.size() * 8)); d_x = (double *)B.p[4];
    PECHK(hipMalloc(&B.p[5], (size_t)m * m * 8)); d_L = (double *)B.p[5];
    PECHK(hipMalloc(&B.p[6], (size_t)m * 8)); d_cs = (double *)B.p[6];
    PECHK(hipMemcpy(d_lab, P.lab.data(), P.lab.size() * 2, hipMemcpyHostToDevice));
    PECHK(hipMemcpy(d_K, P.K.data(), (size_t)m * 4, hipMemcpyHostToDevice));
    PECHK(hipMemcpy(d_nis, P.nis.data(), (size_t)m * 8, hipMemcpyHostToDevice));
    PECHK(hipMemcpy(d_ea, P.ea.data(), (size_t)m * 8, hipMemcpyHostToDevice));
    PECHK(hipMemcpy(d_x, P.xlogx.data(), P.xlogx.size() * 8, hipMemcpyHostToDevice));

    if (P.Kmax > 16384) return fail(nullptr, RC_ERR_CAPACITY, "point estimate: %d clusters in one sample (limit 16384)", P.Kmax);
    Args A{};
    A.lab = d_lab; A.K = d_K; A.nis = d_nis; A.ea = d_ea; A.xlogx = d_x; A.L = d_L;
    A.m = (int)m; A.n = (int)n; A.ld = P.ld;
    A.tab_cells = P.Kmax * P.Kmax;
    A.ntile = (int)((m + TILE - 1) / TILE);
    A.N = (double)n; A.logN = std::log((double)n); A.t1 = (double)n * ((double)n - 1) / 2;
    // launch shape: W waves per block, each with touched list + table in LDS when that fits
    hipDeviceProp_t prop;
    PECHK(hipGetDeviceProperties(&prop, device));
    const size_t lds_max = 160 * 1024 - 1024;
    const size_t per_wave = 4 * ((size_t)TOUCH_CAP + (size_t)A.tab_cells);
    int W = 4;
    while (W > 1 && W * per_wave > lds_max) W >>= 1;
    const bool global_tab = W * per_wave > lds_max;
    const int ntiles_upper = A.ntile * (A.ntile + 1) / 2;
    int grid;
    size_t lds;
    if (global_tab) {
        W = 4;
        lds = 4 * (size_t)W * TOUCH_CAP;
        grid = std::min(A.ntile * A.ntile, prop.multiProcessorCount * 2);
        // keep the scratch under 8 GiB
        while (grid > 1 && (size_t)grid * W * A.tab_cells * 4 > ((size_t)8 << 30)) grid >>= 1;
        A.gtab_stride = (unsigned long long)A.tab_cells;
        PECHK(hipMalloc(&B.p[7], (size_t)grid * W * A.tab_cells * 4)); d_gt = (unsigned *)B.p[7];
        PECHK(hipMemset(d_gt, 0, (size_t)grid * W * A.tab_cells * 4));
        A.gtab = d_gt;
    } else {
        lds = W * per_wave;
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, lds_max / lds));
        grid = std::min(A.ntile * A.ntile, prop.multiProcessorCount * per_cu * 2);
        PECHK(hipFuncSetAttribute((const void *)k_pair_losses<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    (void)ntiles_upper;
    hipEvent_t e0, e1;
    PECHK(hipEventCreate(&e0)); PECHK(hipEventCreate(&e1));
    double ms_total = 0;
    for (int t = 0; t < nkinds; ++t) {
        A.kind = kinds[t];
        PECHK(hipMemset(d_L, 0, (size_t)m * m * 8));
        PECHK(hipEventRecord(e0, 0));
        if (global_tab) k_pair_losses<true><<<grid, 64 * W, lds, 0>>>(A);
        else k_pair_losses<false><<<grid, 64 * W, lds, 0>>>(A);
        PECHK(hipGetLastError());
        PECHK(hipEventRecord(e1, 0));
        PECHK(hipEventSynchronize(e1));
        float ms = 0;
        PECHK(hipEventElapsedTime(&ms, e0, e1));
        ms_total += ms;
        if (outs && outs[t]) PECHK(hipMemcpy(outs[t], d_L, (size_t)m * m * 8, hipMemcpyDeviceToHost));
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (colsum) {
        k_colsum<<<(unsigned)((m + 127) / 128), 128>>>(d_L, (int)m, d_cs);
        PECHK(hipGetLastError());
        PECHK(hipMemcpy(colsum, d_cs, (size_t)m * 8, hipMemcpyDeviceToHost));
    }
    if (kernel_ms) *kernel_ms = ms_total;
    if (prep_out) { prep_out->K = P.K; prep_out->nis = P.nis; prep_out->ea = P.ea; }
    return RC_OK;
}

}  // namespace pe

extern "C" int32_t rc_loss_matrix(int32_t device, const int64_t *samples, int64_t m, int64_t n, int32_t loss,
                                  double *lossmatrix, double *colsum, int64_t *argmin, double *kernel_ms)
{
    if (loss < RC_LOSS_BINDER || loss > RC_LOSS_ID) return fail(nullptr, RC_ERR_ARG, "rc_loss_matrix: invalid loss specifier %d", loss);
    if (m < 1) return fail(nullptr, RC_ERR_ARG, "rc_loss_matrix: no samples");
    std::vector<double> cs((size_t)m);
    const int kinds[1] = {loss};
    double *outs[1] = {lossmatrix};
    int32_t rc = pe::run(device, samples, m, n, kinds, 1, outs, cs.data(), kernel_ms, nullptr);
    if (rc != RC_OK) return rc;
    int64_t best = 0;
    for (int64_t j = 1; j < m; ++j)
        if (cs[(size_t)j] < cs[(size_t)best]) best = j;                 // argmin: first minimum (pointestimate.jl:57)
    if (colsum) std::copy(cs.begin(), cs.end(), colsum);
    if (argmin) *argmin = best;
    return RC_OK;
}

extern "C" int32_t rc_pair_measures(int32_t device, const int64_t *a, const int64_t *b, int64_t n, rc_pair_measures_t *out)
{
    if (!a || !b || !out) return fail(nullptr, RC_ERR_ARG, "rc_pair_measures: NULL argument");
    if (n < 2) return fail(nullptr, RC_ERR_ARG, "rc_pair_measures: need n >= 2");
    std::vector<int64_t> both((size_t)(2 * n));
    std::copy(a, a + n, both.begin());
    std::copy(b, b + n, both.begin() + n);
    double T2[4], E[4];
    const int kinds[2] = {100, 101};
    double *outs[2] = {T2, E};
    pe::Prep P;
    int32_t rc = pe::run(device, both.data(), 2, n, kinds, 2, outs, nullptr, nullptr, &P);
    if (rc != RC_OK) return rc;
    const double t2 = T2[1], e_ab = E[1], nis = P.nis[0], njs = P.nis[1], e_a = P.ea[0], e_b = P.ea[1];
    const double N = (double)n, logN = std::log(N);
    const double t1 = N * (N - 1) / 2, t3 = 0.5 * (nis + njs);
    const double nc = (N * (N * N + 1) - (N + 1) * nis - (N + 1) * njs + 2 * (nis * njs) / N) / (2 * (N - 1));
    const double A = t1 + t2 - t3, Dd = -t2 + t3;
    out->ari = (t1 == nc) ? 0.0 : (A - nc) / (t1 - nc);
    out->ri = A / t1;
    out->mirkin = Dd / t1;
    out->hubert = (A - Dd) / t1;
    out->ha = logN - e_a / N;
    out->hb = logN - e_b / N;
    out->mi = (e_ab - e_a - e_b) / N + logN;
    out->nmi = 2 * out->mi / (out->ha + out->hb);
    out->vi = out->ha + out->hb - 2 * out->mi;
    const double hmax = std::max(out->ha, out->hb);
    out->id = hmax - out->mi;
    out->nid = 1 - out->mi / hmax;
    return RC_OK;
}
