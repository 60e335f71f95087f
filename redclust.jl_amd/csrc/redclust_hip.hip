// redclust_hip.hip — MI355X (gfx950 / CDNA4) implementation of RedClust.jl's Gibbs label sweep and its
// observables behind the C ABI of include/redclust_hip.h.  Written for gfx950 only.  (pointestimate.inc.hip,
// chain.inc.hip and chains.inc.hip — the MPEL loss matrix, the native iteration loop and the chain-parallel driver — are
// included at the end of this file.)
//
// Reference path (RedClust.jl v1.2.2, paths under the reference checkout):
//   sample_labels_Gibbs!  src/mcmc.jl:158-256     loglik  src/mcmc.jl:1-56     logprior  src/mcmc.jl:58-78
//   sample_logweights     src/utils.jl:2-6        matsum/vecsum  src/utils.jl:9-38
//   adjacencymatrix / sortlabels  src/utils.jl:59-74      MCMCData / MCMCState  src/types.jl:131-157
//
// Design (DESIGN.md has the long form):
//   * D and logD live in HBM as 64-bit FIXED-POINT integers (q = rint(x·2^e), e chosen so that any sum
//     of n entries fits in int64).  Every row/block sum is therefore exact and independent of the order of
//     summation: launch geometry, atomics and incremental updates cannot change a single bit of a result.
//   * S[k][i] = Σ_j D[i,j]·[c_j in slot k]  (and the same for logD) is the n×K sufficient-statistic
//     table of the sweep (the matsum(D,[i],clust_k) of mcmc.jl:210-213 for every i and k at once).
//     A row-reduction kernel recomputes it from the matrices once per sweep; three of them, all exact:
//       k_bulk       every entry of D and logD, any point order (lanes own columns, the wave walks rows grouped by
//                    cluster: register accumulators, coalesced 16 B/lane loads, no LDS) — 2·n²·8 B;
//       k_bulk_sym   upper triangle only, 32×128 LDS tiles read in both orientations, persistent blocks — n²·8 B;
//       k_bulk_syml  upper triangle only, one WAVE per 64×128 unit, no block barriers, wave-private LDS transposition
//                    + fused-DPP reduction; the default together with the DERIVED logD: when the caller gives only D,
//                    logD is not stored — every consumer evaluates rint(log(Dq)·2^eL) with one shared table log
//                    (rc_qlog), so this kernel reads only D's upper triangle, n²/2·8 B.
//     Points are kept in an internal cluster-contiguous order (pi / ipi), invisible through the ABI.
//   * The sequential dependence of the sweep is resolved exactly by speculation: every point is scored
//     and drawn in parallel under the committed state; the ordered batch of points whose draw differs from their
//     label is simulated in order (moves, births, deaths, relabelings), every later point is drawn again under
//     the batch entries that precede it, and the entries before the first point whose draw changes are
//     committed together (S corrected exactly — integers).  The Gumbel noise of a candidate is keyed by
//     (sweep, point, cluster LABEL), so births and deaths leave the other candidates' noise alone and batch
//     like plain moves.  Draws are deterministic functions of (state, counter-based uniforms), so the result
//     is identical to the sequential loop.  k_resolve runs this inside ONE persistent launch (two grid
//     barriers per batch); at stationarity it is a single scoring pass.
//     Inside a sweep the scores of (point, cluster) pairs that no change has touched are kept (score cache), the batch's
//     largest group of corrections is shared by all candidate streams of a point, the grid barrier is two-level, and the
//     epilogue (label snapshot, point order of the next layout) is built by all blocks.
//   * Sweeps are software-pipelined: sweep t (row reduction, then k_resolve) lives on one stream per sweep
//     parity; the row reduction of sweep t+1 runs on the other stream concurrently with k_resolve(t) — three
//     reduction blocks and one resolver block per CU — under the labels known before sweep t; the label changes
//     of sweep t are added to that table by k_resolve with the same commutative integer atomics, so the table
//     k_resolve(t+1) reads is exactly the row sums under the labels after sweep t (k_resolve clears the
//     generation two sweeps ahead).  A third stream carries the observables.
//   * The iteration loop (chain.inc.hip) speculates that split–merge proposals are rejected: proposals are
//     decided by worker threads on state snapshots while the sweeps run on; chains.inc.hip runs one chain per
//     GPU and merges the co-clustering counts over RCCL.
//   * Scores use the regrouped arithmetic of SURVEY.md §7 H2 (size-only lgamma terms tabulated on the
//     host in long double; log(β+S) = log β + log1p(S/β)); terms common to all candidates (L2_i, the
//     subtracted minimum) are dropped — they cannot change the Gumbel-max argmax.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "../../include/redclust_hip.h"

// Environment switches (INTEGRATION.md "Environment switches").  The OPERATIONAL ones are read when a context is created (or, for the
// RC_CHAIN_* ones, copied into the context then) — never per sweep; the few that make sense at run time can be changed afterwards
// with rc_set_option.  The TUNING / DIAGNOSTIC ones exist only in builds with -DRC_DIAG (tools/, build_exp/): in the product build
// rc_env_diag is a constant null and the code behind it folds away.
static inline const char *rc_env(const char *name) { return getenv(name); }
#ifdef RC_DIAG
static inline const char *rc_env_diag(const char *name) { return getenv(name); }
#else
static inline const char *rc_env_diag(const char *) { return nullptr; }
#endif

#ifdef RC_POISON   // diagnostic builds: every device allocation starts out filled with 0xAB, so that a read of memory that
                   // was never initialised gives the same wrong answer on every machine instead of depending on what the
                   // allocator recycled
static hipError_t rc_poison_malloc(void **p, size_t sz, const char *site)
{
    hipError_t e = (hipMalloc)(p, sz);
    const char *only = getenv("RC_POISON_ONLY");   // poison just the allocation sites whose text contains this string
    if (e == hipSuccess && (!only || strstr(site, only))) {
        e = hipMemset(*p, 0xAB, sz);
        if (e == hipSuccess) e = hipDeviceSynchronize();   // the library's streams are non-blocking: not ordered after the null stream
    }
    return e;
}
#define hipMalloc(p, sz) rc_poison_malloc((void **)(p), (sz), #p)
#endif

typedef long long ll2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

#define RC_KEY_NONE 0xFFFFFFFFFFFFFFFFull
#ifndef RC_RES_THREADS
#define RC_RES_THREADS 512  // k_resolve block beside a row reduction: 32 points x 16 candidate streams; 2 waves/SIMD so it co-resides with k_bulk
#endif
#ifndef RC_RES_THREADS_INC
#define RC_RES_THREADS_INC 1024  // block size in the incremental mode — no row reduction beside the resolver, the CU is its own: four waves per SIMD
                                 // hide the latency of the score arithmetic (f64 logs, dependent chains) twice as well as two: moving regime of
                                 // bench.py 4.55 k -> 4.89 k sweeps/s, first pass 41.7 -> 39.4, validation 33.5 -> 29.5, commit 33.1 -> 28.4 us per sweep
#endif
#define RC_RES_THREADS_MAX (RC_RES_THREADS_INC > RC_RES_THREADS ? RC_RES_THREADS_INC : RC_RES_THREADS)
#ifndef RC_RES_WIDE_MIN_K
#define RC_RES_WIDE_MIN_K 96     // clusters from which the incremental mode takes the wide block
#endif
#ifndef RC_PTS
#define RC_PTS 32           // points per chunk of the resolver (a power of two <= 32): a block's threads are RC_PTS points x (threads / RC_PTS)
                            // candidate streams; chunks are dealt to the blocks cyclically.  Finer chunks would spread the points still
                            // open in a later round of a sweep — a suffix of the point order — over more blocks (at n = 8192 a 32-point
                            // chunk is all a block has), but every chunk pays the pass's fixed costs (block barriers, the reduction
                            // over the streams): measured in the moving regime 2,520 sweeps/s with 32, 2,490 with 16, 2,020 with 8
#endif
#define RC_PTS_LOG2 (RC_PTS == 32 ? 5 : RC_PTS == 16 ? 4 : RC_PTS == 8 ? 3 : RC_PTS == 4 ? 2 : -1)
static_assert(RC_PTS_LOG2 > 0, "RC_PTS must be 4, 8, 16 or 32");
#define RC_MAX_KCAP 4096          // slot capacity up to which the resolver's per-slot tables live in LDS (the fast path)
#define RC_WIDE_MAX_KCAP 32767    // ... and the most a context holds at all: beyond RC_MAX_KCAP the tables live in global memory and the sweep is
                                  // the plain sequential kernel k_sweep_wide (slow; slot ids are 16-bit in the label snapshots and batch records)
#define RC_RES_ONE_STREAM_MAX_N 1024   // up to this size the in-order resolver chain wins (n = 1000: 26 k -> 32 k sweeps/s; n >= 2000: even or worse)
#define RC_USED_LDS_MAX_N 16384   // up to this n the label-occupancy bitset of the resolver lives in LDS (n/8 bytes)
#ifndef RC_BIRTH_MAX
#define RC_BIRTH_MAX 24           // new clusters per resolver batch (a cut, not a limit of the chain: the rest of the batch is announced again).  Only the
                                  // first sweeps from a poor labelling meet it: 24 / 48 / 72 / 96 / 144 births per batch all take 75 rounds for the first sweep
                                  // from uniformly random labels — what ends a round there is the first violation, not the cut — and the simulation of the
                                  // births behind it is wasted: 14.3 / 15.8 / 16.5 / 16.3 / 16.9 ms (round 4; moving regime of bench.py unchanged, 6 births per batch)
#endif
#define RC_MAXB 512          // tentative changers validated per resolve round
#define RC_SPIN_LIMIT (1u << 23)
#if defined(RC_PROF_SYML) || defined(RC_TRACE_RESOLVE)   // profiling / diagnostic builds: records behind the work counter
#define RC_WORK_BYTES (64 + 8192 * 128)
#else
#define RC_WORK_BYTES 64
#endif

// error bits in DevScalars.err
#define RC_DERR_CAPACITY 1
#define RC_DERR_BARRIER 2

struct DevScalars {
    int K;                  // number of non-empty clusters
    int n_changes;          // label changes committed in the last sweep
    int n_rounds;           // scoring passes of the last sweep
    int err;                // RC_DERR_* bits
    int smallest_empty;     // smallest empty label (1-based), n+1 if none   (findfirst(clustsizes .== 0), mcmc.jl:199)
    int slot_hi;            // 1 + highest slot index used since rc_set_state (rows >= slot_hi of every S buffer are zero)
    int last_change_sweep;  // internal index of the last sweep that changed a label (-1: none)
    int runs;               // number of label runs in natural point order (#{i : slot[i] != slot[i-1]} + 1)
    // written by the sweep that ran out of slots (RC_DERR_CAPACITY): the host grows the tables and resumes it (recover_capacity)
    int fail_t;             // internal index of that sweep
    int resume_after;       // every point <= resume_after (caller's order) is final; the next one needs a new slot
    int fail_changes;       // label changes it had committed
    int fail_rounds;        // rounds it had run
};

// Per-sweep summary written by block 0 of k_resolve straight into host-mapped pinned memory: the host needs K and
// the cluster sizes after every sweep for the scalar r / p updates (src/mcmc.jl:84-89,139-144) —
// one stream synchronisation and no copy call.
constexpr int RC_REC_SLOTS = 2;  // sample slots of the asynchronous recorder (rc_run_chain)

struct HostSummary {
    int K, n_changes, n_rounds, err, slot_hi, seq, runs, fail_t;
    int resume_after, fail_changes, fail_rounds, pad1;   // (see DevScalars)
    int size_label[2];      // really [2 * kcap_max] (hsum_bytes): [2k] = size of slot k, [2k+1] = its 1-based label (0 = free)
};
static size_t hsum_bytes(long long kcap_max) { return sizeof(HostSummary) + 2 * (size_t)kcap_max * sizeof(int); }

// Everything a kernel needs, passed by value.
struct View {
    int n, ld, kcap;
    unsigned *wide_scratch;    // wide contexts (kcap > RC_MAX_KCAP): [(n+31)/32] label bitset, then 2·(kcap+1) ints of k_derive_wide / k_sweep_wide; null otherwise
    unsigned *used_scratch;    // [G][(n+31)/32] label bitsets of the resolver blocks when n > RC_USED_LDS_MAX_N
    int cu_cache;              // 1: k_resolve keeps the indices and slots of each block's points in LDS when the block has at most RC_CPB_LDS chunks (rc_set_option "lds_point_cache")
    double *wc;                // [kcap][ldw] score cache of the resolver (eval_chunk), valid inside one launch; null = off
    int ldw;                   // n rounded up to whole chunks
    int wc_always;             // 1: fill the cache in every sweep; 0: only when the previous sweep changed labels
    int maxb;                  // batch capacity of the resolver (<= RC_MAXB; smaller when that makes its LDS fit beside the row reduction)
    const void *Dq48;          // [n][ld] D again, packed to 48 bits per entry (internal order): what k_bulk_syml2's fast path streams; null = none
    const void *Dq, *Lq;       // [n][ld] fixed point: int64 (bits = 64) or int32 (bits = 32); rows/columns in INTERNAL order
    int bits;
    const long long *diagq;    // [n] Dq[i][i]
    int derived;               // 1: logD is not stored; Lq(i,j) = rc_qlog(Dq(i,j)) for i != j, 0 on the diagonal
    double qsD, qsL;           // 2^-eD, 2^eL
    const double2 *ltab;       // [128] table of rc_qlog (see there)
    const double2 *flt;        // [128] table of rc_flog: (1/c_i rounded, -log of that double) — the logarithms of the scores
    int qeD;                   // eD
    long long *SD[3], *SL[3];  // three generations of the [kcap][ld] row-sum table (software pipelining)
    int *slot_of;              // [n] slot of every point, INTERNAL point order
    const int *pi;             // [n] original point index -> internal index (cluster-contiguous layout chosen at rc_set_state)
    int *slot_size;            // [kcap]
    int *slot_label;           // [kcap] 1-based label, 0 = free slot
    short *slot_pos;           // [kcap] rank of the slot's label among active labels
    short *slot_act;           // [kcap] slot_act[pos] = slot
    int *perm[2], *pslot[2];   // [n] rows grouped by slot, and the slot of each sorted position (two generations)
    int *snap[2];              // [n] slot of every point as it was when the generation was written (k_bulk_sym reads these)
    int *work[2];              // work-item counters of k_bulk_sym (two generations)
    const double *A;           // [n+1] size table
    u64 *keys[2];              // [2n+8] one word per resolve round: first violation (batch rounds) (two generations)
    u64 *cword[2];             // [2][nchunks + 1] per round parity and 32-point chunk: (round stamp << 32) | mask of tentative changers (two generations)
    unsigned *rec;             // [2][n] per round parity: (own slot << 16) | (target slot + 1) of a tentative changer (0 target = new cluster)
    unsigned *arrive[2];       // grid-barrier arrival counters (two generations)
    const int4 *ufast, *uslow; // unit lists of k_bulk_syml2 (c0, first row, end row, 0), grouped by wave: the units its fast path takes / the rest
    const int *wfast, *wslow;  // [nwaves + 1] offsets of each wave's units in the two lists (build_syml2_lists: balanced by cost)
    int nfast, nslow;
    DevScalars *sc;
    HostSummary *hsum;         // device address of the host-mapped summary
    double scD, scL;           // 2^-eD, 2^-eL
    double alpha, beta, zeta, gamma, delta1, delta2, cL;
    int repulsion;
    long long maxK;
};

struct SweepArgs {
    double r, logp, log1mp;
    unsigned k0, k1, sw_lo, sw_hi;
    int t;    // internal sweep index since rc_set_state: selects key / perm generations
    int own_gen, next_gen;  // S generation read (and corrected in place); generation being filled for the next sweep (-1: none)
    int zero_gen;           // S generation this launch clears for the row reduction two sweeps ahead (-1: none)
    int dbg;  // timing ablations only (RC_DEBUG_FLAGS / rc_set_option "debug_flags", -DRC_DIAG builds): 1 = skip candidate loop, 2 = skip grid barrier, 4 = skip gumbel, 16 = no score-cache stores
    // a sweep resumed after the slot tables were grown (recover_capacity): the points <= after0 are final already, changes0 /
    // rounds0 are what the first part of the sweep had committed / run.  A fresh sweep: -1, 0, 0.
    int after0, changes0, rounds0;
    int prune;   // candidates that cannot win skip their noise ("Pruned candidates", eval_chunk): exact either way; the host turns it on while few labels move
};

// ---------------------------------------------------------------------------------------------------
// Uniform stream: Philox4x32-10, counter (pos, i, sweep_lo, sweep_hi), key = seed.  (include/redclust_hip.h)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double rc_uniform(const SweepArgs &a, unsigned i, unsigned pos)
{
    unsigned c0 = pos, c1 = i, c2 = a.sw_lo, c3 = a.sw_hi, k0 = a.k0, k1 = a.k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const u64 p0 = (u64)0xD2511F53u * (u64)c0, p1 = (u64)0xCD9E8D57u * (u64)c2;   // (one 32 x 32 -> 64 multiply each)
        const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const u64 bits = (((u64)c0 << 32) | c1) >> 12;
    return ((double)bits + 0.5) * 0x1p-52;
}

// ---------------------------------------------------------------------------------------------------
// Staging (MCMCData, src/types.jl:145-157)
// ---------------------------------------------------------------------------------------------------
// flags: bit0 asymmetric, bit1 non-finite, bit2 non-positive off-diagonal entry
__global__ void k_check(const double *__restrict__ D, int n, unsigned *flags, u64 *maxabs_bits)
{
    const size_t total = (size_t)n * n;
    unsigned f = 0;
    u64 m = 0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(t / n), j = (int)(t % n);
        const double x = D[t];
        if (x != D[(size_t)j * n + i]) f |= 1u;
        if (!(fabs(x) <= 1.79769313486231570e308)) f |= 2u;
        if (i != j && !(x > 0.0)) f |= 4u;
        const u64 b = (u64)__double_as_longlong(fabs(x));
        m = b > m ? b : m;
    }
    if (f) atomicOr(flags, f);
    atomicMax(maxabs_bits, m);
}

__global__ void k_make_log(const double *__restrict__ D, double *__restrict__ L, int n)
{
    const size_t total = (size_t)n * n;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(t / n), j = (int)(t % n);
        L[t] = (i == j) ? 0.0 : log(D[t]);  // types.jl:155: log.(D - Diagonal(D) + I)
    }
}

// flags: bit0 asymmetric (checked when n > 0: X is n×n), bit1 non-finite
__global__ void k_maxabs(const double *__restrict__ X, size_t total, unsigned *flags, u64 *maxabs_bits, int n)
{
    u64 m = 0;
    unsigned f = 0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        if (n > 0 && X[t] != X[(t % (size_t)n) * (size_t)n + t / (size_t)n]) f |= 1u;
        const double x = fabs(X[t]);
        if (!(x <= 1.79769313486231570e308)) f |= 2u;
        const u64 b = (u64)__double_as_longlong(x);
        m = b > m ? b : m;
    }
    if (f) atomicOr(flags, f);
    atomicMax(maxabs_bits, m);
}

// Pairwise Euclidean distances of n points in R^dim (MCMCData(points), src/types.jl:159-162:
// pairwise(Euclidean(), makematrix(pnts), dims=2); Distances.jl evaluates sqrt(max(|a_i|² + |a_j|² − 2 a_i·a_j, 0)),
// mirrors the triangle and zeroes the diagonal).  pts is n×dim row-major (= the dim×N column-major matrix of
// makematrix).  64×64 pairs per block, 4×4 per thread, dim tiled by 16 through LDS; the dot product runs in
// ascending coordinate order for both (i,j) and (j,i), so the result is exactly symmetric.
__global__ __launch_bounds__(256) void k_pairwise(const double *__restrict__ pts, int n, int dim, double *__restrict__ D)
{
    const int tj = blockIdx.x, ti = blockIdx.y;
    if (ti < tj) return;  // lower block triangle (incl. diagonal blocks); mirrored on store
    __shared__ double A[16][65], B[16][65];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    double acc[4][4], sa[4], sb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { sa[u] = 0; sb[u] = 0;
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0; }
    for (int k0 = 0; k0 < dim; k0 += 16) {
        for (int q = threadIdx.x; q < 16 * 64; q += 256) {
            const int kk = q & 15, r = q >> 4;
            const int gi = ti * 64 + r, gj = tj * 64 + r, k = k0 + kk;
            A[kk][r] = (gi < n && k < dim) ? pts[(size_t)gi * dim + k] : 0.0;
            B[kk][r] = (gj < n && k < dim) ? pts[(size_t)gj * dim + k] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = A[kk][ty * 4 + u]; b[u] = B[kk][tx * 4 + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                sa[u] += a[u] * a[u];
                sb[u] += b[u] * b[u];
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] += a[u] * b[v];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = ti * 64 + ty * 4 + u, j = tj * 64 + tx * 4 + v;
            if (i < n && j < n) {
                const double d2 = sa[u] + sb[v] - 2.0 * acc[u][v];
                const double d = (i == j) ? 0.0 : sqrt(d2 > 0.0 ? d2 : 0.0);
                D[(size_t)i * n + j] = d;
                if (ti != tj) D[(size_t)j * n + i] = d;
            }
        }
}

// ---------------------------------------------------------------------------------------------------
// Derived logD.  When the caller gives only D (MCMCData computes logD = log.(D − Diagonal(D) + I) itself,
// types.jl:155), the fixed-point logD need not be stored at all: Lq(i,j) = rc_qlog(Dq(i,j)) ≈ rint(log(Dq(i,j)·2^-eD)·2^eL) is a
// pure function of Dq(i,j), evaluated by every consumer with this one routine, so the row reduction reads HALF the bytes.
// What matters for exactness is only that every consumer uses this same function (all sums are sums of these integers); what
// matters for parity is that it is log to well below the 1.2e-10 the derived mode allows itself (tests/test_gpu_headline.py
// checks it against libm at N = 8192).
//   x = dq·2^-eD = 2^k·m, m ∈ [1,2);  j = top 7 fraction bits of m;  c_j = 1 + (j+½)/128;  r = m/c_j − 1 ∈ [−1/257, 1/257]
//   log x = k·ln2 + log c_j + log1p(r),   log1p(r) = r + r²(−1/2 + r(1/3 − r/4))   (next term r⁵/5 < 1.8e-13)
//   Lq = k·LN2S + T_j + 2^eL·log1p(r), rounded to the integer by accumulating on the "magic" 1.5·2^52:
//        v = fma((double)k, LN2S, T_j), T_j = rint(log c_j·2^eL) + 1.5·2^52 (integer-valued, per context);  w = fma(log1p(r), 2^eL, v);
//        Lq = bits(w) − bits(1.5·2^52)         (|Lq| < 2^51 by the choice of eL, so v and w stay in [2^52, 2^53): unit spacing)
// 16 VALU instructions per entry — six of them double-precision FMAs / multiplies — and one 16-byte table read (round 2's form of
// the same idea took 23: a degree-6 polynomial, the scaling and the rounding as separate steps); the libm log is ~100.
// ltab[j] = (2/c_j, T_j): 128 entries, rebuilt per context once eL is known (create_impl).
// ---------------------------------------------------------------------------------------------------
// Front end: the fixed-point entry is an integer below 2^52 (create_impl caps eD accordingly in the derived mode), so
// OR-ing it into the mantissa of 2^52 and subtracting 2^52 converts it to a double exactly in two instructions; exponent,
// table index and mantissa then come from the HIGH dword of that double with 32-bit operations (no count-leading-zeros,
// no 64-bit shifts).  dq = 0 (padding, masked entries) gives a finite value the callers discard.
// (round 4: the mantissa is one v_frexp_mant_f64 — m/2 in [0.5, 1), the table holds 2/c_j: the same product, the same r, bit for
// bit — instead of and + or + a register copy; the table offset is taken in bytes (shift + mask instead of shift + mask + shift))
struct QlogPrep { int j; unsigned joff; double kd, m; };
__device__ __forceinline__ QlogPrep rc_qlog_prep(long long dq, int eD)
{
    const double x = __longlong_as_double(dq | 0x4330000000000000ll) - 0x1p52;
    const unsigned hi = (unsigned)__double2hiint(x);
    int kbias;                                                              // 1023 + eD, opaque to the optimiser (which splits the sum into two vector subtractions per entry otherwise)
    asm("s_add_i32 %0, %1, 0x3ff" : "=s"(kbias) : "s"(eD) : "scc");
    QlogPrep P;
    P.kd = (double)((int)(hi >> 20) - kbias);                               // x·2^-eD = m·2^k, m in [1,2)
    P.j = (int)((hi >> 13) & 127u);                                         // top 7 fraction bits
    P.joff = (hi >> 9) & 0x7f0u;                                            // 16 j: byte offset of entry j in a plain table
    P.m = __builtin_amdgcn_frexp_mant(x);                                   // m / 2
    return P;
}
__device__ __forceinline__ double2 rc_qlog_entry(const double2 *__restrict__ tab, const QlogPrep &P)
{
    return *(const double2 *)((const char *)tab + P.joff);
}
// the value for dq > 0 — no select on dq: callers that may hold dq <= 0 mask the result
__device__ __forceinline__ long long rc_qlog_raw(const QlogPrep &P, double2 t, double sL, double third = 1.0 / 3, double mquarter = -1.0 / 4)
{
    const double r = fma(P.m, t.x, -1.0);
    double p = fma(r, mquarter, third);   // (1/3 and -1/4: the streaming kernel hands them over in a vector / a scalar register pair, see rc_third_vgpr)
    p = fma(r, p, -1.0 / 2);
    const double q = fma(r * r, p, r);                                      // log1p(r)
    const double v = fma(P.kd, 0.69314718055994530942 * sL, t.y);           // (sL is a power of two: the product is ln2 scaled exactly)
    const double w = fma(q, sL, v);
    return __double_as_longlong(w) - __double_as_longlong(0x1.8p52);
}
// 1/3 in a vector register pair, opaque to the optimiser: fma(r, -1/4, 1/3) is then ONE v_fma_f64 (register, scalar constant, register).
// With both constants scalar the instruction may hold only one of them, and the compiler copies 1/3 into the accumulator of a v_fmac
// first: one more vector instruction per entry of the matrix.
__device__ __forceinline__ double rc_mquarter_sgpr()
{
    int hi;
    asm volatile("s_mov_b32 %0, 0xbfd00000" : "=s"(hi));            // -1/4, opaque: as a literal it exists only in the two-operand v_fmac form
    return __hiloint2double(hi, 0);
}
__device__ __forceinline__ double rc_third_vgpr()
{
    unsigned lo, hi;
    asm volatile("v_mov_b32 %0, 0x55555555\n\tv_mov_b32 %1, 0x3fd55555" : "=v"(lo), "=v"(hi));
    return __hiloint2double((int)hi, (int)lo);
}
__device__ __forceinline__ long long rc_qlog_finish(long long dq, const QlogPrep &P, double2 t, double sL)
{
    return dq > 0 ? rc_qlog_raw(P, t, sL) : 0ll;                            // padding and masked entries carry dq = 0
}
__device__ __forceinline__ long long rc_qlog(long long dq, int eD, double sL, const double2 *__restrict__ tab)
{
    const QlogPrep P = rc_qlog_prep(dq, eD);
    return rc_qlog_finish(dq, P, rc_qlog_entry(tab, P), sL);
}

// ---------------------------------------------------------------------------------------------------
// The logarithms of the candidate scores (round 4).  A candidate of k_resolve needs two log1p (cohesion and repulsion terms,
// mcmc.jl:223-241) and the Gumbel noise -log(-log u) (utils.jl:4): with the device library's double-precision routines these are
// 135 + 135 + 170 of its ~570 VALU instructions — the library works in double-double to stay below one ulp for EVERY argument,
// special values included.  Here every argument is a positive normal number (u in (0,1), its negative log, 1 + a non-negative
// ratio), so one table-driven routine serves all of them in ~25 instructions at the same ~1 ulp:
//   x = 2^k z, z in [0.6875, 1.375) (the exponent is split at 0.6875 so that x near 1 has k = 0: no cancellation against k ln2);
//   i = top 7 mantissa bits of z's offset: 128 intervals; c_i = its centre — but c_i = 1 in the two intervals that touch 1.0, where
//   r = z - 1 is exact and the result keeps its RELATIVE accuracy (what -log u needs when u is close to 1: the large noise values);
//   r = fma(z, invc_i, -1) with invc_i = 1/c_i rounded, logc_i = -log(invc_i) to 64 bits rounded once (built on the host in long
//   double: log z = log1p(r) + logc_i holds for the STORED reciprocal, whatever its rounding); |r| <= 2^-7;
//   log1p(r) = r + r^2 (-1/2 + r/3 - ... - r^6/8)   (next term r^9/9 <= 1.2e-20);
//   result = (k ln2_hi + logc_i + r) [two-sum] + (k ln2_lo + r^2 p + extra).
// log1p(x) = log(u) + c/u with u + c = 1 + x exactly (two-sum): c/u <= 2^-53, through `extra`.
// (the method of the table-driven libm logarithms; tables and code are this library's own)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double rc_flog(double x, const double2 *__restrict__ tab, double extra = 0.0)
{
    const int hi = __double2hiint(x);
    const int tmp = hi - 0x3fe60000;
    const int k = tmp >> 20;                                                // (arithmetic: x < 0.6875 has k < 0)
    const double z = __hiloint2double(hi - (tmp & (int)0xfff00000u), __double2loint(x));
    const double2 t = tab[(tmp >> 13) & 127];
    const double r = fma(z, t.x, -1.0);
    const double kd = (double)k;
    const double h = fma(kd, 0x1.62e42fefa3800p-1, t.y);                    // k ln2_hi is exact (ln2_hi has 42 significant bits)
    const double t1 = h + r;
    const double lo = fma(kd, 0x1.ef35793c76730p-45, (h - t1) + r) + extra; // (|h| >= |r| or h = 0: the two-sum is exact)
    double p = fma(r, -1.0 / 8, 1.0 / 7);
    p = fma(r, p, -1.0 / 6);
    p = fma(r, p, 1.0 / 5);
    p = fma(r, p, -1.0 / 4);
    p = fma(r, p, 1.0 / 3);
    p = fma(r, p, -0.5);
    return fma(r * r, p, lo) + t1;
}
__device__ __forceinline__ double rc_flog1p(double x, const double2 *__restrict__ tab)      // x >= 0
{
    const double u = 1.0 + x, v = u - 1.0;
    const double c = (1.0 - (u - v)) + (x - v);
    return rc_flog(u, tab, c * __builtin_amdgcn_rcp(u));
}
__device__ __forceinline__ double rc_gumbel(double un, const double2 *__restrict__ tab)     // -log(-log u), u in (0, 1)
{
    return -rc_flog(-rc_flog(un, tab), tab);
}

// logD entry (row, col) in internal order, stored or derived; xd = Dq(row, col) when the caller has it already
__device__ __forceinline__ long long rc_load_L(const View &V, int row, int col, long long xd)
{
    if (V.derived && !V.Lq) return row == col ? 0ll : rc_qlog(xd, V.qeD, V.qsL, V.ltab);
    const size_t e = (size_t)row * V.ld + col;
    return (V.bits == 64) ? ((const long long *)V.Lq)[e] : (long long)((const int *)V.Lq)[e];
}

// one-off scan for the derived mode: smallest off-diagonal Dq (must be > 0) and largest |log| (fixes eL)
__global__ void k_derived_scan(const long long *__restrict__ Dq, int n, int ld, int eD, long long *min_dq, u64 *maxabs_bits)
{
    long long mn = 0x7fffffffffffffffll;
    double mx = 0.0;
    const size_t total = (size_t)n * n;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t i = t / n, j = t % n;
        if (i == j) continue;
        const long long dq = Dq[i * ld + j];
        mn = min(mn, dq);
        if (dq > 0) mx = fmax(mx, fabs(log(ldexp((double)dq, -eD))));   // (one-off; the table of rc_qlog needs eL, which this fixes)
    }
    atomicMin((long long *)min_dq, mn);
    atomicMax((unsigned long long *)maxabs_bits, (unsigned long long)__double_as_longlong(mx));
}

// The derived values materialised once (same routine, so the same integers): the resolver's single-entry look-ups and
// k_apply_moves read this copy, which costs them one load instead of a dependent load + ~30 instructions per entry;
// the streaming row reduction keeps deriving on the fly and never touches it.
__global__ void k_derived_fill(const long long *__restrict__ Dq, int n, int ld, int eD, double sL,
                               const double2 *__restrict__ tab, long long *__restrict__ Lq)
{
    const size_t total = (size_t)n * n;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t i = t / n, j = t % n;
        Lq[i * ld + j] = (i == j) ? 0ll : rc_qlog(Dq[i * ld + j], eD, sL, tab);
    }
}

// dequantised logD for rc_get_matrix in the derived mode (caller's order: Dq_src)
__global__ void k_derived_matrix(const long long *__restrict__ Dq, int n, int ld, int eD, double sL, double scale,
                                 const double2 *__restrict__ tab, double *__restrict__ out)
{
    const size_t total = (size_t)n * n;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t i = t / n, j = t % n;
        out[t] = (i == j) ? 0.0 : (double)rc_qlog(Dq[i * ld + j], eD, sL, tab) * scale;
    }
}

// rows rows[0..nrows) of the device matrix (caller's order) as doubles, value = q·2^-e; derived logD through rc_qlog
template <typename T>
__global__ __launch_bounds__(256) void k_get_rows(const T *__restrict__ Q, const int *__restrict__ rows, int n, int ld, double scale,
                                                  int derive, int eD, double sL, const double2 *__restrict__ tab, double *__restrict__ out)
{
    const int r = rows[blockIdx.y];
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const long long q = (long long)Q[(size_t)r * ld + j];
        out[(size_t)blockIdx.y * n + j] = derive ? ((j == r) ? 0.0 : (double)rc_qlog(q, eD, sL, tab) * scale) : (double)q * scale;
    }
}

// Σ_j X[i][j] of every row of D and of logD in the CALLER's point order (the *_src copies; derived logD through rc_qlog, 0 on the
// diagonal): a plain one-block-per-row sum that shares nothing with the row-reduction kernels (not their layout, not their packed
// copy, not their unit lists) — the tests hold Σ_k S[k][i] against it at sizes where the checker cannot hold the matrix (config 5).
template <typename T>
__global__ __launch_bounds__(256) void k_rowtotals(const T *__restrict__ Dq, const T *__restrict__ Lq_or_null, int n, int ld, int derive,
                                                   int eD, double sL, const double2 *__restrict__ tab, long long *__restrict__ outD,
                                                   long long *__restrict__ outL)
{
    __shared__ long long red[2][4];
    const int i = blockIdx.x;
    long long sd = 0, sl = 0;
    for (int j = threadIdx.x; j < n; j += 256) {
        const long long q = (long long)Dq[(size_t)i * ld + j];
        sd += q;
        if (Lq_or_null) sl += (long long)Lq_or_null[(size_t)i * ld + j];
        else if (derive && j != i) sl += rc_qlog(q, eD, sL, tab);
    }
    for (int off = 32; off > 0; off >>= 1) { sd += __shfl_down(sd, off); sl += __shfl_down(sl, off); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sd; red[1][threadIdx.x >> 6] = sl; }
    __syncthreads();
    if (threadIdx.x == 0) {
        outD[i] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        outL[i] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_relayout(const T *__restrict__ src, const int *__restrict__ ipi, int n, int ld, T *__restrict__ out)
{
    const int w = blockIdx.y;
    const T *row = src + (size_t)ipi[w] * ld;
    for (int x = blockIdx.x * 256 + threadIdx.x; x < n; x += gridDim.x * 256) out[(size_t)w * ld + x] = row[ipi[x]];
}

__global__ void k_gather_ll(const long long *__restrict__ src, const int *__restrict__ ipi, int n, long long *__restrict__ out)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < n) out[w] = src[ipi[w]];
}

// D once more, 48 bits per entry (two entries in three dwords), for the streaming row reduction: 6 instead of 8 bytes per entry.
// The derived mode caps eD so that every entry is below 2^47 (create_impl); row pitch ld entries, as Dq.
__global__ __launch_bounds__(256) void k_pack48(const long long *__restrict__ Dq, size_t pairs, unsigned *__restrict__ out)
{
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < pairs; t += (size_t)gridDim.x * blockDim.x) {
        const ll2 v = *(const ll2 *)(Dq + 2 * t);
        const u64 a = (u64)v.x, b = (u64)v.y;
        out[3 * t] = (unsigned)a;                                               // the low words as they are, the two 16-bit tops share the middle word
        out[3 * t + 1] = (unsigned)(a >> 32) | ((unsigned)(b >> 32) << 16);
        out[3 * t + 2] = (unsigned)b;
    }
}

// fixed-point matrix back to doubles (value = q·2^-e), for checks
template <typename T>
__global__ void k_dequantize(const T *__restrict__ Q, int n, int ld, double scale, double *__restrict__ out)
{
    const size_t total = (size_t)n * n;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t i = t / n, j = t % n;
        out[t] = (double)Q[i * ld + j] * scale;
    }
}

template <typename T>
__global__ void k_quantize(const double *__restrict__ X, int n, int ld, int e, T *__restrict__ Q,
                           long long *__restrict__ diag_or_null)
{
    const size_t total = (size_t)n * n;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(t / n), j = (int)(t % n);
        const long long q = __double2ll_rn(scalbn(X[t], e));
        Q[(size_t)i * ld + j] = (T)q;
        if (diag_or_null && i == j) diag_or_null[i] = q;
    }
}

// ---------------------------------------------------------------------------------------------------
// perm / pslot: rows grouped by slot (order inside a group is irrelevant — sums are exact integers).
// One block.  LDS: 2*kcap ints.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void build_perm_block(const View &V, int gen, int *lds_off /*kcap*/, int *lds_cur /*kcap*/)
{
    int *perm = V.perm[gen], *pslot = V.pslot[gen];
    for (int k = threadIdx.x; k < V.kcap; k += blockDim.x) lds_cur[k] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < V.n; i += blockDim.x) atomicAdd(&lds_cur[V.slot_of[i]], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int o = 0;
        for (int k = 0; k < V.kcap; ++k) { lds_off[k] = o; o += lds_cur[k]; }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < V.kcap; k += blockDim.x) lds_cur[k] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < V.n; i += blockDim.x) {
        const int s = V.slot_of[i];
        const int p = lds_off[s] + atomicAdd(&lds_cur[s], 1);
        perm[p] = i;
        pslot[p] = s;
    }
    __syncthreads();
}

// The same order built by all G blocks of the resolver at the end of a sweep: the slot offsets are the running sums of the
// slot sizes (every block holds them), block b places the points of its share of the slots — one pass over slot_of per block,
// positions inside a slot from LDS counters (the order inside a cluster is arbitrary here as above).  lds: 2·kcap + 1 ints.
__device__ __forceinline__ void build_perm_grid(const View &V, int gen, const int *size /*LDS, [hi]*/, int hi, int G, int *lds)
{
    int *off = lds, *cur = lds + V.kcap + 1;
    for (int k = threadIdx.x; k <= V.kcap; k += blockDim.x) { off[k] = (k < hi) ? size[k] : 0; if (k < V.kcap) cur[k] = 0; }
    __syncthreads();
    if (threadIdx.x < 64) {   // exclusive offsets (wave 0: per-lane runs + shuffle scan)
        const int per = (hi + 63) / 64, c0 = (int)threadIdx.x * per, c1 = min(c0 + per, hi);
        int sum = 0;
        for (int k = c0; k < c1; ++k) sum += off[k];
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if ((int)threadIdx.x >= d) incl += up;
        }
        int o = incl - sum;
        for (int k = c0; k < c1; ++k) { const int x = off[k]; off[k] = o; o += x; }
    }
    __syncthreads();
    const int k0 = (int)((long long)hi * blockIdx.x / G), k1 = (int)((long long)hi * (blockIdx.x + 1) / G);
    if (k0 == k1) return;
    int *perm = V.perm[gen], *pslot = V.pslot[gen];
    for (int i = threadIdx.x; i < V.n; i += blockDim.x) {
        const int s = V.slot_of[i];
        if (s >= k0 && s < k1) {
            const int p = off[s] + atomicAdd(&cur[s], 1);
            perm[p] = i;
            pslot[p] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// k_bulk — THE HBM-BOUND KERNEL.  Row-bucket reduction of D and logD into S (matsum(D,[i],clust_k) and
// matsum(logD,[i],clust_k), src/mcmc.jl:210-213, for all i and k).  Algorithmic traffic 2·n²·8 bytes.
//   block = 256 threads = 512 consecutive columns i (2 per lane, 16-byte non-temporal loads, 4 KiB contiguous
//   per row and matrix); blockIdx.y = split of the cluster-sorted row list; rows of one cluster accumulate in
//   registers and are flushed with exact 64-bit integer atomics when the cluster changes.
//   Prologue: clears the S generation that the sweep after next will fill (nobody reads it any more).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bulk_flush(long long *SD, long long *SL, size_t ld, int slot, int i, long long d0,
                                           long long d1, long long l0, long long l1)
{
    u64 *pd = (u64 *)(SD + (size_t)slot * ld + i);
    u64 *pl = (u64 *)(SL + (size_t)slot * ld + i);
    __hip_atomic_fetch_add(pd, (u64)d0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(pd + 1, (u64)d1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(pl, (u64)l0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(pl + 1, (u64)l1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define RC_BULK_U 8
// 16-byte row segment per lane: two int64 columns or four int32 columns
template <typename T> struct Seg;
template <> struct Seg<long long> { typedef ll2 vec; static constexpr int C = 2; };
template <> struct Seg<int> { typedef int vec __attribute__((ext_vector_type(4))); static constexpr int C = 4; };

template <typename T, bool DERIVED = false>
__global__ __launch_bounds__(256) void k_bulk(View V, int rows_per_split, int wgen, int zgen, int pgen)
{
    typedef typename Seg<T>::vec vec;
    constexpr int C = Seg<T>::C;
    const size_t ld = (size_t)V.ld;
    __shared__ double2 ltab_sh[DERIVED ? 128 : 1];
    if (DERIVED) { if (threadIdx.x < 128) ltab_sh[threadIdx.x] = V.ltab[threadIdx.x]; __syncthreads(); }
    const int qeD = V.qeD;
    const double qsL = V.qsL;
    // derived logD of the C entries of row j at this thread's columns (diagonal and padding -> 0)
    auto derive = [&](const vec &d, int j, int col0) {
        vec l;
#pragma unroll
        for (int q = 0; q < C; ++q) l[q] = (j == col0 + q) ? 0 : (T)rc_qlog((long long)d[q], qeD, qsL, ltab_sh);
        return l;
    };
    (void)zgen;  // the generation for the next sweep is cleared by k_resolve (SweepArgs.zero_gen)
    const int i = (blockIdx.x * 256 + threadIdx.x) * C;
    const int p0 = blockIdx.y * rows_per_split;
    const int p1 = min(V.n, p0 + rows_per_split);
    if (p0 >= p1) return;
    const int *__restrict__ perm = V.perm[pgen];
    const int *__restrict__ pslot = V.pslot[pgen];
    const T *__restrict__ Dq = (const T *)V.Dq + i;
    const T *__restrict__ Lq = (const T *)V.Lq + i;
    long long *SD = V.SD[wgen], *SL = V.SL[wgen];
    long long aD[C], aL[C];
#pragma unroll
    for (int q = 0; q < C; ++q) { aD[q] = 0; aL[q] = 0; }
    auto flush = [&](int slot) {
#pragma unroll
        for (int q = 0; q < C; q += 2) bulk_flush(SD, SL, ld, slot, i + q, aD[q], aD[q + 1], aL[q], aL[q + 1]);
#pragma unroll
        for (int q = 0; q < C; ++q) { aD[q] = 0; aL[q] = 0; }
    };
    int cur = pslot[p0];
    int p = p0;
    for (; p + RC_BULK_U <= p1; p += RC_BULK_U) {
        int j[RC_BULK_U], s[RC_BULK_U];
        vec d[RC_BULK_U], l[RC_BULK_U];
#pragma unroll
        for (int u = 0; u < RC_BULK_U; ++u) { j[u] = perm[p + u]; s[u] = pslot[p + u]; }
#pragma unroll
        for (int u = 0; u < RC_BULK_U; ++u) {
            d[u] = __builtin_nontemporal_load((const vec *)(Dq + (size_t)j[u] * ld));  // streamed once per sweep
            if (!DERIVED) l[u] = __builtin_nontemporal_load((const vec *)(Lq + (size_t)j[u] * ld));
        }
        if (DERIVED) {
#pragma unroll
            for (int u = 0; u < RC_BULK_U; ++u) l[u] = derive(d[u], j[u], i);
        }
#pragma unroll
        for (int u = 0; u < RC_BULK_U; ++u) {
            if (s[u] != cur) { flush(cur); cur = s[u]; }
#pragma unroll
            for (int q = 0; q < C; ++q) { aD[q] += (long long)d[u][q]; aL[q] += (long long)l[u][q]; }
        }
    }
    for (; p < p1; ++p) {
        const int j = perm[p], sl = pslot[p];
        const vec d = __builtin_nontemporal_load((const vec *)(Dq + (size_t)j * ld));
        const vec l = DERIVED ? derive(d, j, i) : __builtin_nontemporal_load((const vec *)(Lq + (size_t)j * ld));
        if (sl != cur) { flush(cur); cur = sl; }
#pragma unroll
        for (int q = 0; q < C; ++q) { aD[q] += (long long)d[q]; aL[q] += (long long)l[q]; }
    }
    flush(cur);
}


// ---------------------------------------------------------------------------------------------------
// k_bulk_sym — symmetric row-bucket reduction (64-bit storage): reads only the upper triangle of D and logD.
// Every entry x = X[a][b], a < b, contributes to S[slot_a][b] (direction 1) and to S[slot_b][a] (direction 2); the
// diagonal goes to S[slot_a][a].  Algorithmic traffic n²·8 B (both matrices) instead of 2·n²·8 B.
//   A 32-row × 128-column tile of each matrix (strictly-upper entries, others zeroed) is staged in LDS.
//   Waves 0-1: one column per lane, accumulate DOWN the rows in registers carried across the tiles of a work item;
//   waves 2-3: one (row, matrix) per lane, accumulate ALONG 64 columns.  Rows/columns are taken 8 at a time; an
//   8-chunk whose points share a slot is summed without looking at labels (fast path), a mixed chunk element by
//   element.  Flushes are coalesced 64-bit integer atomics, so the result is exact for ANY labelling; the kernel is
//   merely fastest when points of a cluster are contiguous (few label runs) — the host picks k_bulk otherwise.
//   Persistent blocks pull work items (column block J, range of row tiles) from an atomic counter, heavy column
//   blocks first.  Labels come from the snapshot generation written two sweeps ago (see software pipelining).
// ---------------------------------------------------------------------------------------------------
#define RC_SYM_TR 32
#define RC_SYM_TC 128
#define RC_SYM_TP (RC_SYM_TC + 2)  // row pitch in elements: row threads hit distinct LDS banks
template <bool DERIVED>
__global__ __launch_bounds__(256) void k_bulk_sym(View V, int wgen, int zgen, int sgen, int cgen, int item_tiles, int nitems)
{
    __shared__ __attribute__((aligned(16))) long long tt[2][RC_SYM_TR][RC_SYM_TP];  // [matrix][row][col]
    __shared__ double2 ltab_sh[DERIVED ? 128 : 1];
    if (DERIVED && threadIdx.x < 128) ltab_sh[threadIdx.x] = V.ltab[threadIdx.x];   // visible after the first barrier of the work loop
    const int qeD = V.qeD;
    const double qsL = V.qsL;
    __shared__ int item_sh, J_sh;
    __shared__ int cslot[RC_SYM_TC], rslot[RC_SYM_TR];
    __shared__ int cchk[RC_SYM_TC / 8], rchk[RC_SYM_TR / 8];  // slot of an 8-wide chunk if uniform, else -2
    const int tid = threadIdx.x;
    const size_t ld = (size_t)V.ld;
    (void)zgen;  // generations are cleared and work counters re-armed by k_resolve (SweepArgs.zero_gen)
    const long long *__restrict__ Dq = (const long long *)V.Dq;
    const long long *__restrict__ Lq = (const long long *)V.Lq;
    const int *__restrict__ slot = V.snap[sgen];
    long long *SD = V.SD[wgen], *SL = V.SL[wgen];
    const int n = V.n;
    const int ncb = (n + RC_SYM_TC - 1) / RC_SYM_TC;
    int *counter = V.work[cgen];
    for (;;) {
        __syncthreads();
        if (tid == 0) {
            int item = atomicAdd(counter, 1);
            int J = -1;
            if (item < nitems) {
                for (J = ncb - 1;; --J) {
                    const int ntile = (min(RC_SYM_TC * J + RC_SYM_TC, n) + RC_SYM_TR - 1) / RC_SYM_TR;
                    const int cnt = (ntile + item_tiles - 1) / item_tiles;
                    if (item < cnt) break;
                    item -= cnt;
                }
            }
            item_sh = item; J_sh = J;
        }
        __syncthreads();
        const int J = J_sh, item = item_sh;
        if (J < 0) break;
        const int c0 = J * RC_SYM_TC;
        const int ntile = (min(c0 + RC_SYM_TC, n) + RC_SYM_TR - 1) / RC_SYM_TR;   // rows a >= c0+128 have no column b > a here
        const int t_begin = item * item_tiles, t_end = min(ntile, t_begin + item_tiles);
        if (tid < RC_SYM_TC) cslot[tid] = (c0 + tid < n) ? slot[c0 + tid] : -1;
        __syncthreads();
        if (tid < RC_SYM_TC / 8) {
            const int s0 = cslot[tid * 8];
            bool u = true;
            for (int q = 1; q < 8; ++q) u = u && (cslot[tid * 8 + q] == s0);
            cchk[tid] = u ? s0 : -2;
        }
        // loader role: 32 rows x 64 sixteen-byte pieces per matrix = 8 pieces per thread and matrix
        ll2 d[8], l[8];
        auto issue = [&](int t) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int piece = q * 256 + tid, lr = piece >> 6, lp = piece & 63;
                const int r = min(t * RC_SYM_TR + lr, n - 1);
                d[q] = __builtin_nontemporal_load((const ll2 *)(Dq + (size_t)r * ld + c0 + lp * 2));
                if (!DERIVED) l[q] = __builtin_nontemporal_load((const ll2 *)(Lq + (size_t)r * ld + c0 + lp * 2));
            }
        };
        // derived logD: only D is loaded, so the registers of the logD pieces hold a second D tile — two tiles in flight
        auto issue_buf = [&](ll2 (&buf)[8], int t) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int piece = q * 256 + tid, lr = piece >> 6, lp = piece & 63;
                const int r = min(t * RC_SYM_TR + lr, n - 1);
                buf[q] = __builtin_nontemporal_load((const ll2 *)(Dq + (size_t)r * ld + c0 + lp * 2));
            }
        };
        if (DERIVED) {
            issue_buf(d, t_begin);
            if (t_begin + 1 < t_end) issue_buf(l, t_begin + 1);
        } else {
            issue(t_begin);
        }
        long long accD = 0, accL = 0;
        int cur = -1;
        for (int t = t_begin; t < t_end; ++t) {
            const int r0 = t * RC_SYM_TR;
            __syncthreads();  // the previous tile has been consumed
            if (tid < RC_SYM_TR) rslot[tid] = (r0 + tid < n) ? slot[r0 + tid] : -1;
            if (!DERIVED) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int piece = q * 256 + tid, lr = piece >> 6, lp = piece & 63;
                    const int r = r0 + lr, b = c0 + lp * 2;
                    const bool live = r < n;
                    ll2 x = d[q], y = l[q];
                    if (!(live && b > r)) { x.x = 0; y.x = 0; }
                    if (!(live && b + 1 > r)) { x.y = 0; y.y = 0; }
                    *(ll2 *)&tt[0][lr][lp * 2] = x;
                    *(ll2 *)&tt[1][lr][lp * 2] = y;
                }
            } else {
                // derived logD: free the load registers first, so that the loads of tile t+2 fly under the logs
                auto stage = [&](ll2 (&buf)[8]) {
                    ll2 x[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int piece = q * 256 + tid, lr = piece >> 6, lp = piece & 63;
                        const int r = r0 + lr, b = c0 + lp * 2;
                        const bool live = r < n;
                        x[q] = buf[q];
                        if (!(live && b > r)) x[q].x = 0;      // masked entries and the zero padding beyond column n give 0
                        if (!(live && b + 1 > r)) x[q].y = 0;
                    }
                    if (t + 2 < t_end) issue_buf(buf, t + 2);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int piece = q * 256 + tid, lr = piece >> 6, lp = piece & 63;
                        ll2 y;
                        y.x = rc_qlog(x[q].x, qeD, qsL, ltab_sh);
                        y.y = rc_qlog(x[q].y, qeD, qsL, ltab_sh);
                        *(ll2 *)&tt[0][lr][lp * 2] = x[q];
                        *(ll2 *)&tt[1][lr][lp * 2] = y;
                    }
                };
                if ((t - t_begin) & 1) stage(l); else stage(d);
            }
            if (!DERIVED && t + 1 < t_end) issue(t + 1);  // in flight while this tile is reduced
            __syncthreads();
            if (tid < RC_SYM_TR / 8) {
                const int s0 = rslot[tid * 8];
                bool u = true;
                for (int q = 1; q < 8; ++q) u = u && (rslot[tid * 8 + q] == s0);
                rchk[tid] = u ? s0 : -2;
            }
            __syncthreads();
            if (tid < RC_SYM_TC) {
                // direction 1: column b = c0 + tid gathers the rows of the tile, grouped by the rows' slots
                const int b = c0 + tid;
#pragma unroll 1  // keeps the kernel under 192 VGPRs so that a k_resolve block fits on the CU beside two of these
                for (int ch = 0; ch < RC_SYM_TR / 8; ++ch) {
                    long long xd[8], xl[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) { xd[q] = tt[0][ch * 8 + q][tid]; xl[q] = tt[1][ch * 8 + q][tid]; }
                    const int cs_ = __builtin_amdgcn_readfirstlane(rchk[ch]);
                    if (cs_ != -2) {
                        if (cs_ != cur) {
                            if (cur >= 0) {
                                if (accD) __hip_atomic_fetch_add((u64 *)(SD + (size_t)cur * ld + b), (u64)accD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (accL) __hip_atomic_fetch_add((u64 *)(SL + (size_t)cur * ld + b), (u64)accL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            accD = accL = 0; cur = cs_;
                        }
                        accD += ((xd[0] + xd[1]) + (xd[2] + xd[3])) + ((xd[4] + xd[5]) + (xd[6] + xd[7]));
                        accL += ((xl[0] + xl[1]) + (xl[2] + xl[3])) + ((xl[4] + xl[5]) + (xl[6] + xl[7]));
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int sr = __builtin_amdgcn_readfirstlane(rslot[ch * 8 + q]);
                            if (sr != cur) {
                                if (cur >= 0) {
                                    if (accD) __hip_atomic_fetch_add((u64 *)(SD + (size_t)cur * ld + b), (u64)accD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    if (accL) __hip_atomic_fetch_add((u64 *)(SL + (size_t)cur * ld + b), (u64)accL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                                accD = accL = 0; cur = sr;
                            }
                            accD += xd[q]; accL += xl[q];
                        }
                    }
                }
            } else {
                // direction 2: wave 2 -> columns 0..63, wave 3 -> 64..127; lanes 0-31 rows of D, lanes 32-63 rows of logD
                const int q2 = tid - RC_SYM_TC, half = q2 >> 6, r = q2 & 31, mat = (q2 >> 5) & 1;
                long long *S = mat ? SL : SD;
                const int arow = r0 + r;
                long long acc = 0;
                int cc = -1;
#pragma unroll
                for (int ch = 0; ch < 8; ++ch) {
                    const int cb = half * 64 + ch * 8;
                    const ll2 x0 = *(const ll2 *)&tt[mat][r][cb], x1 = *(const ll2 *)&tt[mat][r][cb + 2];
                    const ll2 x2 = *(const ll2 *)&tt[mat][r][cb + 4], x3 = *(const ll2 *)&tt[mat][r][cb + 6];
                    const int cs_ = __builtin_amdgcn_readfirstlane(cchk[cb >> 3]);
                    if (cs_ != -2) {
                        if (cs_ != cc) {
                            if (cc >= 0 && acc && arow < n) __hip_atomic_fetch_add((u64 *)(S + (size_t)cc * ld + arow), (u64)acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            acc = 0; cc = cs_;
                        }
                        acc += ((x0.x + x0.y) + (x1.x + x1.y)) + ((x2.x + x2.y) + (x3.x + x3.y));
                    } else {
                        const long long xs[8] = {x0.x, x0.y, x1.x, x1.y, x2.x, x2.y, x3.x, x3.y};
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int sc = __builtin_amdgcn_readfirstlane(cslot[cb + q]);
                            if (sc != cc) {
                                if (cc >= 0 && acc && arow < n) __hip_atomic_fetch_add((u64 *)(S + (size_t)cc * ld + arow), (u64)acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                acc = 0; cc = sc;
                            }
                            acc += xs[q];
                        }
                    }
                }
                if (cc >= 0 && acc && arow < n) __hip_atomic_fetch_add((u64 *)(S + (size_t)cc * ld + arow), (u64)acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (tid < RC_SYM_TC && cur >= 0) {
            const int b = c0 + tid;
            if (accD) __hip_atomic_fetch_add((u64 *)(SD + (size_t)cur * ld + b), (u64)accD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (accL) __hip_atomic_fetch_add((u64 *)(SL + (size_t)cur * ld + b), (u64)accL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // diagonal (S includes j = i): D[a][a] -> S[slot_a][a]; logD's diagonal is 0 (types.jl:155)
        if (t_begin == 0 && tid < RC_SYM_TC) {
            const int a_ = c0 + tid;
            if (a_ < n) {
                const long long x = V.diagq[a_];
                if (x) __hip_atomic_fetch_add((u64 *)(SD + (size_t)slot[a_] * ld + a_), (u64)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// k_bulk_symw — wave-autonomous symmetric row-bucket reduction (64-bit storage).  Same contract as k_bulk_sym (upper
// triangle only, exact integer result for any labelling), different machine mapping: no LDS tiles and no block
// barriers.  A work item is 64 rows × 128 columns of the upper triangle and belongs to ONE wave:
//   * lane ℓ owns columns c0+2ℓ, c0+2ℓ+1 and streams the rows with 16-byte non-temporal loads, several rows in flight;
//   * direction 1 (S[slot_row][col] += x): register accumulators per lane, flushed when the row's slot changes — as k_bulk;
//   * direction 2 (S[slot_col][row] += x): the 128 columns of a row belong to one or two clusters, so per row and
//     distinct column slot the wave reduces its lanes' values with DPP adds (no LDS traffic) and lane (row − a0) keeps
//     the total; after the 64 rows one coalesced 64-bit atomic per matrix and slot writes them out.
//   Waves never wait for each other, so loads, the logs of the derived mode and the reductions of different waves
//   overlap freely, and occupancy is bounded by registers only.
// ---------------------------------------------------------------------------------------------------
#define RC_SW_ROWS 64
#define RC_SW_FINE 8    // rows per unit in the light column blocks handed out last (k_bulk_syml)
#define RC_SW_COLS 128
#define RC_SW_U 8

// Sums of TWO 64-bit values over the 64 lanes of a wave, results uniform.  DPP butterfly with the permuted operand
// fused into v_add_co / v_addc_co (the compiler's own lowering of the same reduction needs 38 VALU instructions per
// value — zero-fills, v_mov_dpp, 64-bit add — against 12 here).  The two reductions are interleaved, so an
// instruction reads a register written four instructions earlier: the 2 wait states a DPP read needs after a VALU
// write are covered without s_nop.  Steps: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror (every lane of
// a 16-lane row holds the row total), row_bcast15 -> rows 1, 3, row_bcast31 -> rows 2, 3: lane 63 holds the total.
__device__ __forceinline__ void wave_sum2_dpp(long long &a, long long &b)
{
    unsigned alo = (unsigned)(u64)a, ahi = (unsigned)((u64)a >> 32), blo = (unsigned)(u64)b, bhi = (unsigned)((u64)b >> 32);
#define RC_DPP_STEP(ctrl)                                            \
    "v_add_co_u32_dpp %0, vcc, %0, %0 " ctrl "\n\t"                  \
    "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc " ctrl "\n\t"            \
    "v_add_co_u32_dpp %2, vcc, %2, %2 " ctrl "\n\t"                  \
    "v_addc_co_u32_dpp %3, vcc, %3, %3, vcc " ctrl "\n\t"
    asm volatile("s_nop 1\n\t"
                 RC_DPP_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 RC_DPP_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 RC_DPP_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
                 RC_DPP_STEP("row_mirror row_mask:0xf bank_mask:0xf")
                 RC_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 RC_DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 0"
                 : "+v"(alo), "+v"(ahi), "+v"(blo), "+v"(bhi)
                 :
                 : "vcc");
#undef RC_DPP_STEP
    const unsigned ral = (unsigned)__builtin_amdgcn_readlane((int)alo, 63), rah = (unsigned)__builtin_amdgcn_readlane((int)ahi, 63);
    const unsigned rbl = (unsigned)__builtin_amdgcn_readlane((int)blo, 63), rbh = (unsigned)__builtin_amdgcn_readlane((int)bhi, 63);
    a = (long long)(((u64)rah << 32) | ral);
    b = (long long)(((u64)rbh << 32) | rbl);
}

template <bool DERIVED>
__global__ __launch_bounds__(256) void k_bulk_symw(View V, int wgen, int zgen, int sgen, int cgen, int nitems)
{
    __shared__ double2 ltab_sh[DERIVED ? 128 : 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const size_t ld = (size_t)V.ld;
    if (DERIVED && tid < 128) ltab_sh[tid] = V.ltab[tid];
    (void)zgen;  // generations are cleared and work counters re-armed by k_resolve (SweepArgs.zero_gen)
    __syncthreads();  // the table is visible; from here on the waves are on their own
    const long long *__restrict__ Dq = (const long long *)V.Dq;
    const long long *__restrict__ Lq = (const long long *)V.Lq;
    const int *__restrict__ slot = V.snap[sgen];
    long long *SD = V.SD[wgen], *SL = V.SL[wgen];
    const int n = V.n;
    const int ncb = (n + RC_SW_COLS - 1) / RC_SW_COLS;
    const int qeD = V.qeD;
    const double qsL = V.qsL;
    int *counter = V.work[cgen];
    auto add64 = [](long long *p, long long v) { __hip_atomic_fetch_add((u64 *)p, (u64)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // Units are dealt round-robin to the waves: a shared work counter would be hit by thousands of waves at once
    // (a returning atomic on one address costs ~25 ns each in L2: 100 µs for 4096 waves) and all units of one
    // granularity cost the same anyway.
    const int nwaves = (int)gridDim.x * 4;
    (void)counter;
    for (int unit = (int)blockIdx.x * 4 + (tid >> 6); unit < nitems; unit += nwaves) {
        int item = unit;
        int J = ncb - 1;
        for (;; --J) {   // heavy column blocks first
            const int cnt = (min(RC_SW_COLS * J + RC_SW_COLS, n) + RC_SW_ROWS - 1) / RC_SW_ROWS;
            if (item < cnt) break;
            item -= cnt;
        }
        const int c0 = J * RC_SW_COLS, a0 = item * RC_SW_ROWS;
        const int a1 = min(a0 + RC_SW_ROWS, min(c0 + RC_SW_COLS, n));   // rows a >= c0+128 have no column b > a here
        const int col0 = c0 + 2 * lane, col1 = col0 + 1;
        const int cs0 = col0 < n ? slot[col0] : -1, cs1 = col1 < n ? slot[col1] : -1;
        const int rowslots = (a0 + lane < n) ? slot[a0 + lane] : -1;   // slot of row a0 + lane (read back with readlane)
        // distinct column slots of this item (cluster-contiguous layout: one or two); beyond four: slow path
        int ds0 = -1, ds1 = -1, ds2 = -1, ds3 = -1, M = 0;
        bool rem0 = cs0 >= 0, rem1 = cs1 >= 0;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const u64 b0 = __ballot(rem0), b1 = __ballot(rem1);
            if (!(b0 | b1)) break;
            const int s_ = b0 ? __builtin_amdgcn_readlane(cs0, __ffsll((long long)b0) - 1) : __builtin_amdgcn_readlane(cs1, __ffsll((long long)b1) - 1);
            if (m == 0) ds0 = s_; else if (m == 1) ds1 = s_; else if (m == 2) ds2 = s_; else ds3 = s_;
            M = m + 1;
            rem0 = rem0 && cs0 != s_;
            rem1 = rem1 && cs1 != s_;
        }
        const bool overflow = (__ballot(rem0) | __ballot(rem1)) != 0;   // uniform
        const bool uni = (M == 1) && !overflow && __ballot(cs0 == ds0 && cs1 == ds0) == ~0ull;   // all 128 columns in one cluster
        long long aD0 = 0, aD1 = 0, aL0 = 0, aL1 = 0;       // direction 1, slot `cur`
        long long rD0 = 0, rD1 = 0, rD2 = 0, rD3 = 0, rL0 = 0, rL1 = 0, rL2 = 0, rL3 = 0;   // direction 2, row a0 + lane
        int cur = -1;
        auto flush1 = [&]() {
            if (cur >= 0) {
                if (aD0) add64(SD + (size_t)cur * ld + col0, aD0);
                if (aD1) add64(SD + (size_t)cur * ld + col1, aD1);
                if (aL0) add64(SL + (size_t)cur * ld + col0, aL0);
                if (aL1) add64(SL + (size_t)cur * ld + col1, aL1);
            }
            aD0 = aD1 = aL0 = aL1 = 0;
        };
        for (int a = a0; a < a1; a += RC_SW_U) {
            ll2 d[RC_SW_U], l[RC_SW_U];
#pragma unroll
            for (int u = 0; u < RC_SW_U; ++u) {
                const int r = min(a + u, n - 1);
                d[u] = __builtin_nontemporal_load((const ll2 *)(Dq + (size_t)r * ld + col0));
                if (!DERIVED) l[u] = __builtin_nontemporal_load((const ll2 *)(Lq + (size_t)r * ld + col0));
            }
#pragma unroll
            for (int u = 0; u < RC_SW_U; ++u) {
                const int row = a + u;
                if (row >= a1) break;                                   // uniform
                ll2 x = d[u], y;
                if (!DERIVED) y = l[u];
                if (!(col0 > row)) { x.x = 0; y.x = 0; }               // strictly upper triangle only
                if (!(col1 > row)) { x.y = 0; y.y = 0; }
                if (DERIVED) { y.x = rc_qlog(x.x, qeD, qsL, ltab_sh); y.y = rc_qlog(x.y, qeD, qsL, ltab_sh); }
                // direction 1
                const int sr = __builtin_amdgcn_readlane(rowslots, row - a0);
                if (sr != cur) { flush1(); cur = sr; }
                aD0 += x.x; aD1 += x.y; aL0 += y.x; aL1 += y.y;
                // direction 2
                const bool mine = lane == row - a0;
                if (uni) {
                    long long tD = x.x + x.y, tL = y.x + y.y;
                    wave_sum2_dpp(tD, tL);
                    if (mine) { rD0 += tD; rL0 += tL; }
                } else {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        if (m >= M) break;                              // uniform
                        const int s_ = m == 0 ? ds0 : m == 1 ? ds1 : m == 2 ? ds2 : ds3;
                        long long tD = (cs0 == s_ ? x.x : 0) + (cs1 == s_ ? x.y : 0);
                        long long tL = (cs0 == s_ ? y.x : 0) + (cs1 == s_ ? y.y : 0);
                        wave_sum2_dpp(tD, tL);
                        if (mine) {
                            if (m == 0) { rD0 += tD; rL0 += tL; } else if (m == 1) { rD1 += tD; rL1 += tL; }
                            else if (m == 2) { rD2 += tD; rL2 += tL; } else { rD3 += tD; rL3 += tL; }
                        }
                    }
                    if (overflow) {   // columns of a fifth, sixth ... cluster: element-wise atomics (rare)
                        if (rem0) { if (x.x) add64(SD + (size_t)cs0 * ld + row, x.x); if (y.x) add64(SL + (size_t)cs0 * ld + row, y.x); }
                        if (rem1) { if (x.y) add64(SD + (size_t)cs1 * ld + row, x.y); if (y.y) add64(SL + (size_t)cs1 * ld + row, y.y); }
                    }
                }
            }
        }
        flush1();
        {   // direction 2 write-out: lane r holds the totals of row a0 + r
            const int row = a0 + lane;
            if (row < a1) {
                if (M > 0) { if (rD0) add64(SD + (size_t)ds0 * ld + row, rD0); if (rL0) add64(SL + (size_t)ds0 * ld + row, rL0); }
                if (M > 1) { if (rD1) add64(SD + (size_t)ds1 * ld + row, rD1); if (rL1) add64(SL + (size_t)ds1 * ld + row, rL1); }
                if (M > 2) { if (rD2) add64(SD + (size_t)ds2 * ld + row, rD2); if (rL2) add64(SL + (size_t)ds2 * ld + row, rL2); }
                if (M > 3) { if (rD3) add64(SD + (size_t)ds3 * ld + row, rD3); if (rL3) add64(SL + (size_t)ds3 * ld + row, rL3); }
            }
        }
        // diagonal (S includes j = i): D[a][a] -> S[slot_a][a]; logD's diagonal is 0 (types.jl:155)
        if (item == 0) {
            if (col0 < n) { const long long x = V.diagq[col0]; if (x) add64(SD + (size_t)cs0 * ld + col0, x); }
            if (col1 < n) { const long long x = V.diagq[col1]; if (x) add64(SD + (size_t)cs1 * ld + col1, x); }
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// k_bulk_syml — wave-autonomous symmetric reduction with a wave-PRIVATE LDS transposition (64-bit storage).
// Like k_bulk_symw a wave owns a 64-row × 128-column item and never meets a block barrier; direction 2 is done the
// cheap way: four rows at a time the wave writes its tile (4 × 128 values per matrix) to its own 8 KiB of LDS and reads
// it back transposed — lane = (row r = lane >> 4, column octet q = lane & 15) sums 8 values per matrix — followed by a
// 4-step DPP reduction inside the 16-lane DPP row.  LDS operations of one wave execute in order, so no
// synchronisation is needed.  The 16 lanes of DPP row r then all hold the totals of tile row r; lane 16r + k keeps
// them for tile k of the item, i.e. lane ℓ accumulates row a0 + 4(ℓ & 15) + (ℓ >> 4).
// ---------------------------------------------------------------------------------------------------
#define RC_SL_R 4
#ifndef RC_SL_LOGS
#define RC_SL_LOGS 2   // logs evaluated together (2 or 4): four need ~20 more VGPRs, which at the 128 of four waves per SIMD spill inside the tile loop
#endif
// LDS row of a tile: 16 column octets of 8 values, each padded to 10 (80 B): the transposed b128 reads of a quarter wave
// (16 lanes, one octet each) then fall on 16 different 16-byte bank groups instead of 4
#define RC_SL_O 10
#define RC_SL_P (16 * RC_SL_O)

// sums of four 64-bit values over each 16-lane DPP row (all lanes of the row receive the totals)
__device__ __forceinline__ void row16_sum4_dpp(long long &a, long long &b, long long &c, long long &d)
{
    unsigned r0 = (unsigned)(u64)a, r1 = (unsigned)((u64)a >> 32), r2 = (unsigned)(u64)b, r3 = (unsigned)((u64)b >> 32);
    unsigned r4 = (unsigned)(u64)c, r5 = (unsigned)((u64)c >> 32), r6 = (unsigned)(u64)d, r7 = (unsigned)((u64)d >> 32);
#define RC_DPP_STEP(ctrl)                                            \
    "v_add_co_u32_dpp %0, vcc, %0, %0 " ctrl "\n\t"                  \
    "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc " ctrl "\n\t"            \
    "v_add_co_u32_dpp %2, vcc, %2, %2 " ctrl "\n\t"                  \
    "v_addc_co_u32_dpp %3, vcc, %3, %3, vcc " ctrl "\n\t"            \
    "v_add_co_u32_dpp %4, vcc, %4, %4 " ctrl "\n\t"                  \
    "v_addc_co_u32_dpp %5, vcc, %5, %5, vcc " ctrl "\n\t"            \
    "v_add_co_u32_dpp %6, vcc, %6, %6 " ctrl "\n\t"                  \
    "v_addc_co_u32_dpp %7, vcc, %7, %7, vcc " ctrl "\n\t"
    asm volatile("s_nop 1\n\t"
                 RC_DPP_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 RC_DPP_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 RC_DPP_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
                 RC_DPP_STEP("row_mirror row_mask:0xf bank_mask:0xf")
                 "s_nop 0"
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                 :
                 : "vcc");
#undef RC_DPP_STEP
    a = (long long)(((u64)r1 << 32) | r0); b = (long long)(((u64)r3 << 32) | r2);
    c = (long long)(((u64)r5 << 32) | r4); d = (long long)(((u64)r7 << 32) | r6);
}

#ifdef RC_PROF_SYML
#define RC_PF(stmt) stmt
#else
#define RC_PF(stmt)
#endif

// The unit loop of the wave-autonomous symmetric reduction (k_bulk_syml).  tt: the calling wave's private 10 KiB of LDS (log table in
// the padding, see k_bulk_syml).
//   Work units: column block J (heavy blocks first) × a range of rows.  Blocks J >= jsplit are cut into units of gcoarse
//   rows and the light blocks below jsplit, which come last in the list, into units of gfine (8..32) rows, so that the
//   tail of the launch is a few tiles long (and a small problem still spreads over the chip).
//   Units are dealt round-robin to the waves (first_unit, first_unit + nwaves, ...): a shared work counter hit by
//   thousands of waves at once serialises in L2 (~25 ns per returning atomic), and handing units out dynamically from
//   eight counters with a graded tail measured SLOWER than this static list (serial sweep 117-130 µs against 111).
//   Unit pipeline: while the last tile of a unit is reduced, the next unit's first tile and the slots of its columns and
//   rows are already in flight (a unit's set-up otherwise costs a full memory round trip, ~20 % of a 64-row unit).
//   A unit's column set-up (the two clusters that own its 128 columns, per-lane class masks) is done once; its rows are
//   taken in 64-row halves (direction-2 totals live one row per lane).
// T: storage type of the matrices — long long, or int for 32-bit storage (logD stored): a lane then loads 8 instead of 16 bytes
// per row and matrix and widens its two entries on use; everything behind the load is the same 64-bit pipeline.
template <typename T> struct SymlRaw;
template <> struct SymlRaw<long long> { typedef ll2 vec; };
template <> struct SymlRaw<int> { typedef int vec __attribute__((ext_vector_type(2))); };
template <bool DERIVED, typename T = long long>
__device__ __forceinline__ void syml_units(const View &V, long long (*tt)[RC_SL_R][RC_SL_P], int wgen, int sgen, int nitems,
                                           int jsplit, int gfine, int gcoarse, int first_unit, int nwaves, long long *pf_out,
                                           const int4 *__restrict__ ulist = nullptr)
{
    typedef typename SymlRaw<T>::vec rawvec;
    const int lane = threadIdx.x & 63;
    const size_t ld = (size_t)V.ld;
    const T *__restrict__ Dq = (const T *)V.Dq + 2 * lane;
    const T *__restrict__ Lq = (const T *)V.Lq + 2 * lane;
    const int *__restrict__ slot = V.snap[sgen];
    long long *SD = V.SD[wgen], *SL = V.SL[wgen];
    const int n = V.n;
    const int ncb = (n + RC_SW_COLS - 1) / RC_SW_COLS;
    const int qeD = V.qeD;
    const double qsL = V.qsL;
    const int tr = lane >> 4, tq = lane & 15;   // transposed role: tile row, column octet
#ifdef RC_EXP_NO_FLUSH   // timing experiment (profiles/r02): every flush computed, none issued
    auto add64 = [](long long *p, long long v) { if (v == 0x7fffffffffffffffll) __hip_atomic_fetch_add((u64 *)p, (u64)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
#else
    auto add64 = [](long long *p, long long v) { __hip_atomic_fetch_add((u64 *)p, (u64)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
#endif
    RC_PF(long long pf_wait = 0; long long pf_tiles = 0; long long pf_setup = 0; long long pf_log = 0; long long pf_d2 = 0; long long pf_issue = 0; long long pf_ldsw = 0; long long pf_d1 = 0;)
    RC_PF(const long long pf_t0 = __builtin_amdgcn_s_memtime(); const long long pf_r0 = __builtin_amdgcn_s_memrealtime();)
    (void)pf_out;
    // unit index -> (column block, first row, end row)
    int c0 = 0, a0 = 0, a1 = 0, item = 0;
    auto decode = [&](int u, int &oc0, int &oa0, int &oa1, int &oitem) {
        if (ulist) {                                                     // host-built list (k_bulk_syml2's slow units)
            const int4 e = ulist[u];
            oc0 = e.x; oa0 = e.y; oa1 = e.z; oitem = e.y == 0 ? 0 : 1;   // (item 0 = the column block's first unit: adds the diagonal)
            return;
        }
        int it = u, J = ncb - 1, g = gcoarse;
        for (;; --J) {
            g = (J >= jsplit) ? gcoarse : gfine;
            const int cnt = (min(RC_SW_COLS * J + RC_SW_COLS, n) + g - 1) / g;
            if (it < cnt) break;
            it -= cnt;
        }
        oc0 = J * RC_SW_COLS; oa0 = it * g; oitem = it;
        oa1 = min(oa0 + g, min(oc0 + RC_SW_COLS, n));                  // rows a >= c0+128 have no column b > a here
    };
    rawvec d[RC_SL_R], l[RC_SL_R];
    auto issue = [&](int a, int cb0) {
#pragma unroll
        for (int u = 0; u < RC_SL_R; ++u) {
            const int r = min(a + u, n - 1);
            d[u] = __builtin_nontemporal_load((const rawvec *)(Dq + (size_t)r * ld + cb0));
            if (!DERIVED) l[u] = __builtin_nontemporal_load((const rawvec *)(Lq + (size_t)r * ld + cb0));
        }
    };
    int unit = first_unit;
    int cs0 = -1, cs1 = -1, rowslots = -1;
    if (unit < nitems) {
        decode(unit, c0, a0, a1, item);
        issue(a0, c0);
        cs0 = (c0 + 2 * lane < n) ? slot[c0 + 2 * lane] : -1;
        cs1 = (c0 + 2 * lane + 1 < n) ? slot[c0 + 2 * lane + 1] : -1;
        rowslots = (a0 + lane < n) ? slot[a0 + lane] : -1;           // slot of row h0 + lane (read back with readlane)
    }
    while (unit < nitems) {
        RC_PF(const long long pf_u0 = __builtin_amdgcn_s_memtime();)
        const int nunit = unit + nwaves;
        int nc0 = 0, na0 = 0, na1 = 0, nitem = 0, ncs0 = -1, ncs1 = -1, nrowslots = -1;
        const int col0 = c0 + 2 * lane, col1 = col0 + 1;
        // The two clusters that own most of the 128 columns (A, B); columns of any other cluster go the slow way, element by
        // element.  Candidates: the clusters of the first, middle and last column and of the first column that differs
        // from the first; the two with the most columns win.  (Taking simply "the first column's cluster and the first that
        // differs" made one stray point ahead of a cluster boundary push the whole second cluster down the slow path:
        // 40 strays among 8192 points cost a factor of 4.)
        int dsA, dsB = -1;
        {
            const int k0 = __builtin_amdgcn_readfirstlane(cs0);         // column c0 always exists
            const u64 not0 = __ballot(cs0 >= 0 && cs0 != k0), not1 = __ballot(cs1 >= 0 && cs1 != k0);
            int k1 = -1;
            if (not0 | not1) {
                const int l0 = not0 ? __ffsll((long long)not0) - 1 : 64, l1 = not1 ? __ffsll((long long)not1) - 1 : 64;
                k1 = (l0 <= l1) ? __builtin_amdgcn_readlane(cs0, l0 & 63) : __builtin_amdgcn_readlane(cs1, l1 & 63);
            }
            dsA = k0; dsB = k1;
            if (k1 >= 0) {
                const int k2 = __builtin_amdgcn_readlane(cs0, 32), k3 = __builtin_amdgcn_readlane(cs1, 63);
                auto count = [&](int k) { return k < 0 ? 0 : __popcll(__ballot(cs0 == k)) + __popcll(__ballot(cs1 == k)); };
                const int n0 = count(k0), n1 = count(k1), n2 = (k2 == k0 || k2 == k1) ? 0 : count(k2),
                          n3 = (k3 == k0 || k3 == k1 || k3 == k2) ? 0 : count(k3);
                // largest and second largest of (k0,n0) (k1,n1) (k2,n2) (k3,n3)
                int ka = k0, na = n0, kb = k1, nbb = n1;
                if (nbb > na) { int t_ = ka; ka = kb; kb = t_; t_ = na; na = nbb; nbb = t_; }
                if (n2 > na) { kb = ka; nbb = na; ka = k2; na = n2; } else if (n2 > nbb) { kb = k2; nbb = n2; }
                if (n3 > na) { kb = ka; nbb = na; ka = k3; na = n3; } else if (n3 > nbb) { kb = k3; nbb = n3; }
                dsA = ka; dsB = kb;
            }
        }
        const bool rem0 = cs0 >= 0 && cs0 != dsA && cs0 != dsB, rem1 = cs1 >= 0 && cs1 != dsA && cs1 != dsB;
        const bool overflow = (__ballot(rem0) | __ballot(rem1)) != 0;   // uniform, rare
        // classes of my eight transposed columns c0 + 8 tq + j (bit j of mB: cluster B; of mO: neither, or padding) from
        // ballots over the lanes that hold those columns' slots (lane 4 tq + j/2, component j & 1): no second round of loads
        const bool inB0 = dsB >= 0 && cs0 == dsB, inB1 = dsB >= 0 && cs1 == dsB;
        const u64 bB0 = __ballot(inB0), bB1 = __ballot(inB1);
        const u64 bO0 = __ballot(cs0 != dsA && !inB0), bO1 = __ballot(cs1 != dsA && !inB1);
        auto spread4 = [](unsigned v) { return (v & 1u) | ((v & 2u) << 1) | ((v & 4u) << 2) | ((v & 8u) << 3); };
        const unsigned mB = spread4((unsigned)(bB0 >> (4 * tq)) & 0xFu) | (spread4((unsigned)(bB1 >> (4 * tq)) & 0xFu) << 1);
        const unsigned mO = spread4((unsigned)(bO0 >> (4 * tq)) & 0xFu) | (spread4((unsigned)(bO1 >> (4 * tq)) & 0xFu) << 1);
        long long aD0 = 0, aD1 = 0, aL0 = 0, aL1 = 0;                   // direction 1, slot `cur` (carried down the whole unit)
        long long rDA = 0, rLA = 0, rDB = 0, rLB = 0;                   // direction 2: row h0 + 4 (lane & 15) + (lane >> 4)
        int cur = -1, h0 = a0;
        auto flush1 = [&]() {
            if (cur >= 0) {
                if (aD0) add64(SD + (size_t)cur * ld + col0, aD0);
                if (aD1) add64(SD + (size_t)cur * ld + col1, aD1);
                if (aL0) add64(SL + (size_t)cur * ld + col0, aL0);
                if (aL1) add64(SL + (size_t)cur * ld + col1, aL1);
            }
            aD0 = aD1 = aL0 = aL1 = 0;
        };
        // after the loads of this unit's last tile have been taken: the next unit's first tile and slots
        auto prefetch_next = [&]() {
            if (nunit < nitems) {
                decode(nunit, nc0, na0, na1, nitem);
                issue(na0, nc0);
                ncs0 = (nc0 + 2 * lane < n) ? slot[nc0 + 2 * lane] : -1;
                ncs1 = (nc0 + 2 * lane + 1 < n) ? slot[nc0 + 2 * lane + 1] : -1;
                nrowslots = (na0 + lane < n) ? slot[na0 + lane] : -1;
            }
        };
        // One 4-row tile.  MASK = false for the units that lie entirely above the diagonal blocks, hold no padding column
        // and a whole number of tiles (97 % of them at n = 8192): every entry is a live strictly-upper one, so the eight
        // triangle selects and their compares per lane and row disappear.  In the derived mode the logs are taken from the
        // raw entries and masked afterwards with the same predicate (a masked or padding entry would otherwise need its
        // own dq > 0 test).
        const int colm0 = col0 < n ? col0 : -1, colm1 = col1 < n ? col1 : -1;   // padding columns never pass `col > row`
        auto tile = [&](auto maskc, int a) __attribute__((always_inline)) {
            constexpr bool MASK = decltype(maskc)::value;
            ll2 x[RC_SL_R], y[RC_SL_R];
            RC_PF({ const long long w0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pf_wait += __builtin_amdgcn_s_memtime() - w0; pf_tiles += 1; })
#pragma unroll
            for (int u = 0; u < RC_SL_R; ++u) {
                x[u].x = d[u].x; x[u].y = d[u].y;
                if (!DERIVED) { y[u].x = l[u].x; y[u].y = l[u].y; }
            }
            RC_PF(const long long pi0 = __builtin_amdgcn_s_memtime();)
            if (a + RC_SL_R < a1) issue(a + RC_SL_R, c0);               // next tile in flight under this one's work
            else prefetch_next();
            RC_PF(const long long pl0 = __builtin_amdgcn_s_memtime(); pf_issue += pl0 - pi0;)
            if (DERIVED) {   // the eight logs of this tile, RC_SL_LOGS at a time: table reads first, then the arithmetic
#pragma unroll
                for (int h = 0; h < RC_SL_R; h += RC_SL_LOGS / 2) {
                    QlogPrep pp[RC_SL_LOGS];
                    double2 tv[RC_SL_LOGS];
#pragma unroll
                    for (int u = 0; u < RC_SL_LOGS; ++u) pp[u] = rc_qlog_prep((u & 1) ? x[h + u / 2].y : x[h + u / 2].x, qeD);
#pragma unroll
                    for (int u = 0; u < RC_SL_LOGS; ++u) tv[u] = *(const double2 *)(&tt[0][0][0] + pp[u].j * RC_SL_O + 8);
#pragma unroll
                    for (int u = 0; u < RC_SL_LOGS; ++u) {
                        const long long q = rc_qlog_raw(pp[u], tv[u], qsL);
                        if (u & 1) y[h + u / 2].y = q; else y[h + u / 2].x = q;
                    }
                }
            }
            RC_PF(asm volatile("" ::: "memory"); pf_log += __builtin_amdgcn_s_memtime() - pl0;)
            if (MASK) {
#pragma unroll
                for (int u = 0; u < RC_SL_R; ++u) {
                    const int row = a + u;
                    const bool live = row < a1;
                    if (!(live && colm0 > row)) { x[u].x = 0; y[u].x = 0; }  // strictly upper triangle, rows of this item, no padding
                    if (!(live && colm1 > row)) { x[u].y = 0; y[u].y = 0; }
                }
            }
            RC_PF(const long long pw0 = __builtin_amdgcn_s_memtime();)
#pragma unroll
            for (int u = 0; u < RC_SL_R; ++u) {
                *(ll2 *)&tt[0][u][(lane >> 2) * RC_SL_O + (lane & 3) * 2] = x[u];
                *(ll2 *)&tt[1][u][(lane >> 2) * RC_SL_O + (lane & 3) * 2] = y[u];
            }
            RC_PF(const long long pw1 = __builtin_amdgcn_s_memtime(); pf_ldsw += pw1 - pw0;)
            // direction 1
#pragma unroll
            for (int u = 0; u < RC_SL_R; ++u) {
                const int row = a + u;
                if (!MASK || row < a1) {                                // uniform
                    const int sr = __builtin_amdgcn_readlane(rowslots, row - h0);
                    if (sr != cur) { flush1(); cur = sr; }
                    aD0 += x[u].x; aD1 += x[u].y; aL0 += y[u].x; aL1 += y[u].y;
                }
            }
            RC_PF(const long long pd0 = __builtin_amdgcn_s_memtime(); pf_d1 += pd0 - pw1;)
            // direction 2: transposed read of the wave's own tile (same-wave LDS operations execute in order)
            __builtin_amdgcn_wave_barrier();
            long long sDA = 0, sLA = 0, sDB = 0, sLB = 0;
            {
                long long vD[8], vL[8];
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const ll2 pd = *(const ll2 *)&tt[0][tr][RC_SL_O * tq + j], pl = *(const ll2 *)&tt[1][tr][RC_SL_O * tq + j];
                    vD[j] = pd.x; vD[j + 1] = pd.y; vL[j] = pl.x; vL[j + 1] = pl.y;
                }
                if ((mB | mO) == 0) {                                   // all eight columns in cluster A (the usual case)
                    sDA = ((vD[0] + vD[1]) + (vD[2] + vD[3])) + ((vD[4] + vD[5]) + (vD[6] + vD[7]));
                    sLA = ((vL[0] + vL[1]) + (vL[2] + vL[3])) + ((vL[4] + vL[5]) + (vL[6] + vL[7]));
                } else if (mB == 0xffu) {
                    sDB = ((vD[0] + vD[1]) + (vD[2] + vD[3])) + ((vD[4] + vD[5]) + (vD[6] + vD[7]));
                    sLB = ((vL[0] + vL[1]) + (vL[2] + vL[3])) + ((vL[4] + vL[5]) + (vL[6] + vL[7]));
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const bool isB = (mB >> j) & 1, isO = (mO >> j) & 1;
                        if (isB) { sDB += vD[j]; sLB += vL[j]; } else if (!isO) { sDA += vD[j]; sLA += vL[j]; }
                    }
                }
                if (overflow) {   // columns of a third cluster: element-wise atomics (uniform branch: every lane shuffles)
                    const int row = a + tr;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int cj = c0 + 8 * tq + j;
                        // slot of column cj from the lane that holds it (lane 4 tq + j/2, component j & 1): an LDS-crossbar
                        // shuffle, not a load — a global load here would make the wave wait for the next tile's prefetch too
                        const int sj = __shfl((j & 1) ? cs1 : cs0, 4 * tq + (j >> 1));
                        if (((mO >> j) & 1) && cj < n && (!MASK || row < a1)) {
                            if (vD[j]) add64(SD + (size_t)sj * ld + row, vD[j]);
                            if (vL[j]) add64(SL + (size_t)sj * ld + row, vL[j]);
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            row16_sum4_dpp(sDA, sLA, sDB, sLB);
            if (tq == ((a - h0) >> 2)) { rDA += sDA; rLA += sLA; rDB += sDB; rLB += sLB; }
            RC_PF(pf_d2 += __builtin_amdgcn_s_memtime() - pd0;)
        };
        const bool interior = (a1 <= c0) && (c0 + RC_SW_COLS <= n) && (((a1 - a0) & (RC_SL_R - 1)) == 0);   // uniform
        RC_PF(pf_setup += __builtin_amdgcn_s_memtime() - pf_u0;)
        for (h0 = a0; h0 < a1; h0 += RC_SW_ROWS) {                      // 64-row halves of a longer unit
            const int h1 = min(h0 + RC_SW_ROWS, a1);
            if (h0 > a0) rowslots = (h0 + lane < n) ? slot[h0 + lane] : -1;
            if (interior) { for (int a = h0; a < h1; a += RC_SL_R) tile(std::false_type{}, a); }
            else { for (int a = h0; a < h1; a += RC_SL_R) tile(std::true_type{}, a); }
            // direction 2 write-out: lane ℓ holds the totals of row h0 + 4 (ℓ & 15) + (ℓ >> 4)
            const int row = h0 + 4 * tq + tr;
            if (row < h1) {
                if (rDA) add64(SD + (size_t)dsA * ld + row, rDA);
                if (rLA) add64(SL + (size_t)dsA * ld + row, rLA);
                if (dsB >= 0) { if (rDB) add64(SD + (size_t)dsB * ld + row, rDB); if (rLB) add64(SL + (size_t)dsB * ld + row, rLB); }
            }
            rDA = rLA = rDB = rLB = 0;
        }
        flush1();
        // diagonal (S includes j = i): D[a][a] -> S[slot_a][a]; logD's diagonal is 0 (types.jl:155)
        if (item == 0) {
            if (col0 < n) { const long long x = V.diagq[col0]; if (x) add64(SD + (size_t)cs0 * ld + col0, x); }
            if (col1 < n) { const long long x = V.diagq[col1]; if (x) add64(SD + (size_t)cs1 * ld + col1, x); }
        }
        unit = nunit; c0 = nc0; a0 = na0; a1 = na1; item = nitem; cs0 = ncs0; cs1 = ncs1; rowslots = nrowslots;
    }
    RC_PF(if (lane == 0 && pf_out) { pf_out[0] = __builtin_amdgcn_s_memtime() - pf_t0; pf_out[1] = pf_wait; pf_out[2] = pf_tiles; pf_out[3] = pf_setup;
                                     pf_out[4] = __builtin_amdgcn_s_memrealtime() - pf_r0; pf_out[5] = pf_log; pf_out[6] = pf_d2;
                                     pf_out[7] = (pf_r0 << 20) | (long long)((__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xF) << 16) |
                                                 (long long)(__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (15 << 11)) & 0xFFFF);
                                     pf_out[8] = pf_issue; pf_out[9] = pf_ldsw; pf_out[10] = pf_d1; })
}

// the log table of the derived mode into the padding of every wave's private tile (see k_bulk_syml)
__device__ __forceinline__ void syml_load_table(const View &V, long long (*tl)[2][RC_SL_R][RC_SL_P])
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = lane + 64 * h;
        *(double2 *)(&tl[wv][0][0][0] + j * RC_SL_O + 8) = V.ltab[j];   // octets are contiguous: entry j sits in octet j
    }
}

#ifndef RC_SYML_MINWAVES
#define RC_SYML_MINWAVES 4
#endif
template <bool DERIVED, typename T>
__device__ __forceinline__ void syml_kernel_body(const View &V, int wgen, int zgen, int sgen, int cgen, int nitems, int jsplit, int gfine, int gcoarse)
{
    // [wave][matrix][row][octet-padded col]: 10 KiB per wave, 40 KiB per block = four blocks (16 waves) per CU exactly.
    // The log table of the derived mode (128 × 16 B) lives in the padding: entry j of a wave's private copy sits in
    // the two spare elements of octet (j & 15) of tile row (j >> 4) [matrix = j >> 6, row = (j >> 4) & 3].
    __shared__ __attribute__((aligned(16))) long long tl[4][2][RC_SL_R][RC_SL_P];
    if (DERIVED) syml_load_table(V, tl);
    (void)zgen;  // generations are cleared and work counters re-armed by k_resolve (SweepArgs.zero_gen)
    __syncthreads();  // the table is visible; from here on the waves are on their own
    // the wave's number as a SCALAR (readfirstlane: threadIdx.x >> 6 is wave-uniform, but only the programmer knows): unit numbers,
    // row ranges, tile loops and the row part of every address then live in SGPRs — scalar branches instead of exec-mask
    // save / restore pairs around every uniform `if`, s_mul / s_add instead of v_mad_i64_i32 per loaded row
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int w = (int)blockIdx.x * 4 + wv;
    long long *pf = nullptr;
    RC_PF(if (w < 8192) pf = (long long *)((char *)V.work[cgen] + 64) + (size_t)w * 16;)
    (void)cgen;
    syml_units<DERIVED, T>(V, tl[wv], wgen, sgen, nitems, jsplit, gfine, gcoarse, w, (int)gridDim.x * 4, pf);
}
template <bool DERIVED>
__global__ __launch_bounds__(256, RC_SYML_MINWAVES) void k_bulk_syml(View V, int wgen, int zgen, int sgen, int cgen, int nitems, int jsplit, int gfine, int gcoarse)
{
    syml_kernel_body<DERIVED, long long>(V, wgen, zgen, sgen, cgen, nitems, jsplit, gfine, gcoarse);
}
// the same for 32-bit storage (logD stored): RC_SYM_VARIANT=2 on a 32-bit context
__global__ __launch_bounds__(256, RC_SYML_MINWAVES) void k_bulk_syml32(View V, int wgen, int zgen, int sgen, int cgen, int nitems, int jsplit, int gfine, int gcoarse)
{
    syml_kernel_body<false, int>(V, wgen, zgen, sgen, cgen, nitems, jsplit, gfine, gcoarse);
}


// ---------------------------------------------------------------------------------------------------
// k_bulk_syml2 — the wave-autonomous symmetric reduction again, its common case rewritten around the instruction count (round 3).
// k_bulk_syml measured: 25.5 M VALU wave-instructions per launch at n = 8192 (48 per matrix entry), 43 % of a wave's cycles issuing,
// 25 % stalled behind the other waves of its SIMD, 33 % parked on LDS round trips (profiles/r02/sq_counters_n8192.txt; per phase
// of a 4-row tile: the logs 26 %, direction 2 29 %, direction 1 11 %; the tile's loads are waited for 1 % of the time): bound by
// its own instruction stream, not by HBM.  What changed, same contract (upper triangle only, exact for any labelling):
//  * the unit list is built by the host (View.ufast / .uslow): no per-unit decode loop; the wave number is a scalar
//    (readfirstlane), so unit bounds, row loops and row addresses live in SGPRs;
//  * FAST units (97 % at n = 8192) lie entirely above the diagonal blocks, hold 128 real columns and a multiple of 8 rows:
//    no triangle masks at all;
//  * the log is 16 instructions instead of 23 (rc_qlog);
//  * direction 2 (S[slot_col][row]) sends ONE pre-added value per lane and row through LDS — x[col 2l] + x[col 2l+1], both
//    columns of a lane belong to one cluster except in the lane a cluster boundary splits — 8 rows at a time: the wave writes
//    8 × 64 values per matrix (ds_write_b64) and reads them back as (row = lane >> 3, lane octet = lane & 7), 8 values per lane:
//    7 adds and a 3-step DPP reduction over 8 lanes per 8 rows, where k_bulk_syml did 7 adds and 4 steps per 4 rows on
//    twice the LDS traffic; the B accumulators exist only in units whose columns span two clusters;
//    lanes that a boundary splits, or whose columns belong to a third cluster, add their elements with atomics (exact);
//  * the log table is shared by the block (2 KiB) and the tile buffers shrink to 8.25 KiB per wave: 35 KiB per block.
// The remaining units (diagonal blocks, the ragged last column block) go through the round-2 unit code (syml_units with a
// list), in the same launch, after the fast ones.
// ---------------------------------------------------------------------------------------------------
// LDS geometry of the fast path.  A 16-byte LDS access is served 16 lanes per cycle, each lane on one of the 16 bank quads.
//  * p-buffer pitch: 68 long longs (544 B = 34 quads): in a transposed read lane (tr, tq) fetches 16 B at quad 2·tr + 4·tq + j/2
//    (mod 16) — the lanes tq and tq + 4 of a row would meet on one quad, so the upper half of an octet row takes its four 16-byte
//    pieces in the order 1, 0, 3, 2 (RC_S2_SWZ; the pieces are only summed): the 16 lanes of two rows then cover the 16 quads
//    (round 3: pitch 66, no swizzle — SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 35 %).
//  * log table: RC_S2_TABREP copies of every 16-byte entry side by side (entry j, copy lane mod TABREP at quad TABREP·j + copy): the
//    table look-up is a gather by the entry's mantissa bits, different in every lane — with one copy 16 lanes throw their reads
//    at 16 quads at random.
#ifndef RC_S2_PITCH
#define RC_S2_PITCH 68
#endif
#ifndef RC_S2_SWZ
#define RC_S2_SWZ 1
#endif
#ifndef RC_S2_TABREP
#define RC_S2_TABREP 1
#endif
// (beyond 40 KiB of LDS per block four blocks no longer fit a CU and the compiler stops holding the kernel to 128 registers — three of
// these waves and one resolver wave per SIMD must fit its 512 — so the cap is stated)
#if RC_S2_TABREP > 1
#define RC_S2_VGPR_CAP __attribute__((amdgpu_num_vgpr(127)))
#else
#define RC_S2_VGPR_CAP
#endif

// sums of two / four 64-bit values over each group of 8 consecutive lanes (all lanes of the group receive the totals)
__device__ __forceinline__ void row8_sum2_dpp(long long &a, long long &b)
{
    unsigned r0 = (unsigned)(u64)a, r1 = (unsigned)((u64)a >> 32), r2 = (unsigned)(u64)b, r3 = (unsigned)((u64)b >> 32);
#define RC_DPP_STEP2(ctrl)                                           \
    "v_add_co_u32_dpp %0, vcc, %0, %0 " ctrl "\n\t"                  \
    "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc " ctrl "\n\t"            \
    "v_add_co_u32_dpp %2, vcc, %2, %2 " ctrl "\n\t"                  \
    "v_addc_co_u32_dpp %3, vcc, %3, %3, vcc " ctrl "\n\t"
    asm volatile("s_nop 1\n\t"
                 RC_DPP_STEP2("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 RC_DPP_STEP2("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 RC_DPP_STEP2("row_half_mirror row_mask:0xf bank_mask:0xf")
                 "s_nop 0"
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3)
                 :
                 : "vcc");
#undef RC_DPP_STEP2
    a = (long long)(((u64)r1 << 32) | r0); b = (long long)(((u64)r3 << 32) | r2);
}

// storage of D as the fast path streams it: int64 entries (16 B per lane and row), or — PACK — the 48-bit packed copy (12 B per
// lane and row: two entries in three dwords; View.Dq48), unpacked with three 32-bit operations
typedef unsigned rc_u3 __attribute__((ext_vector_type(3)));
typedef rc_u3 rc_u3a4 __attribute__((aligned(4)));
template <bool PACK> struct S2Raw;
template <> struct S2Raw<false> {
    typedef ll2 T;
    static __device__ __forceinline__ ll2 unpack(const ll2 &r) { return r; }
};
template <> struct S2Raw<true> {
    typedef rc_u3 T;
    static __device__ __forceinline__ ll2 unpack(const rc_u3 &r)
    {
        ll2 x;
        x.x = (long long)((u64)r.x | ((u64)(r.y & 0xffffu) << 32));
        x.y = (long long)((u64)r.z | ((u64)(r.y >> 16) << 32));
        return x;
    }
};

// a uniform pointer, opaque to the optimiser (it would otherwise fold the lane offset into the base and carry one 64-bit address
// per loaded row in vector registers): address = scalar pair + 32-bit lane offset, the scalar-base form of the global instructions
#define RC_GLOBAL_AS __attribute__((address_space(1)))
#define RC_CONST_AS __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ RC_GLOBAL_AS T *rc_uniform_ptr(T *p)
{
    const u64 v = (u64)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (RC_GLOBAL_AS T *)(((u64)hi << 32) | lo);   // (global address space kept: a generic pointer would make these flat loads)
}

// lane i receives lane i + 1's / lane i - 1's value (0 beyond the wave): wave-wide DPP shifts (gfx9: wave_shl / wave_shr)
__device__ __forceinline__ int wave_from_next(int v)
{
    int r = 0;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1" : "+v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ int wave_from_prev(int v)
{
    int r = 0;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1" : "+v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ long long wave_from_prev64(long long v)
{
    unsigned lo = 0, hi = 0;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_mov_b32_dpp %1, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1"
                 : "+v"(lo), "+v"(hi) : "v"((unsigned)(u64)v), "v"((unsigned)((u64)v >> 32)));
    return (long long)(((u64)hi << 32) | lo);
}

template <bool DERIVED, bool PACK>
__device__ __forceinline__ void syml2_fast(const View &V, long long *pb /* [2][8][RC_S2_PITCH], wave-private */, long long *bounce /* [128], wave-private */,
                                           const double2 *tab /* [128], block-shared */,
                                           int wgen, int sgen, int first_unit, int end_unit /* this wave's units: [first_unit, end_unit) of V.ufast */,
                                           long long *pf_out = nullptr)
{
    typedef typename S2Raw<PACK>::T raw_t;
    RC_PF(long long pf_wait = 0; long long pf_tiles = 0; long long pf_setup = 0; long long pf_log = 0; long long pf_d2 = 0; long long pf_issue = 0; long long pf_ldsw = 0; long long pf_d1 = 0;)
    RC_PF(const long long pf_t0 = __builtin_amdgcn_s_memtime(); const long long pf_r0 = __builtin_amdgcn_s_memrealtime();)
    (void)pf_out;
    const int lane = threadIdx.x & 63;
    const int tr = lane >> 3, tq = lane & 7;                            // transposed role: row of the 8-row group, lane octet
    const size_t ld = (size_t)V.ld;
    // every address is (uniform pointer) + (32-bit unsigned lane offset): the loads then take the scalar-base form and no per-row
    // 64-bit address lives in vector registers
    const long long *__restrict__ Dq = (const long long *)V.Dq;
    const unsigned *__restrict__ D48 = (const unsigned *)V.Dq48;              // PACK: 6 bytes per entry, rows of ld entries
    const long long *__restrict__ Lq = (const long long *)V.Lq;
    const unsigned lane2 = 2u * (unsigned)lane, lane3 = 3u * (unsigned)lane;
    const int *__restrict__ slot = V.snap[sgen];
    long long *SD = V.SD[wgen], *SL = V.SL[wgen];
    const int qeD = V.qeD;
    const double qsL = V.qsL;
    const double qthird = rc_third_vgpr(), qmq = rc_mquarter_sgpr();
#ifndef RC_S2_EXP
#define RC_S2_EXP 0      // timing experiments only (tools/syml_variants.py): 1 no atomics, 2 no logs, 4 no direction 2, 8 no direction 1
#endif
    auto add64 = [](long long *p, long long v) {
#ifdef RC_S2_WG_ATOMICS   // timing experiment only (NOT coherent across XCDs): do atomics that stay in the issuing XCD's L2 run faster?
        if (!(RC_S2_EXP & 1) || v == 0x7fffffffffffffffll) __hip_atomic_fetch_add((u64 *)p, (u64)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        if (!(RC_S2_EXP & 1) || v == 0x7fffffffffffffffll) __hip_atomic_fetch_add((u64 *)p, (u64)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    };
    long long *const pD = pb, *const pL = pb + 8 * RC_S2_PITCH;
    // TWO tiles (8 rows) in flight per wave: with the instruction count of a tile cut to a third, a tile is consumed faster than HBM
    // answers under load (measured with everything but the loads removed).  The rows are taken in pairs; the pair two tiles ahead is
    // requested the moment a pair is taken, so the registers holding loaded rows never exceed two tiles plus a pair.
#ifndef RC_S2_NP
#define RC_S2_NP 2       // row pairs in flight per wave: 4 = two tiles ahead, 2 = one (4 spills inside the loop at 128 registers: fatal, see k_bulk_syml2)
#endif
#ifndef RC_S2_LOGS
#define RC_S2_LOGS 2     // logs evaluated together (table reads first, then the arithmetic): 4 or 2 (4 spills, as above)
#endif
    constexpr int NP = RC_S2_NP;
    raw_t d[NP][2];
    ll2 l[NP][2];
    // the rows are requested strictly in order — down a unit, then on into the wave's next unit — so the (uniform) address of the
    // next row to request is carried along and advanced by the row pitch: two scalar adds per row instead of a 64-bit multiply
    const size_t pitchD = PACK ? ld / 2 * 3 : ld;                          // row pitch in elements of the streamed type's base pointer
    RC_GLOBAL_AS const unsigned *np48 = nullptr;
    RC_GLOBAL_AS const long long *npD = nullptr, *npL = nullptr;
    auto set_next = [&](int row, int cb0) {
        const size_t e = (size_t)row * ld + (size_t)cb0;
        if (PACK) np48 = rc_uniform_ptr(D48 + e / 2 * 3); else npD = rc_uniform_ptr(Dq + e);
        if (!DERIVED) npL = rc_uniform_ptr(Lq + e);
    };
    auto issue_pair = [&](int sl) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (PACK) { d[sl][u] = __builtin_nontemporal_load((RC_GLOBAL_AS const typename std::conditional<PACK, rc_u3a4, ll2>::type *)(np48 + lane3)); np48 += pitchD; }
            else { d[sl][u] = __builtin_nontemporal_load((RC_GLOBAL_AS const typename std::conditional<PACK, rc_u3a4, ll2>::type *)(npD + lane2)); npD += pitchD; }
            if (!DERIVED) { l[sl][u] = __builtin_nontemporal_load((RC_GLOBAL_AS const ll2 *)(npL + lane2)); npL += ld; }
        }
    };
    // (uniform by construction — the wave number is a scalar — but say so: unit bounds, row loops and row addresses stay in SGPRs)
    RC_CONST_AS const int *const ufast_w = (RC_CONST_AS const int *)(u64)V.ufast;   // scalar loads (a vector load here would make the wave wait for all its prefetched rows: one counter)
    auto load_unit = [&](int u_) {
        return make_int4(__builtin_amdgcn_readfirstlane(ufast_w[4 * u_]), __builtin_amdgcn_readfirstlane(ufast_w[4 * u_ + 1]), __builtin_amdgcn_readfirstlane(ufast_w[4 * u_ + 2]), 0);
    };
    int unit = __builtin_amdgcn_readfirstlane(first_unit);
    end_unit = __builtin_amdgcn_readfirstlane(end_unit);
    int4 U = make_int4(0, 0, 0, 0), Un = make_int4(0, 0, 0, 0);
    int cs0 = -1, cs1 = -1;
    if (unit < end_unit) {
        U = load_unit(unit);
        set_next(U.y, U.x);
#pragma unroll
        for (int q = 0; q < NP; ++q) issue_pair(q);
        cs0 = slot[min(U.x + 2 * lane, V.n - 1)]; cs1 = slot[min(U.x + 2 * lane + 1, V.n - 1)];   // (a ragged last block: padding columns take the last point's slot and carry zeros)
    }
    while (unit < end_unit) {
        RC_PF(long long pf_u0 = __builtin_amdgcn_s_memtime();)
        const int c0 = U.x, a0 = U.y, a1 = U.z;
        const int nunit = unit + 1;
        const bool have_next = nunit < end_unit;
        if (have_next) Un = load_unit(nunit);
        int ncs0 = -1, ncs1 = -1;
        // the two clusters that own most of the 128 columns (as k_bulk_syml: majority among four probes)
        int dsA, dsB = -1;
        {
            const int k0 = __builtin_amdgcn_readfirstlane(cs0);
            const u64 not0 = __ballot(cs0 != k0), not1 = __ballot(cs1 != k0);
            if (not0 | not1) {
                const int l0 = not0 ? __ffsll((long long)not0) - 1 : 64, l1 = not1 ? __ffsll((long long)not1) - 1 : 64;
                const int k1 = (l0 <= l1) ? __builtin_amdgcn_readlane(cs0, l0 & 63) : __builtin_amdgcn_readlane(cs1, l1 & 63);
                const int k2 = __builtin_amdgcn_readlane(cs0, 32), k3 = __builtin_amdgcn_readlane(cs1, 63);
                auto count = [&](int k) { return __popcll(__ballot(cs0 == k)) + __popcll(__ballot(cs1 == k)); };
                const int n0 = count(k0), n1 = count(k1), n2 = (k2 == k0 || k2 == k1) ? 0 : count(k2),
                          n3 = (k3 == k0 || k3 == k1 || k3 == k2) ? 0 : count(k3);
                int ka = k0, na = n0, kb = k1, nbb = n1;
                if (nbb > na) { int t_ = ka; ka = kb; kb = t_; t_ = na; na = nbb; nbb = t_; }
                if (n2 > na) { kb = ka; nbb = na; ka = k2; na = n2; } else if (n2 > nbb) { kb = k2; nbb = n2; }
                if (n3 > na) { kb = ka; nbb = na; ka = k3; na = n3; } else if (n3 > nbb) { kb = k3; nbb = n3; }
                dsA = ka; dsB = kb;
            } else {
                dsA = k0;
            }
            dsA = __builtin_amdgcn_readfirstlane(dsA); dsB = __builtin_amdgcn_readfirstlane(dsB);
        }
        // A lane is "whole" when its two columns belong to one of the two clusters together: its pre-added value goes through LDS.
        // A lane that the boundary between the two clusters splits (column 2 l in one, 2 l + 1 in the other — every other boundary)
        // is a DONOR: it keeps its first column as its value and hands the second to the lane on its right, whose columns are the
        // second cluster's (one wave-wide DPP shift per row and matrix, only in units that have such a lane).  Every other lane
        // (a third cluster, a stray point) adds its elements with atomics, row by row.
        const bool diag = a1 > c0 || c0 + RC_SW_COLS > V.n;               // uniform: the unit reaches into the diagonal block (triangle mask) or holds padding columns
        const int colx = c0 + (int)lane2, coly = colx + 1;
        const bool hasB = dsB >= 0;                                       // uniform
        const bool in0 = cs0 == dsA || (hasB && cs0 == dsB), in1 = cs1 == dsA || (hasB && cs1 == dsB);
        const bool whole0 = (cs0 == cs1) && in0;
        const int r_cs0 = wave_from_next(cs0), r_whole = wave_from_next(whole0 ? 1 : 0);
        const bool donor = !whole0 && in0 && in1 && lane < 63 && r_whole != 0 && r_cs0 == cs1;
        const bool whole = whole0 || donor;                               // contributes a value of class(cs0) through LDS
        const bool any_odd = __ballot(!whole) != 0;                       // uniform
        const bool any_donor = __ballot(donor) != 0;                      // uniform
        // class of the source lanes: bit l set = lane l's pre-added value belongs to cluster B.  Kept as the (scalar) ballot and cut
        // to the eight source lanes 8 tq .. 8 tq + 7 of a transposed value where it is used: a per-lane copy is one more long-lived
        // vector register, and a spilled register in this loop is fatal — its reload waits on the counter the prefetched rows use
        const u64 bB = __ballot(whole && cs0 == dsB);
        long long aD0 = 0, aD1 = 0, aL0 = 0, aL1 = 0;                   // direction 1, slot `cur`
        long long rDA = 0, rLA = 0, rDB = 0, rLB = 0;                   // direction 2: row h0 + 8 (lane & 7) + (lane >> 3)
        int cur = -1;
        // The 64-bit atomics execute at the memory side, 64-byte request by 64-byte request, at about a fifth of the read rate
        // — so every atomic instruction should fill its requests.  A lane's direction-1 sums are two adjacent columns (16 contiguous
        // bytes, but one instruction adds only one of them: 16 half-used requests each), its direction-2 totals the rows
        // 8 (l & 7) + (l >> 3) (64 different requests per instruction): both go through a 1 KiB bounce in the wave's LDS first,
        // after which lane l holds column c0 + l and c0 + 64 + l, or row h0 + l — 8 full requests per instruction.
        auto flush1 = [&]() {
            if (cur >= 0) {
                long long *const rowD = SD + (size_t)cur * ld + c0, *const rowL = SL + (size_t)cur * ld + c0;
                __builtin_amdgcn_wave_barrier();
                *(ll2 *)&bounce[2 * lane] = ll2{aD0, aD1};
                __builtin_amdgcn_wave_barrier();
                const long long d0 = bounce[lane], d1 = bounce[64 + lane];
                __builtin_amdgcn_wave_barrier();
                *(ll2 *)&bounce[2 * lane] = ll2{aL0, aL1};
                __builtin_amdgcn_wave_barrier();
                const long long l0 = bounce[lane], l1 = bounce[64 + lane];
                __builtin_amdgcn_wave_barrier();
                if (d0) add64(rowD + lane, d0);
                if (d1) add64(rowD + 64 + lane, d1);
                if (l0) add64(rowL + lane, l0);
                if (l1) add64(rowL + 64 + lane, l1);
            }
            aD0 = aD1 = aL0 = aL1 = 0;
        };
        for (int h0 = a0; h0 < a1; h0 += 64) {                            // 64-row halves (direction-2 totals live one row per lane)
            const int h1 = min(h0 + 64, a1);
            const int rowslots = (h0 + lane < h1) ? slot[h0 + lane] : -1; // slot of row h0 + lane (read back with readlane)
            // rows whose slot differs from the row before (bit l: row h0 + l; row h0 itself is compared with the slot being summed):
            // a pair of rows without such a bit — nearly all — adds into the running sums without looking at slots
            u64 chg = __ballot(rowslots != wave_from_prev(rowslots) && lane > 0 && h0 + lane < h1);
            if (__builtin_amdgcn_readfirstlane(rowslots) != cur) chg |= 1ull;
            RC_PF(pf_setup += __builtin_amdgcn_s_memtime() - pf_u0;)
            for (int a = h0; a < h1; a += 8) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {                             // the group's four row pairs
                    const int sl = q % NP, ar = a + 2 * q;
                    RC_PF({ const long long w0 = __builtin_amdgcn_s_memtime(); if (NP == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); pf_wait += __builtin_amdgcn_s_memtime() - w0; pf_tiles += 1; })
                    ll2 x[2], y[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) { x[u] = S2Raw<PACK>::unpack(d[sl][u]); if (!DERIVED) y[u] = l[sl][u]; }
                    // the pair NP pairs ahead goes into the registers this one leaves: of this unit, or of the wave's next unit
                    if (ar + 2 * NP < a1) issue_pair(sl);
                    else if (have_next) {
                        if (ar + 2 * NP == a1) set_next(Un.y, Un.x);     // the first rows of the wave's next unit
                        issue_pair(sl);
                        if (q == 3) { ncs0 = slot[min(Un.x + 2 * lane, V.n - 1)]; ncs1 = slot[min(Un.x + 2 * lane + 1, V.n - 1)]; }
                    }
                    RC_PF(const long long pl0 = __builtin_amdgcn_s_memtime();)
                    if (RC_S2_EXP & 2) {
                        y[0] = x[0]; y[1] = x[1];
                    } else if (DERIVED) {                                 // RC_S2_LOGS logs together: table reads first, then the arithmetic
#pragma unroll
                        for (int g_ = 0; g_ < 4; g_ += RC_S2_LOGS) {
                            QlogPrep pp[RC_S2_LOGS];
                            double2 tv[RC_S2_LOGS];
#pragma unroll
                            for (int u = 0; u < RC_S2_LOGS; ++u) pp[u] = rc_qlog_prep(((g_ + u) & 1) ? x[(g_ + u) >> 1].y : x[(g_ + u) >> 1].x, qeD);
#pragma unroll
                            for (int u = 0; u < RC_S2_LOGS; ++u) {
                                if (RC_S2_EXP & 16) tv[u] = tab[RC_S2_TABREP * ((lane & 15) + 16 * ((ar + u + g_) & 7))];   // timing experiment: a conflict-free gather
                                else if (RC_S2_TABREP == 1) tv[u] = rc_qlog_entry(tab, pp[u]);
                                else tv[u] = tab[RC_S2_TABREP * pp[u].j + (RC_S2_TABREP > 1 ? (lane & (RC_S2_TABREP - 1)) : 0)];
                            }
#pragma unroll
                            for (int u = 0; u < RC_S2_LOGS; ++u) {
                                const long long v = rc_qlog_raw(pp[u], tv[u], qsL, qthird, qmq);
                                if ((g_ + u) & 1) y[(g_ + u) >> 1].y = v; else y[(g_ + u) >> 1].x = v;
                            }
                        }
                    }
                    if (diag) {   // strictly upper triangle: element (row, col) lives iff col > row (the logs of dead entries are garbage: masked too)
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            if (!(colx > ar + u && colx < V.n)) { x[u].x = 0; y[u].x = 0; }
                            if (!(coly > ar + u && coly < V.n)) { x[u].y = 0; y[u].y = 0; }
                        }
                    }
                    RC_PF(asm volatile("" ::: "memory"); const long long pw0 = __builtin_amdgcn_s_memtime(); pf_log += pw0 - pl0;)
                    // direction 2: one pre-added value per lane, row and matrix into the wave's buffer
                    if (!(RC_S2_EXP & 4)) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            long long vd = x[u].x + x[u].y, vl = y[u].x + y[u].y;
                            if (any_donor) {
                                const long long gd = wave_from_prev64(donor ? x[u].y : 0), gl = wave_from_prev64(donor ? y[u].y : 0);
                                vd = (donor ? x[u].x : vd) + gd; vl = (donor ? y[u].x : vl) + gl;
                            }
                            if (any_odd && !whole) { vd = 0; vl = 0; }
                            pD[(2 * q + u) * RC_S2_PITCH + lane] = vd;
                            pL[(2 * q + u) * RC_S2_PITCH + lane] = vl;
                        }
                    }
                    if (any_odd) {   // uniform, rare: the elements of the lanes that are not whole
                        if (!whole) {
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const int row = ar + u;
                                add64(SD + (size_t)cs0 * ld + row, x[u].x); add64(SL + (size_t)cs0 * ld + row, y[u].x);
                                add64(SD + (size_t)cs1 * ld + row, x[u].y); add64(SL + (size_t)cs1 * ld + row, y[u].y);
                            }
                        }
                    }
                    RC_PF(const long long pd0 = __builtin_amdgcn_s_memtime(); pf_ldsw += pd0 - pw0;)
                    // direction 1
                    if (RC_S2_EXP & 8) { aD0 += x[0].x ^ x[1].y; aL0 += y[0].x ^ y[1].y; } else if (((chg >> (ar - h0)) & 3ull) == 0) {
                        aD0 += x[0].x + x[1].x; aD1 += x[0].y + x[1].y; aL0 += y[0].x + y[1].x; aL1 += y[0].y + y[1].y;
                    } else {
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int sr = __builtin_amdgcn_readlane(rowslots, ar + u - h0);
                            if (sr != cur) { flush1(); cur = sr; }
                            aD0 += x[u].x; aD1 += x[u].y; aL0 += y[u].x; aL1 += y[u].y;
                        }
                    }
                    RC_PF(asm volatile("" ::: "memory"); pf_d1 += __builtin_amdgcn_s_memtime() - pd0;)
                }
                // direction 2: transposed read of the 8-row group (same-wave LDS operations execute in order)
                if (RC_S2_EXP & 4) continue;
                RC_PF(const long long pr0 = __builtin_amdgcn_s_memtime();)
                __builtin_amdgcn_wave_barrier();
                long long vD[8], vL[8];
                const int jsw = (RC_S2_SWZ && (tq & 4)) ? 2 : 0;       // (see RC_S2_PITCH: the upper half of an octet row reads its pieces in the order 1, 0, 3, 2)
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const ll2 qd = *(const ll2 *)&pD[tr * RC_S2_PITCH + 8 * tq + (j ^ jsw)], ql = *(const ll2 *)&pL[tr * RC_S2_PITCH + 8 * tq + (j ^ jsw)];
                    vD[j] = qd.x; vD[j + 1] = qd.y; vL[j] = ql.x; vL[j + 1] = ql.y;
                }
                __builtin_amdgcn_wave_barrier();
                const bool mine = tq == ((a - h0) >> 3);
                if (!hasB) {
                    long long sD = ((vD[0] + vD[1]) + (vD[2] + vD[3])) + ((vD[4] + vD[5]) + (vD[6] + vD[7]));
                    long long sL = ((vL[0] + vL[1]) + (vL[2] + vL[3])) + ((vL[4] + vL[5]) + (vL[6] + vL[7]));
                    row8_sum2_dpp(sD, sL);
                    if (mine) { rDA += sD; rLA += sL; }
                } else {
                    unsigned mB = (unsigned)(bB >> (8 * tq)) & 0xFFu;
                    if (jsw) mB = ((mB & 0x33u) << 2) | ((mB >> 2) & 0x33u);   // value j of this lane came from source lane 8 tq + (j ^ 2)
                    long long sDA = 0, sLA = 0, sDB = 0, sLB = 0;
                    if (mB == 0) {
                        sDA = ((vD[0] + vD[1]) + (vD[2] + vD[3])) + ((vD[4] + vD[5]) + (vD[6] + vD[7]));
                        sLA = ((vL[0] + vL[1]) + (vL[2] + vL[3])) + ((vL[4] + vL[5]) + (vL[6] + vL[7]));
                    } else if (mB == 0xffu) {
                        sDB = ((vD[0] + vD[1]) + (vD[2] + vD[3])) + ((vD[4] + vD[5]) + (vD[6] + vD[7]));
                        sLB = ((vL[0] + vL[1]) + (vL[2] + vL[3])) + ((vL[4] + vL[5]) + (vL[6] + vL[7]));
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            if ((mB >> j) & 1) { sDB += vD[j]; sLB += vL[j]; } else { sDA += vD[j]; sLA += vL[j]; }
                        }
                    }
                    row8_sum2_dpp(sDA, sLA);
                    row8_sum2_dpp(sDB, sLB);
                    if (mine) { rDA += sDA; rLA += sLA; rDB += sDB; rLB += sLB; }
                }
                RC_PF(asm volatile("" ::: "memory"); pf_d2 += __builtin_amdgcn_s_memtime() - pr0;)
            }
            RC_PF(pf_u0 = __builtin_amdgcn_s_memtime();)
            // direction 2 write-out: lane l holds the totals of row h0 + 8 (l & 7) + (l >> 3); through the bounce lane l gets row h0 + l
            {
                const int mine_ = 8 * tq + tr, row = h0 + lane;
                __builtin_amdgcn_wave_barrier();
                bounce[mine_] = rDA; bounce[64 + mine_] = rLA;
                __builtin_amdgcn_wave_barrier();
                const long long tD = bounce[lane], tL = bounce[64 + lane];
                __builtin_amdgcn_wave_barrier();
                if (row < h1) {
                    if (tD) add64(SD + (size_t)dsA * ld + row, tD);
                    if (tL) add64(SL + (size_t)dsA * ld + row, tL);
                }
                if (hasB) {
                    bounce[mine_] = rDB; bounce[64 + mine_] = rLB;
                    __builtin_amdgcn_wave_barrier();
                    const long long uD = bounce[lane], uL = bounce[64 + lane];
                    __builtin_amdgcn_wave_barrier();
                    if (row < h1) {
                        if (uD) add64(SD + (size_t)dsB * ld + row, uD);
                        if (uL) add64(SL + (size_t)dsB * ld + row, uL);
                    }
                }
            }
            rDA = rLA = rDB = rLB = 0;
        }
        flush1();
        // diagonal (S includes j = i): D[a][a] -> S[slot_a][a], added by the column block's first unit; logD's diagonal is 0
        if (a0 == 0) {
            const int col0 = c0 + 2 * lane;
            const long long x0 = col0 < V.n ? V.diagq[col0] : 0, x1 = col0 + 1 < V.n ? V.diagq[col0 + 1] : 0;
            if (x0) add64(SD + (size_t)cs0 * ld + col0, x0);
            if (x1) add64(SD + (size_t)cs1 * ld + col0 + 1, x1);
        }
        unit = nunit; U = Un; cs0 = ncs0; cs1 = ncs1;
        RC_PF(pf_setup += __builtin_amdgcn_s_memtime() - pf_u0;)
    }
    RC_PF(if (lane == 0 && pf_out) { pf_out[0] = __builtin_amdgcn_s_memtime() - pf_t0; pf_out[1] = pf_wait; pf_out[2] = pf_tiles; pf_out[3] = pf_setup;
                                     pf_out[4] = __builtin_amdgcn_s_memrealtime() - pf_r0; pf_out[5] = pf_log; pf_out[6] = pf_d2;
                                     pf_out[7] = (pf_r0 << 20) | (long long)((__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xF) << 16) |
                                                 (long long)(__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (15 << 11)) & 0xFFFF);
                                     pf_out[8] = pf_issue; pf_out[9] = pf_ldsw; pf_out[10] = pf_d1; })
}

// 36.8 KiB of LDS per block: [log table 2 KiB][4 waves x 8.25 KiB of p-buffers][4 x 1 KiB bounces].  Four blocks per CU stay
// possible, which is what makes the compiler hold the kernel to 128 registers (three of these waves and one resolver wave per SIMD
// fill its 512).  Nothing in this kernel may spill: a scratch reload waits on vmcnt, the counter the prefetched rows use, i.e. for
// every row in flight (one spilled class mask made each 8-row group of a two-cluster unit wait for all its loads: 5,400 cycles
// instead of 730) — which is why the ragged units have their own launch (k_bulk_syml_list) instead of sharing this kernel's
// register allocation.
template <bool DERIVED, bool PACK>
__global__ __launch_bounds__(256, RC_SYML_MINWAVES) RC_S2_VGPR_CAP void k_bulk_syml2(View V, int wgen, int sgen, int cgen)
{
    constexpr int TABLL = 256 * RC_S2_TABREP;        // long longs of the log table (128 entries x 16 B x copies)
    __shared__ __attribute__((aligned(16))) long long lds[TABLL + 4 * (2 * 8 * RC_S2_PITCH + 128)];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int w = (int)blockIdx.x * 4 + wv;
    double2 *tab = (double2 *)lds;
    if (DERIVED)
        for (int q = threadIdx.x; q < 128 * RC_S2_TABREP; q += 256) tab[q] = V.ltab[q / RC_S2_TABREP];
    __syncthreads();
    long long *pf = nullptr;
    RC_PF(if (w < 8192 - 256) pf = (long long *)((char *)V.work[cgen] + 64) + (size_t)w * 16;)
    syml2_fast<DERIVED, PACK>(V, lds + TABLL + (size_t)wv * (2 * 8 * RC_S2_PITCH), lds + TABLL + 4 * (2 * 8 * RC_S2_PITCH) + (size_t)wv * 128, tab, wgen, sgen, V.wfast[w], V.wfast[w + 1], pf);
    (void)cgen;
}

// the units k_bulk_syml2's fast path does not take (a ragged last column block: n not a multiple of 128), by the round-2 unit code,
// dealt round-robin; launched behind it on the same stream when there are any (both add into S with exact integer atomics)
template <bool DERIVED>
__global__ __launch_bounds__(256, RC_SYML_MINWAVES) void k_bulk_syml_list(View V, int wgen, int sgen)
{
    __shared__ __attribute__((aligned(16))) long long tl[4][2][RC_SL_R][RC_SL_P];
    if (DERIVED) syml_load_table(V, tl);
    __syncthreads();
    const int wv = (int)(threadIdx.x >> 6);
    syml_units<DERIVED, long long>(V, tl[wv], wgen, sgen, V.nslow, 0, 8, 8, (int)blockIdx.x * 4 + wv, (int)gridDim.x * 4, nullptr, V.uslow);
}

// k_bulk_sym32 — the same symmetric reduction for 32-bit storage: TR-row × 256-column int32 tiles (1 KiB row
// segments), all 256 threads take a column in direction 1 and then a (row, matrix, part of a 64-column quarter) in direction 2.
// TR = 16 (default since round 4): 34 KiB blocks, three per CU — blocks in three different phases of (wait for the tile, write it to
// LDS, barrier, reduce, barrier) instead of two, and at ≤ 128 registers a resolver wave fits beside three of its waves on a SIMD (four
// blocks per CU read 5 % faster alone, but the resolver then waits for the reduction to retire: fewer sweeps per second).  A piece's
// load for the next tile is issued right behind its LDS write, and every thread derives the rows' chunk uniformity itself (two barriers
// per tile instead of three).  TR = 32, two 67 KiB blocks per CU, is the round-1 shape (diag builds: RC_SYM32_TR=32).
#define RC_SYM32_TC 256
#define RC_SYM32_TP (RC_SYM32_TC + 4)
#ifdef RC_DIAG   // timing ablations (wrong sums): debug flag 32 drops the direction-1 flushes, 64 the direction-2 flushes
#define RC_SYM32_ABL1 && !(zgen & 32)
#define RC_SYM32_ABL2 && !(zgen & 64)
#else
#define RC_SYM32_ABL1
#define RC_SYM32_ABL2
#endif
template <int TR>
__global__ __launch_bounds__(256, TR == 16 ? 4 : 2) void k_bulk_sym32(View V, int wgen, int zgen, int sgen, int cgen, int item_tiles, int nitems)
{
    typedef int i4 __attribute__((ext_vector_type(4)));
    constexpr int NQ = TR / 4;            // 16-byte pieces per thread, matrix and tile
    constexpr int PARTS = 32 / TR;        // direction 2: lanes of a wave = TR rows × 2 matrices × PARTS parts of every 8-column chunk
    __shared__ __attribute__((aligned(16))) int tt[2][TR][RC_SYM32_TP];
    __shared__ int item_sh, J_sh;
    __shared__ int cslot[RC_SYM32_TC];
    __shared__ __attribute__((aligned(16))) int rslot[TR];
    __shared__ int cchk[RC_SYM32_TC / 8];
    const int tid = threadIdx.x;
    const size_t ld = (size_t)V.ld;
    (void)zgen;  // generations are cleared and work counters re-armed by k_resolve (SweepArgs.zero_gen); diag builds: ablation flags
    const int *__restrict__ Dq = (const int *)V.Dq;
    const int *__restrict__ Lq = (const int *)V.Lq;
    const int *__restrict__ slot = V.snap[sgen];
    long long *SD = V.SD[wgen], *SL = V.SL[wgen];
    const int n = V.n;
    const int ncb = (n + RC_SYM32_TC - 1) / RC_SYM32_TC;
    int *counter = V.work[cgen];
    for (;;) {
        __syncthreads();
        if (tid == 0) {
            int item = atomicAdd(counter, 1);
            int J = -1;
            if (item < nitems) {
                for (J = ncb - 1;; --J) {
                    const int ntile = (min(RC_SYM32_TC * J + RC_SYM32_TC, n) + TR - 1) / TR;
                    const int cnt = (ntile + item_tiles - 1) / item_tiles;
                    if (item < cnt) break;
                    item -= cnt;
                }
            }
            item_sh = item; J_sh = J;
        }
        __syncthreads();
        const int J = J_sh, item = item_sh;
        if (J < 0) break;
        const int c0 = J * RC_SYM32_TC;
        const int ntile = (min(c0 + RC_SYM32_TC, n) + TR - 1) / TR;
        const int t_begin = item * item_tiles, t_end = min(ntile, t_begin + item_tiles);
        cslot[tid] = (c0 + tid < n) ? slot[c0 + tid] : -1;
        __syncthreads();
        if (tid < RC_SYM32_TC / 8) {
            const int s0 = cslot[tid * 8];
            bool u = true;
            for (int q = 1; q < 8; ++q) u = u && (cslot[tid * 8 + q] == s0);
            cchk[tid] = u ? s0 : -2;
        }
        i4 d[NQ], l[NQ];
        auto issue1 = [&](int t, int q) {
            const int piece = q * 256 + tid, lr = piece >> 6, lp = piece & 63;
            const int r = min(t * TR + lr, n - 1);
            d[q] = __builtin_nontemporal_load((const i4 *)(Dq + (size_t)r * ld + c0 + lp * 4));
            l[q] = __builtin_nontemporal_load((const i4 *)(Lq + (size_t)r * ld + c0 + lp * 4));
        };
#pragma unroll
        for (int q = 0; q < NQ; ++q) issue1(t_begin, q);
        long long accD = 0, accL = 0;
        int cur = -1;
        const int b = c0 + tid;
        for (int t = t_begin; t < t_end; ++t) {
            const int r0 = t * TR;
            __syncthreads();
            if (tid < TR) rslot[tid] = (r0 + tid < n) ? slot[r0 + tid] : -1;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int piece = q * 256 + tid, lr = piece >> 6, lp = piece & 63;
                const int r = r0 + lr, bb = c0 + lp * 4;
                const bool live = r < n;
                i4 x = d[q], y = l[q];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (!(live && bb + e > r)) { x[e] = 0; y[e] = 0; }
                *(i4 *)&tt[0][lr][lp * 4] = x;
                *(i4 *)&tt[1][lr][lp * 4] = y;
                if (t + 1 < t_end) issue1(t + 1, q);   // (the piece's registers are free: its load for the next tile goes out at once)
            }
            __syncthreads();
            // direction 1: column b gathers the rows of the tile
#pragma unroll 1
            for (int ch = 0; ch < TR / 8; ++ch) {
                int xd[8], xl[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { xd[q] = tt[0][ch * 8 + q][tid]; xl[q] = tt[1][ch * 8 + q][tid]; }
                int cs_;
                {
                    const i4 ra = *(const i4 *)&rslot[ch * 8], rb = *(const i4 *)&rslot[ch * 8 + 4];
                    const bool u = ra[1] == ra[0] && ra[2] == ra[0] && ra[3] == ra[0] && rb[0] == ra[0] && rb[1] == ra[0] && rb[2] == ra[0] && rb[3] == ra[0];
                    cs_ = __builtin_amdgcn_readfirstlane(u ? ra[0] : -2);
                }
                if (cs_ != -2) {
                    if (cs_ != cur) {
                        if (cur >= 0) {
                            if (accD RC_SYM32_ABL1) __hip_atomic_fetch_add((u64 *)(SD + (size_t)cur * ld + b), (u64)accD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (accL RC_SYM32_ABL1) __hip_atomic_fetch_add((u64 *)(SL + (size_t)cur * ld + b), (u64)accL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        accD = accL = 0; cur = cs_;
                    }
                    accD += (((long long)xd[0] + xd[1]) + ((long long)xd[2] + xd[3])) + (((long long)xd[4] + xd[5]) + ((long long)xd[6] + xd[7]));
                    accL += (((long long)xl[0] + xl[1]) + ((long long)xl[2] + xl[3])) + (((long long)xl[4] + xl[5]) + ((long long)xl[6] + xl[7]));
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int sr = __builtin_amdgcn_readfirstlane(rslot[ch * 8 + q]);
                        if (sr != cur) {
                            if (cur >= 0) {
                                if (accD RC_SYM32_ABL1) __hip_atomic_fetch_add((u64 *)(SD + (size_t)cur * ld + b), (u64)accD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (accL RC_SYM32_ABL1) __hip_atomic_fetch_add((u64 *)(SL + (size_t)cur * ld + b), (u64)accL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            accD = accL = 0; cur = sr;
                        }
                        accD += xd[q]; accL += xl[q];
                    }
                }
            }
            // direction 2: wave w -> columns 64w..64w+63; a lane = (row, matrix, part): with TR = 32 it reads whole 8-column chunks, with
            // TR = 16 the two parts of a lane pair (lane, lane ^ 32) read four columns each and are added up when the sum is flushed
            {
                const int quarter = tid >> 6, lane = tid & 63, r = lane & (TR - 1), mat = (lane / TR) & 1, part = lane / (2 * TR);
                long long *S = mat ? SL : SD;
                const int arow = r0 + r;
                long long acc = 0;
                int cc = -1;
                auto flush = [&]() {   // (cc is uniform over the wave, so is the control flow)
                    if (PARTS == 2) acc += __shfl_xor(acc, 32);
                    if (cc >= 0 && acc && arow < n && part == 0 RC_SYM32_ABL2) __hip_atomic_fetch_add((u64 *)(S + (size_t)cc * ld + arow), (u64)acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                };
#pragma unroll 2
                for (int ch = 0; ch < 8; ++ch) {
                    const int cb = quarter * 64 + ch * 8;
                    i4 x0, x1;
                    if (PARTS == 1) { x0 = *(const i4 *)&tt[mat][r][cb]; x1 = *(const i4 *)&tt[mat][r][cb + 4]; }
                    else { x0 = *(const i4 *)&tt[mat][r][cb + part * 4]; x1 = i4{0, 0, 0, 0}; }
                    const int cs_ = __builtin_amdgcn_readfirstlane(cchk[cb >> 3]);
                    if (cs_ != -2) {
                        if (cs_ != cc) {
                            flush();
                            acc = 0; cc = cs_;
                        }
                        if (PARTS == 1) acc += (((long long)x0[0] + x0[1]) + ((long long)x0[2] + x0[3])) + (((long long)x1[0] + x1[1]) + ((long long)x1[2] + x1[3]));
                        else acc += ((long long)x0[0] + x0[1]) + ((long long)x0[2] + x0[3]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int sc = __builtin_amdgcn_readfirstlane(cslot[cb + q]);
                            if (sc != cc) {
                                flush();
                                acc = 0; cc = sc;
                            }
                            if (PARTS == 1) acc += (q < 4) ? x0[q & 3] : x1[q & 3];
                            else acc += (part == (q >> 2)) ? x0[q & 3] : 0;
                        }
                    }
                }
                flush();
            }
        }
        if (cur >= 0) {
            if (accD RC_SYM32_ABL1) __hip_atomic_fetch_add((u64 *)(SD + (size_t)cur * ld + b), (u64)accD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (accL RC_SYM32_ABL1) __hip_atomic_fetch_add((u64 *)(SL + (size_t)cur * ld + b), (u64)accL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (t_begin == 0 && b < n) {
            const long long x = V.diagq[b];
            if (x) __hip_atomic_fetch_add((u64 *)(SD + (size_t)slot[b] * ld + b), (u64)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Slot tables replicated in LDS by every block of k_resolve.
// ---------------------------------------------------------------------------------------------------
struct Tab {
    int *size;       // [kcap]
    int *label;      // [kcap] 0 = free
    short *act;      // [kcap] the active slots (the candidates of mcmc.jl:195; in slot order — the order does not matter, see tab_structural)
    double *base_o;  // [kcap] A[s] + log p + log(s-1+r), s = size          (candidate cluster of another point)
    double *base_s;  // [kcap] same with s = size-1                          (the point's own cluster, itself removed)
    double *base_p;  // [kcap + 1] same with s = size+1 (a cluster a batch entry joined: the usual size of a touched candidate under validation —
                     //  without it every such candidate loads A[s] from global memory and takes a log); [kcap] = the constant of a cluster of one (births)
    unsigned *used;  // [(n+31)/32] label occupancy bitset, bit (label-1); built when a round has changers (LDS, or V.used_scratch for large n)
    double *red_v;   // [NW][32] reduction scratch (NW = waves per block)
    int *red_pos, *red_slot;
    int *misc;       // [0]=K [1]=smallest_empty [2]=scratch [3]=nb [4]=hi [5]=fail [6]=barrier ok [7]=slot_hi [8]=#births [9]=#effective [13]=visited [14]=#clean slots (tab_partition) [15]=#dirty [16]=hot slot of the batch (-1: none) [17]=scratch
    u64 *blk_key;    // block-local minimum (first violation)
    // batch of tentative changers of the current round (identical in every block), ascending in point index
    int *bx, *bu;             // [RC_MAXB] point (original index), its internal index
    short *ba, *bb;           // [RC_MAXB] source slot; target slot (after batch_sim: the slot it really goes to)
    int *blab;                // [RC_MAXB] label the entry takes (birth / rename), 0 = none
    short *bK;                // [RC_MAXB] number of clusters after the entry
    unsigned char *bflag;     // [RC_MAXB] RC_BF_* bits
    short *birth;             // [RC_MAXB] the entries that create a cluster, ascending
    short *pairs;             // [2·RC_MAXB] the entries that touch each slot, grouped by slot, ascending within a slot
    short *pairs_tmp;         // [2·RC_MAXB] the same groups before they are ordered
    int *seg;                 // [kcap+1] before / during batch_sim: entries leaving the slot; afterwards seg[k] = end of slot k's group
    unsigned char *joined;    // [kcap] some entry of the batch moves a point INTO the slot
    unsigned char *candie;    // [kcap] the slot could become empty inside the batch (size − leavers < 1): its size is simulated
    short *act2;              // [kcap] the active slots again, those whose cached scores are valid first (tab_partition)
    unsigned char *dirty;     // [kcap] a committed change of this sweep touched the slot: its cached scores are void
    unsigned short *ccnt;     // [nchunks + 1] scratch: changers per chunk / exclusive offsets, saturating at 65535 (only offsets <= batch capacity matter)
    int *cu;                  // [RC_CPB_LDS * RC_PTS] internal index of the points of this block's chunks (pi[i]: fixed for the launch) ...
    short *cown;              // [RC_CPB_LDS * RC_PTS] ... and their slots (kept current by commit_batch); used when `cached` (at most RC_CPB_LDS chunks per block):
    double2 *flt;             // [128] LDS copy of the table of rc_flog
    int cached;               // the first loads of every pass over a chunk — pi[i], then slot_of[pi[i]]: two dependent global round trips — come from LDS
};
#define RC_CPB_LDS 4
#define RC_BF_DEATH 1   // the source cluster becomes empty
#define RC_BF_BIRTH 2   // the target is a new cluster (slot bb, label blab)
#define RC_BF_RENAME 4  // a singleton that takes a fresh, smaller label (slot unchanged, label blab)
#define RC_BF_NOOP 8    // a singleton that draws "new cluster" and keeps its label: nothing changes
#define RC_BF_STAY 16   // (with NOOP) a carried-over guess whose target cluster no longer exists: the entry is a placeholder, the guess is "stays"

#define RC_A16(x) (((x) + 15) & ~(size_t)15)
#define RC_TAB_NOFF 30
__host__ __device__ inline size_t tab_layout(int kcap, int n, int nw, size_t *off /*RC_TAB_NOFF*/, int maxb = RC_MAXB)
{
    size_t o = 0;
    off[0] = o; o = RC_A16(o + sizeof(double) * kcap);          // base_o
    off[1] = o; o = RC_A16(o + sizeof(double) * kcap);          // base_s
    off[2] = o; o = RC_A16(o + sizeof(double) * nw * RC_PTS);   // red_v
    off[3] = o; o = RC_A16(o + sizeof(u64));                    // blk_key
    off[4] = o; o = RC_A16(o + sizeof(int) * kcap);             // size
    off[5] = o; o = RC_A16(o + sizeof(int) * kcap);             // label
    off[6] = o; o = RC_A16(o + sizeof(int) * nw * RC_PTS);      // red_pos
    off[7] = o; o = RC_A16(o + sizeof(int) * nw * RC_PTS);      // red_slot
    off[8] = o; o = RC_A16(o + (n <= RC_USED_LDS_MAX_N ? sizeof(unsigned) * ((n + 31) / 32) : 0));  // used (beyond: per-block global scratch)
    off[9] = o; o = RC_A16(o + sizeof(int) * 24);               // misc
    off[10] = o; o = RC_A16(o + sizeof(short) * kcap);         // act2
    off[11] = o; o = RC_A16(o + sizeof(short) * kcap);          // act
    off[12] = o; o = RC_A16(o + sizeof(int) * maxb);         // bx
    off[13] = o; o = RC_A16(o + sizeof(short) * maxb);       // ba
    off[14] = o; o = RC_A16(o + sizeof(short) * maxb);       // bb
    off[15] = o; o = RC_A16(o + sizeof(int) * (kcap + 1));      // seg
    off[16] = o; o = RC_A16(o + sizeof(unsigned short) * ((n + RC_PTS - 1) / RC_PTS + 1));  // ccnt
    off[17] = o; o = RC_A16(o + sizeof(int) * maxb);         // bu
    off[18] = o; o = RC_A16(o + sizeof(int) * maxb);         // blab
    off[19] = o; o = RC_A16(o + (size_t)kcap);                  // dirty
    off[20] = o; o = RC_A16(o + sizeof(short) * maxb);       // bK
    off[21] = o; o = RC_A16(o + (size_t)maxb);               // bflag
    off[22] = o; o = RC_A16(o + sizeof(short) * maxb);       // birth
    off[23] = o; o = RC_A16(o + sizeof(short) * 2 * maxb);   // pairs
    off[24] = o; o = RC_A16(o + (size_t)kcap);                  // candie
    off[25] = o; o = RC_A16(o + (kcap < 2048 ? (sizeof(int) + sizeof(short)) * RC_CPB_LDS * RC_PTS : 0));   // cu, cown (only below 2048 slots)
    off[26] = o; o = RC_A16(o + (size_t)kcap);                  // joined
    off[27] = o; o = RC_A16(o + sizeof(short) * 2 * maxb);   // pairs_tmp
    off[28] = o; o = RC_A16(o + 128 * sizeof(double2));       // flt: the table of rc_flog (the score logarithms gather from it per lane)
    off[29] = o; o = RC_A16(o + (kcap < 2048 ? sizeof(double) * (kcap + 1) : 0));   // base_p (only below 2048 slots: the largest tables fill the CU as they are)
    return o;
}

__device__ __forceinline__ Tab tab_carve(char *smem, int kcap, int n, int nw, int maxb = RC_MAXB)
{
    size_t off[RC_TAB_NOFF];
    tab_layout(kcap, n, nw, off, maxb);
    Tab T;
    T.base_o = (double *)(smem + off[0]); T.base_s = (double *)(smem + off[1]); T.red_v = (double *)(smem + off[2]);
    T.blk_key = (u64 *)(smem + off[3]); T.size = (int *)(smem + off[4]); T.label = (int *)(smem + off[5]);
    T.red_pos = (int *)(smem + off[6]); T.red_slot = (int *)(smem + off[7]); T.used = (unsigned *)(smem + off[8]);
    T.misc = (int *)(smem + off[9]); T.act = (short *)(smem + off[11]);
    T.bx = (int *)(smem + off[12]); T.ba = (short *)(smem + off[13]); T.bb = (short *)(smem + off[14]);
    T.seg = (int *)(smem + off[15]); T.ccnt = (unsigned short *)(smem + off[16]); T.bu = (int *)(smem + off[17]);
    T.blab = (int *)(smem + off[18]); T.bK = (short *)(smem + off[20]);
    T.bflag = (unsigned char *)(smem + off[21]); T.birth = (short *)(smem + off[22]);
    T.pairs = (short *)(smem + off[23]); T.candie = (unsigned char *)(smem + off[24]);
    T.joined = (unsigned char *)(smem + off[26]); T.pairs_tmp = (short *)(smem + off[27]);
    T.act2 = (short *)(smem + off[10]); T.dirty = (unsigned char *)(smem + off[19]);
    T.cu = (int *)(smem + off[25]); T.cown = (short *)(T.cu + RC_CPB_LDS * RC_PTS); T.cached = 0;
    T.flt = (double2 *)(smem + off[28]); T.base_p = (double *)(smem + off[29]);
    return T;
}

__device__ __forceinline__ size_t tab_bytes_dev(int kcap, int n, int nw, int maxb = RC_MAXB)
{
    size_t off[RC_TAB_NOFF];
    return tab_layout(kcap, n, nw, off, maxb);
}

static size_t tab_bytes(int kcap, int n, int nw, int maxb = RC_MAXB)
{
    size_t off[RC_TAB_NOFF];
    return tab_layout(kcap, n, nw, off, maxb);
}

__device__ __forceinline__ double tab_base(const View &V, const SweepArgs &a, int s)
{
    // A[s] + (log p + log(s - 1 + r)): size-only terms of mcmc.jl:223-226,240-241
    return V.A[s] + (a.logp + log((double)s - 1.0 + a.r));
}

// per-slot score constants of the active slots.  All threads; ends synchronised.
__device__ __forceinline__ void tab_bases(const View &V, const SweepArgs &a, Tab &T)
{
    for (int k = threadIdx.x; k < T.misc[7]; k += blockDim.x) {
        const int s = T.size[k];
        if (T.label[k] > 0) {
            T.base_o[k] = tab_base(V, a, s);
            T.base_s[k] = (s >= 2) ? tab_base(V, a, s - 1) : 0.0;
            if (V.kcap < 2048) T.base_p[k] = (s + 1 <= V.n) ? tab_base(V, a, s + 1) : 0.0;
        }
    }
    __syncthreads();
}

// After a birth / death / rename: label bitset, smallest empty label, candidate ranks.  Ends synchronised.
__device__ __forceinline__ void tab_structural(const View &V, Tab &T)
{
    const int nw = (V.n + 31) / 32, hi = T.misc[7];
    for (int w = threadIdx.x; w < nw; w += blockDim.x) T.used[w] = 0u;
    if (threadIdx.x == 0) T.misc[2] = V.n + 1;
    __syncthreads();
    for (int k = threadIdx.x; k < hi; k += blockDim.x) {
        const int lab = T.label[k];
        if (lab > 0) atomicOr(&T.used[(lab - 1) >> 5], 1u << ((lab - 1) & 31));
    }
    __syncthreads();
    for (int w = threadIdx.x; w < nw; w += blockDim.x) {
        const unsigned inv = ~T.used[w];
        if (inv) {
            const int lab = w * 32 + __ffs((int)inv);  // 1-based label
            if (lab <= V.n) atomicMin(&T.misc[2], lab);
        }
    }
    // the active slots, in slot order (ballot compaction; wave totals through the reduction scratch).  The candidate order is
    // immaterial: the noise is keyed by label and ties between scores go to the smaller label (eval_chunk), so the label-rank
    // order of round 1 — a pass over all slots per slot — is not needed.
    {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, NW = blockDim.x >> 6;
        int base = 0;
        for (int k0 = 0; k0 < hi; k0 += blockDim.x) {
            const int k = k0 + threadIdx.x;
            const bool f = k < hi && T.label[k] > 0;
            const u64 m = __ballot(f);
            if (lane == 0) T.red_pos[wave] = __popcll(m);
            __syncthreads();
            int before = 0, tot = 0;
            for (int w = 0; w < NW; ++w) { const int x = T.red_pos[w]; tot += x; before += (w < wave) ? x : 0; }
            if (f) T.act[base + before + __popcll(m & ((1ull << lane) - 1ull))] = (short)k;
            base += tot;
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) T.misc[1] = T.misc[2];
    __syncthreads();
}

// After a commit with births / deaths / renames, fewer than 2048 slots (commit_batch): the freed labels are already cleared in
// T.used; here the labels the first nc entries took, the smallest empty label, the list of active slots (ballot compaction in
// slot order, as tab_structural) and the per-slot score constants — two block barriers where rebuilding everything took six.
__device__ __forceinline__ void tab_after_commit(const View &V, const SweepArgs &a, Tab &T, int nc)
{
    const int nw = (V.n + 31) / 32, hi = T.misc[7];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, NW = blockDim.x >> 6;
    for (int q = threadIdx.x; q < nc; q += blockDim.x) {
        const int lab = T.blab[q];                                   // (0 unless the entry is a birth or a rename)
        if (lab > 0) atomicOr(&T.used[(lab - 1) >> 5], 1u << ((lab - 1) & 31));
    }
    if (threadIdx.x == 0) T.misc[1] = V.n + 1;
    for (int k = threadIdx.x; k < hi; k += blockDim.x) {            // score constants (tab_bases): sizes and labels are final
        const int s = T.size[k];
        if (T.label[k] > 0) {
            T.base_o[k] = tab_base(V, a, s);
            T.base_s[k] = (s >= 2) ? tab_base(V, a, s - 1) : 0.0;
            if (V.kcap < 2048) T.base_p[k] = (s + 1 <= V.n) ? tab_base(V, a, s + 1) : 0.0;
        }
    }
    int base = 0;
    for (int k0 = 0; k0 < hi; k0 += blockDim.x) {
        const int k = k0 + threadIdx.x;
        const bool f = k < hi && T.label[k] > 0;
        const u64 m = __ballot(f);
        if (lane == 0) T.red_pos[wave] = __popcll(m);
        __syncthreads();
        if (k0 == 0)
            for (int w = threadIdx.x; w < nw; w += blockDim.x) {   // smallest empty label (the bitset is final behind the barrier)
                const unsigned inv = ~T.used[w];
                if (inv) {
                    const int lab = w * 32 + __ffs((int)inv);
                    if (lab <= V.n) atomicMin(&T.misc[1], lab);
                }
            }
        int before = 0, tot = 0;
        for (int w = 0; w < NW; ++w) { const int x = T.red_pos[w]; tot += x; before += (w < wave) ? x : 0; }
        if (f) T.act[base + before + __popcll(m & ((1ull << lane) - 1ull))] = (short)k;
        base += tot;
        __syncthreads();
    }
}

__device__ __forceinline__ void tab_load(const View &V, Tab &T)
{
    for (int k = threadIdx.x; k < V.kcap; k += blockDim.x) {
        T.size[k] = V.slot_size[k];
        T.label[k] = V.slot_label[k];
        T.act[k] = V.slot_act[k];
    }
    if (threadIdx.x == 0) {
        T.misc[0] = V.sc->K;
        T.misc[1] = V.sc->smallest_empty;
        T.misc[7] = V.sc->slot_hi;
        *T.blk_key = RC_KEY_NONE;
    }
    __syncthreads();
}

__device__ __forceinline__ void tab_store(const View &V, const Tab &T)
{
    for (int k = threadIdx.x; k < V.kcap; k += blockDim.x) {
        V.slot_size[k] = T.size[k];
        V.slot_label[k] = T.label[k];
        V.slot_act[k] = T.act[k];
    }
    if (threadIdx.x == 0) {
        V.sc->K = T.misc[0];
        V.sc->smallest_empty = T.misc[1];
        V.sc->slot_hi = T.misc[7];
    }
}

// label snapshot of generation g and the run count of the labels in natural point order.  One block.
// the two halves of snapshot_labels for the resolver's epilogue: the copy by all blocks, the run count by one
__device__ __forceinline__ void snapshot_copy_grid(const View &V, int g, int G)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V.n; i += G * blockDim.x) V.snap[g][i] = V.slot_of[i];
}
__device__ __forceinline__ void snapshot_runs(const View &V, int *lds_cnt)
{
    if (threadIdx.x == 0) *lds_cnt = 0;
    __syncthreads();
    int mine = 0;
    for (int i = threadIdx.x; i < V.n; i += blockDim.x) mine += (i == 0) || (V.slot_of[i] != V.slot_of[i - 1]);
    atomicAdd(lds_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) { V.sc->runs = *lds_cnt; V.hsum->runs = *lds_cnt; }
    __syncthreads();
}

__device__ __forceinline__ void snapshot_labels(const View &V, int g, int *lds_cnt)
{
    if (threadIdx.x == 0) *lds_cnt = 0;
    __syncthreads();
    int mine = 0;
    for (int i = threadIdx.x; i < V.n; i += blockDim.x) {
        const int s = V.slot_of[i];
        V.snap[g][i] = s;
        mine += (i == 0) || (s != V.slot_of[i - 1]);
    }
    atomicAdd(lds_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) { V.sc->runs = *lds_cnt; V.hsum->runs = *lds_cnt; }
    __syncthreads();
}

// tables = false: the slot sizes and labels in the record are still right (a sweep that changed nothing) — 2·kcap stores to
// host memory less in the tail of block 0, which is the end of the launch in the stationary regime
__device__ __forceinline__ void write_summary(const View &V, int n_changes, int n_rounds, bool tables = true)
{
    if (tables)
        for (int k = threadIdx.x; k < V.kcap; k += blockDim.x) {
            V.hsum->size_label[2 * k] = V.slot_size[k];
            V.hsum->size_label[2 * k + 1] = V.slot_label[k];
        }
    if (threadIdx.x == 0) {
        V.hsum->K = V.sc->K;
        V.hsum->n_changes = n_changes;
        V.hsum->n_rounds = n_rounds;
        V.hsum->slot_hi = V.sc->slot_hi;
        V.hsum->err = __hip_atomic_load(&V.sc->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        V.hsum->seq += 1;
    }
}

// After rc_set_state: derive ranks / smallest empty label / perm (both generations) from slot_size, slot_label, slot_of.
// Exact S correction for a batch of label moves made outside a sweep (split–merge proposals, state restore):
// for every move (point x: slot a -> slot b) and every column i: S[a][i] -= x_row[i], S[b][i] += x_row[i], in each
// listed S generation.  256 threads x 2 columns; moves are applied in order, no atomics (the caller has drained
// both streams).
struct MoveList { const int *mv; int count; int gens[2]; int ngens; };
__global__ __launch_bounds__(256) void k_apply_moves(View V, MoveList ML)
{
    const int i = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (i >= V.ld) return;
    for (int m = 0; m < ML.count; ++m) {
        const int x = ML.mv[3 * m], a = ML.mv[3 * m + 1], b = ML.mv[3 * m + 2];
        if (a == b) continue;
        const size_t e = (size_t)x * V.ld + i;
        long long d0, d1, l0, l1;
        if (V.derived && !V.Lq) {
            const ll2 d = *(const ll2 *)((const long long *)V.Dq + e);
            d0 = d.x; d1 = d.y;
            l0 = rc_load_L(V, x, i, d0); l1 = rc_load_L(V, x, i + 1, d1);
        } else if (V.bits == 64) {
            const ll2 d = *(const ll2 *)((const long long *)V.Dq + e), l = *(const ll2 *)((const long long *)V.Lq + e);
            d0 = d.x; d1 = d.y; l0 = l.x; l1 = l.y;
        } else {
            const int2 d = *(const int2 *)((const int *)V.Dq + e), l = *(const int2 *)((const int *)V.Lq + e);
            d0 = d.x; d1 = d.y; l0 = l.x; l1 = l.y;
        }
        for (int g = 0; g < ML.ngens; ++g) {
            long long *SD = V.SD[ML.gens[g]], *SL = V.SL[ML.gens[g]];
            ll2 *pa = (ll2 *)(SD + (size_t)a * V.ld + i), *pb = (ll2 *)(SD + (size_t)b * V.ld + i);
            ll2 *qa = (ll2 *)(SL + (size_t)a * V.ld + i), *qb = (ll2 *)(SL + (size_t)b * V.ld + i);
            ll2 va = *pa, vb = *pb, wa = *qa, wb = *qb;
            va.x -= d0; va.y -= d1; vb.x += d0; vb.y += d1;
            wa.x -= l0; wa.y -= l1; wb.x += l0; wb.y += l1;
            *pa = va; *pb = vb; *qa = wa; *qb = wb;
        }
    }
}

__global__ __launch_bounds__(1024) void k_derive(View V, int rebuild_perm)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Tab T = tab_carve(smem, V.kcap, V.n, 1);
    if (V.n > RC_USED_LDS_MAX_N) T.used = V.used_scratch;   // one block
    for (int k = threadIdx.x; k < V.kcap; k += blockDim.x) { T.size[k] = V.slot_size[k]; T.label[k] = V.slot_label[k]; T.act[k] = 0; }
    if (threadIdx.x == 0) { T.misc[0] = V.sc->K; T.misc[7] = V.sc->slot_hi; }
    __syncthreads();
    tab_structural(V, T);
    tab_store(V, T);
    __syncthreads();
    snapshot_labels(V, 0, &T.misc[2]);
    for (int i = threadIdx.x; i < V.n; i += blockDim.x) V.snap[1][i] = V.snap[0][i];
    __syncthreads();
    write_summary(V, 0, 0);
    __syncthreads();
    if (!rebuild_perm) return;
    int *off = (int *)smem, *cur = off + V.kcap;
    build_perm_block(V, 0, off, cur);
    for (int p = threadIdx.x; p < V.n; p += blockDim.x) { V.perm[1][p] = V.perm[0][p]; V.pslot[1][p] = V.pslot[0][p]; }
}

// ---------------------------------------------------------------------------------------------------
// Scoring + Gumbel-max draw of one chunk of 32 points (src/mcmc.jl:192-252 for each point of the chunk,
// src/utils.jl:2-6 for the draw), under the state held in T.  Thread (pt, st): point pt of the chunk,
// candidate positions st, st+NS, ...  Points with index <= after_i are skipped (already final).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void best_merge(double &bv, int &bp, int &bs, double v, int p, int s)
{
    // first-index tie rule of argmax (utils.jl:5): larger value wins, equal values -> smaller candidate position
    if (s != -2 && (bs == -2 || v > bv || (v == bv && p < bp))) { bv = v; bp = p; bs = s; }
}

// Gumbel noise of a candidate: the uniform of (sweep, point, LABEL of the candidate cluster); key 0 is the "new cluster"
// candidate.  Keying by label rather than by position in the candidate list keeps the noise of every other candidate
// unchanged when a cluster is born, dies or takes a new label, so such a change invalidates the tentative draws of the
// later points no more than a plain move does (include/redclust_hip.h, oracle/rc_oracle.c: orc_uniform).
// mode 0 (tentative): decisions under the committed state; the target of every point goes to V.tent (-1 = new cluster),
//   changers to rec and the chunk's stamped mask word.  A singleton that draws "new cluster" is announced as a changer
//   too: whether it keeps its label or takes a smaller free one depends on the changers before it (batch_sim decides).
// mode 1 (validate): point i is evaluated under the committed state PLUS the first j batch changers that precede it:
//   exact integer corrections of the row sums and sizes of the clusters they touch, the clusters they create (singletons:
//   their row sums are matrix entries), the labels they change, the clusters they empty, the cluster count.  A decision
//   that differs from the tentative one is a violation (block-local minimum in T.blk_key).  Points outside (lo, hi] are
//   skipped.
#define RC_NEWKEY 0x7ffffffe   // order key of the new-cluster candidate: after every label (utils.jl:5 first-index rule)

// Net effect on slot k, as seen from point u, of the batch entries < limit that touch the slot (its group in T.pairs,
// ascending): the exact integer corrections ±Dq[x_q][u], ±Lq[x_q][u] of its two row sums, its size, its label after a
// relabelling.  The matrix entries are fetched four entries at a time — the loads of a group are independent of each other,
// but one load per loop turn costs a memory round trip per entry, and the changers of a batch can crowd into one cluster
// (first sweep from random labels: groups of 100+ entries, 115 µs of skew at the barrier while one thread per point walked them).
__device__ __forceinline__ bool batch_corr(const View &V, const Tab &T, int k, int u, int limit, long long &sd, long long &sl, int &sz, int &lab)
{
    bool touched = false;
    const size_t ld = (size_t)V.ld;
    int e = k ? T.seg[k - 1] : 0;
    const int e1 = T.seg[k];
    while (e < e1) {
        int qq[4], sgn[4], cnt = 0;
        for (; cnt < 4 && e < e1; ++e) {
            const int q = T.pairs[e];
            if (q >= limit) { e = e1; break; }
            const int qa = T.ba[q], qb = T.bb[q];
            if (qa != qb) { qq[cnt] = T.bu[q]; sgn[cnt] = (qb == k) - (qa == k); ++cnt; }
            else lab = T.blab[q];   // the singleton took a new label
        }
        long long xd[4], xl[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (t < cnt) {
                const size_t ee = (size_t)qq[t] * ld + u;
                xd[t] = (V.bits == 64) ? ((const long long *)V.Dq)[ee] : (long long)((const int *)V.Dq)[ee];
                xl[t] = (V.derived && !V.Lq) ? 0 : ((V.bits == 64) ? ((const long long *)V.Lq)[ee] : (long long)((const int *)V.Lq)[ee]);
            }
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (t < cnt) {
                if (V.derived && !V.Lq) xl[t] = rc_load_L(V, qq[t], u, xd[t]);
                sd += sgn[t] * xd[t]; sl += sgn[t] * xl[t]; sz += sgn[t];
                touched = true;
            }
    }
    return touched;
}

// The batch's largest group.  When the changers of a batch crowd into one cluster (first sweeps from a poor labelling: a hundred
// entries join the same cluster), that slot's group is walked — one matrix row entry per batch entry — by the ONE candidate stream
// that scores the slot for a point (validation) or owns its correction (commit), while the other streams of the point have a
// handful of entries each: 80 µs of validation per round on the four blocks that hold the batch, everyone else at the barrier.
// For that slot (T.misc[16], groups of RC_HOT_MIN entries or more) the NS streams of a point share the entries — stream st takes
// entries st, st + NS, ... before `limit` — and add their exact integer partial sums into the point's record in LDS
// (aliasing the argmax scratch T.red_v: 32 B per point — sum D, sum L, size change, contributing entries, new label).
#define RC_HOT_MIN 24
__device__ __forceinline__ void hot_accumulate(const View &V, const Tab &T, int h, int u, int limit, int st, int NS, long long *acc)
{
    const size_t ld = (size_t)V.ld;
    const int e1 = T.seg[h];
    long long sd = 0, sl = 0;
    int sz = 0, cnt = 0;
    for (int e = (h ? T.seg[h - 1] : 0) + st; e < e1; e += NS) {
        const int q = T.pairs[e];
        if (q >= limit) break;   // ascending within the group: the later entries of this stream are beyond the limit too
        const int qa = T.ba[q], qb = T.bb[q];
        if (qa != qb) {
            const int sgn = (qb == h) - (qa == h), x = T.bu[q];
            const size_t ee = (size_t)x * ld + u;
            const long long xd = (V.bits == 64) ? ((const long long *)V.Dq)[ee] : (long long)((const int *)V.Dq)[ee];
            const long long xl = rc_load_L(V, x, u, xd);
            sd += sgn * xd; sl += sgn * xl; sz += sgn; ++cnt;
        } else {
            ((int *)acc)[6] = T.blab[q];   // the singleton took a new label (at most one such entry per slot and batch)
        }
    }
    if (cnt) {
        atomicAdd((unsigned long long *)&acc[0], (unsigned long long)sd);
        atomicAdd((unsigned long long *)&acc[1], (unsigned long long)sl);
        atomicAdd(&((int *)acc)[4], sz);
        atomicAdd(&((int *)acc)[5], cnt);
    }
}

// Pruned candidates.  The Gumbel noise of a candidate is at most RC_GUMBEL_MAX (u <= 1 - 2^-53: -log(-log u) <= 36.7369), so a
// candidate whose noise-free score v0 satisfies v0 + RC_GUMBEL_MAX < (a score some other candidate of the point has for certain)
// cannot be the argmax, whatever its uniform is: its counter hash and the two logs of the noise — 180 of a candidate's ~330 VALU
// instructions — are skipped.  The comparison is strict and the bound exact, so no draw changes.  The certain score is the point's
// own cluster's (evaluated first, by one stream of every wave, and handed to the wave's other stream by a shuffle) or the
// stream's best so far.  In the score cache a pruned candidate is a quiet NaN whose low word holds v0 + RC_GUMBEL_MAX as a float
// rounded up: a later pass skips it again if the bound is still below what it has, and computes it exactly otherwise.
#define RC_GUMBEL_MAX 36.74
__device__ __forceinline__ double rc_pruned_tag(double bound)
{
    return __hiloint2double(0x7ff80000, (int)__float_as_uint(__double2float_ru(bound)));
}
__device__ __forceinline__ double rc_pruned_bound(double tagged) { return (double)__uint_as_float((unsigned)__double2loint(tagged)); }

// Score cache (cmode; V.wc).  The score of (point i, cluster k ≠ i's own) — size term, likelihood, noise of (sweep, i, label) —
// changes inside a sweep only when a committed change touches slot k (size, row sums or label) or, under validation, when a
// batch entry before i does.  The first tentative pass of a sweep evaluates every point against every cluster and stores the
// scores (cmode 1: W[k][i], i in sweep order, 8 B per pair).  A later tentative pass (cmode 2, mode 0) reads them back for the
// slots that are still clean — T.act2 lists those first (tab_partition) — computes the slots the last commit touched and stores
// their new scores, so that after it the cache is current again for every open point and the dirty flags are cleared.  A
// validation pass (cmode 2, mode 1) computes a slot of the batch only for the points that have one of its entries before them.
// Always computed: the point's own cluster (its score excludes the point itself), the clusters born in the batch, the
// new-cluster candidate.  The cached value is the very double the computation would produce again, so decisions are
// unchanged; a computed candidate costs ~330 VALU instructions, a cached one a load (moving regime, K = 206: eight passes per sweep).
// -DRC_PROF_EVAL (profiling builds): the longest thread of the block per part of a VALIDATION pass — cached clean slots / own cluster
// and computed slots / births, new cluster and the arg-max — summed over the chunks into T.misc[19..21] (10 ns ticks)
#ifdef RC_PROF_EVAL
#define RC_PE_STAMP(k) if (mode == 1) { const long long n_ = __builtin_amdgcn_s_memrealtime(); pe_[(k) - 19] += (int)(n_ - pe_t); pe_t = n_; }
#else
#define RC_PE_STAMP(k)
#endif
__device__ __forceinline__ void eval_chunk(const View &V, const SweepArgs &a, Tab &T, const long long *SD, const long long *SL,
                           int chunk, int cidx /* chunk = blockIdx.x + cidx G */, int lo, int hi, int mode, int nb, u64 *cword, unsigned *rec, unsigned stamp, int cmode,
                           u64 *cword_next = nullptr, unsigned *rec_next = nullptr, unsigned stamp_next = 0u)
{
    const int pt = threadIdx.x & (RC_PTS - 1), st = threadIdx.x >> RC_PTS_LOG2, NS = blockDim.x >> RC_PTS_LOG2;
    const int wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
    const int i = chunk * RC_PTS + pt;
    const bool valid = (i < V.n) && (i > lo) && (i <= hi);
#ifdef RC_PROF_EVAL
    int pe_[3] = {0, 0, 0};
    long long pe_t = __builtin_amdgcn_s_memrealtime();
#endif
    const int K = T.misc[0];
    double bestv = -INFINITY;
    int bestpos = 0x7fffffff, bestslot = -2;
    int own = 0, u = 0, j = 0;
    // (tentative passes only: `mode` is a literal at both call sites, so the validation copy of this function carries no pruning code
    // at all — with it, as dead weight, the moving regime ran 3 % slower; sweeps that prune hardly ever validate)
    const bool prune = mode == 0 && a.prune != 0 && !(a.dbg & 5);
    if (valid) {
        // matrices, S and slot_of are stored in the internal (cluster-contiguous) point order
        if (T.cached) { u = T.cu[cidx * RC_PTS + pt]; own = T.cown[cidx * RC_PTS + pt]; }
        else { u = V.pi[i]; own = V.slot_of[u]; }
        // number of batch changers before i (bx ascending)
        if (mode == 1) {
            int lo_ = 0, hi_ = nb;
            while (lo_ < hi_) { const int mid = (lo_ + hi_) >> 1; if (T.bx[mid] < i) lo_ = mid + 1; else hi_ = mid; }
            j = lo_;
        }
    }
    // the batch's largest group, shared by the candidate streams of every point (see hot_accumulate)
    const int hot = (mode == 1) ? T.misc[16] : -1;
    long long *const hotacc = (long long *)T.red_v + 4 * pt;
    if (hot >= 0) {
        if (threadIdx.x < 4 * RC_PTS) ((long long *)T.red_v)[threadIdx.x] = 0;
        __syncthreads();
        if (valid && j > 0) hot_accumulate(V, T, hot, u, j, st, NS, hotacc);
        __syncthreads();
    }
    if (valid) {
        const size_t ld = (size_t)V.ld;
        int so = T.size[own];
        if (mode == 1 && own == hot) so += ((const int *)hotacc)[4];
        else if (mode == 1)
            for (int e = own ? T.seg[own - 1] : 0, e1 = T.seg[own]; e < e1; ++e) {
                const int q = T.pairs[e];
                if (q >= j) break;
                so += (T.bb[q] == own) - (T.ba[q] == own);
            }
        const int single = (so == 1);
        const int Ki = ((mode == 1 && j > 0) ? (int)T.bK[j - 1] : K) - single;
        const long long dg = V.diagq[u];
        double *const wrow = V.wc + i;   // column i of the score cache (used only when cmode != 0)
        const double inv_beta = 1.0 / V.beta, inv_gamma = 1.0 / V.gamma;
        auto consider_v = [&](const int k, long long sd, long long sl) {   // (sd, sl: the slot's two row sums for this point, as stored)
            const int isown = (k == own);
            int sz = T.size[k], lab = T.label[k];
            bool touched;
            if (mode == 1 && k == hot) {
                const int *hi_ = (const int *)hotacc;
                touched = hi_[5] > 0;
                sd += hotacc[0]; sl += hotacc[1]; sz += hi_[4];
                if (hi_[6]) lab = hi_[6];
            } else {
                touched = (mode == 1) && batch_corr(V, T, k, u, j, sd, sl, sz, lab);
            }
            const int s = sz - isown;
            if (s == 0) return;  // empty once i is removed (its own singleton cluster, mcmc.jl:193-196) or emptied by the batch
            sd -= (isown ? dg : 0);                                              // i itself excluded (clusts[i] = -1)
            const double SDr = (double)sd * V.scD, SLr = (double)sl * V.scL;      // logD diagonal is 0 (types.jl:155)
            double base;
            if (!touched) base = isown ? T.base_s[k] : T.base_o[k];
            else {   // (the same doubles tab_base would produce again: the common sizes of a touched cluster are tabulated per slot)
                const int s0 = T.size[k];
                base = (V.kcap < 2048 && s == s0 + 1) ? T.base_p[k] : (s == s0) ? T.base_o[k] : (s == s0 - 1 && s0 >= 2) ? T.base_s[k] : tab_base(V, a, s);
            }
#ifndef RC_NO_PRUNE
            if (prune && !isown && bestslot != -2) {
                // cheap test first: log1p(x) >= x / (1 + x) and log1p(y) <= y bound the noise-free score from above with two
                // multiplications and a refined reciprocal instead of two divisions and two log1p (~20 against ~150 instructions);
                // the slack covers every rounding of this estimate (relative 1e-9 of the terms' magnitudes, far above 2^-53)
                const double A_ = V.alpha + V.delta1 * (double)s, x_ = SDr * inv_beta, d_ = 1.0 + x_;
                double rc_ = __builtin_amdgcn_rcp(d_);
                rc_ = rc_ * (2.0 - d_ * rc_);                                         // one Newton step: relative error ~1e-15
                const double t1 = V.cL * SLr, t2 = A_ * (x_ * rc_);
                const double zy = V.repulsion ? (V.zeta + V.delta2 * (double)s) * (SDr * inv_gamma) : 0.0;
                const double ub = (((base + t1) - t2) + zy) + (RC_GUMBEL_MAX + 1e-3) + 1e-9 * (fabs(base) + fabs(t1) + A_ * x_ + zy);
                if (ub < bestv) {
                    if (cmode != 0 && mode == 0) wrow[(size_t)k * V.ldw] = rc_pruned_tag(ub);
                    return;
                }
            }
#endif
            double lik = V.cL * SLr - (V.alpha + V.delta1 * (double)s) * rc_flog1p(SDr / V.beta, T.flt);
            if (V.repulsion) lik += (V.zeta + V.delta2 * (double)s) * rc_flog1p(SDr / V.gamma, T.flt);
            double v = base + lik;
#ifndef RC_NO_PRUNE
            if (prune && !isown) {
                if (bestslot != -2 && v + RC_GUMBEL_MAX < bestv) {                    // cannot win: no noise needed
                    if (cmode != 0 && mode == 0) wrow[(size_t)k * V.ldw] = rc_pruned_tag(v + RC_GUMBEL_MAX);
                    return;
                }
            }
#endif
            if (!(a.dbg & 4)) {
                const double un = rc_uniform(a, (unsigned)i, (unsigned)lab);
                v = v + rc_gumbel(un, T.flt);
            }
            // tentative passes keep the cache current (dbg 16: timing ablation).  Entry (own slot, i) holds the point's OWN-cluster score
            // (itself removed): a point that is still open has not moved in this sweep, so the entry never means anything else
            if (cmode != 0 && mode == 0 && !(a.dbg & 16)) wrow[(size_t)k * V.ldw] = v;
            if (bestslot == -2 || v > bestv || (v == bestv && lab < bestpos)) { bestv = v; bestpos = lab; bestslot = k; }
        };
        auto consider = [&](const int k) { consider_v(k, SD[(size_t)k * ld + u], SL[(size_t)k * ld + u]); };
        // A point that is a cluster of its own has no own cluster to set the bar: every stream then starts with the new-cluster
        // candidate instead (log, hash and noise: cheap, and a singleton's usual draw).  Lanes of a wave run in lock step — one
        // point without a bar makes its whole wave compute the noise of every candidate.
        const bool new_ok = (V.maxK == 0 || (long long)Ki < V.maxK) && Ki < V.n;
        const bool new_first = prune && single && new_ok;
        if (new_first) {
            const double un = rc_uniform(a, (unsigned)i, 0u);
            bestv = (log((double)(Ki + 1)) + a.r * a.log1mp) + rc_gumbel(un, T.flt);
            bestpos = RC_NEWKEY; bestslot = -1;
        }
        // With pruning every stream starts with the point's own cluster (itself removed): its score is the bar the stream's other
        // candidates must be able to reach (see "Pruned candidates"; the same candidate in several streams is harmless: equal score,
        // equal label).  Without, the own cluster is one candidate of one stream.
        if (cmode != 2) {
            // every candidate is computed: its two row sums are requested one candidate ahead (the loop is a chain of global round
            // trips otherwise — with the table-driven logarithms a candidate is ~260 instructions, less than the latency of its loads)
            const int Kl = (a.dbg & 1) ? 0 : K;
            int pos = prune ? -1 : st;
            int k = (pos < 0) ? own : (pos < Kl ? (int)T.act[pos] : 0);
            long long sd = 0, sl = 0;
            if (pos < Kl) { sd = SD[(size_t)k * ld + u]; sl = SL[(size_t)k * ld + u]; }
            while (pos < Kl) {
                const int posn = (pos < 0) ? st : pos + NS;
                const int kn = posn < Kl ? (int)T.act[posn] : 0;
                long long sdn = 0, sln = 0;
                if (posn < Kl) { sdn = SD[(size_t)kn * ld + u]; sln = SL[(size_t)kn * ld + u]; }
                if (!(prune && pos >= 0 && k == own)) consider_v(k, sd, sl);
                pos = posn; k = kn; sd = sdn; sl = sln;
            }
        } else {
            const int Kc = T.misc[14];
            // the clean slots first (stored scores: their loads are in flight while nothing else is), ...
            // RC_WC_BATCH loads in flight per turn — a turn is one global round trip, and with six a stream's share of ~200 clusters
            // (32 streams per point) is one turn; the slot numbers are read again from LDS where they are used instead of being held
#ifndef RC_WC_BATCH
#define RC_WC_BATCH 6
#endif
            for (int pos = st; pos < Kc; pos += RC_WC_BATCH * NS) {
                double vv[RC_WC_BATCH];
#pragma unroll
                for (int q = 0; q < RC_WC_BATCH; ++q) {
                    const int p_ = pos + q * NS;
                    const int kq = (p_ < Kc) ? (int)T.act2[p_] : own;
                    vv[q] = (kq != own) ? wrow[(size_t)kq * V.ldw] : 0.0;
                }
                unsigned needm = 0u;   // (a bit mask, not an array indexed at run time: that would live in scratch memory)
#pragma unroll
                for (int q = 0; q < RC_WC_BATCH; ++q) {
                    const int p_ = pos + q * NS;
                    if (p_ >= Kc) continue;
                    const int kq = (int)T.act2[p_];
                    if (kq == own) continue;
                    const double v = vv[q];
                    if (v != v) {                                 // pruned when it was stored: still out of reach?
                        if (!(bestslot != -2 && rc_pruned_bound(v) < bestv)) needm |= 1u << q;
                        continue;
                    }
                    const int lab = T.label[kq];
                    if (bestslot == -2 || v > bestv || (v == bestv && lab < bestpos)) { bestv = v; bestpos = lab; bestslot = kq; }
                }
                while (needm) {                                      // (rare: computed exactly, and stored if this pass keeps the cache)
                    const int q = __ffs((int)needm) - 1;
                    needm &= needm - 1u;
                    consider((int)T.act2[pos + q * NS]);
                }
            }
            // ... then the point's own cluster and the slots a change touched (computed)
            RC_PE_STAMP(19)
            for (int pos = (prune || st == ((K + 1) % NS)) ? -1 : Kc + st; pos < K; pos = (pos < 0) ? Kc + st : pos + NS) {
                const int k = (pos < 0) ? own : (int)T.act2[pos];
                if (pos < 0) {
                    // the point's own cluster: its stored score stands while no committed change (dirty) and no batch entry before
                    // the point touched the slot (a lone point has no own-cluster candidate and no stored score)
                    if (single) continue;
                    if (!T.dirty[k] && k != hot) {
                        const int e0 = k ? T.seg[k - 1] : 0;
                        if (mode == 0 || e0 == T.seg[k] || T.pairs[e0] >= j) {
                            const double v = wrow[(size_t)k * V.ldw];
                            const int lab = T.label[k];
                            if (bestslot == -2 || v > bestv || (v == bestv && lab < bestpos)) { bestv = v; bestpos = lab; bestslot = k; }
                            continue;
                        }
                    }
                } else {
                    if (k == own) continue;
                    if (mode == 1 && !T.dirty[k]) {                // a slot of the batch: untouched for this point if no entry of its group precedes it
                        const int e0 = k ? T.seg[k - 1] : 0;
                        if (e0 == T.seg[k] || T.pairs[e0] >= j) {
                            const double v = wrow[(size_t)k * V.ldw];
                            if (v == v) {
                                const int lab = T.label[k];
                                if (bestslot == -2 || v > bestv || (v == bestv && lab < bestpos)) { bestv = v; bestpos = lab; bestslot = k; }
                                continue;
                            }
                            if (bestslot != -2 && rc_pruned_bound(v) < bestv) continue;    // pruned when it was stored, and still out of reach
                        }
                    }
                }
                consider(k);
            }
        }
        // clusters created by the changers before i: singletons {x_q}, row sums = row x_q of the matrices
        RC_PE_STAMP(20)
        if (mode == 1 && j > 0) {
            int lo_ = 0, hi_ = T.misc[8];
            while (lo_ < hi_) { const int mid = (lo_ + hi_) >> 1; if (T.birth[mid] < j) lo_ = mid + 1; else hi_ = mid; }
            for (int bi = st; bi < lo_; bi += NS) {
                const int q = T.birth[bi], lab = T.blab[q];
                const size_t e = (size_t)T.bu[q] * ld + u;
                const long long xd = (V.bits == 64) ? ((const long long *)V.Dq)[e] : (long long)((const int *)V.Dq)[e];
                const long long xl = rc_load_L(V, T.bu[q], u, xd);
                const double SDr = (double)xd * V.scD, SLr = (double)xl * V.scL;
                double lik = V.cL * SLr - (V.alpha + V.delta1) * rc_flog1p(SDr / V.beta, T.flt);
                if (V.repulsion) lik += (V.zeta + V.delta2) * rc_flog1p(SDr / V.gamma, T.flt);
                double v = (V.kcap < 2048 ? T.base_p[V.kcap] : tab_base(V, a, 1)) + lik;
                if (!(a.dbg & 4)) {
                    const double un = rc_uniform(a, (unsigned)i, (unsigned)lab);
                    v = v + rc_gumbel(un, T.flt);
                }
                if (bestslot == -2 || v > bestv || (v == bestv && lab < bestpos)) { bestv = v; bestpos = lab; bestslot = T.bb[q]; }
            }
        }
        // new-cluster candidate, last in the candidate order (mcmc.jl:198-203, 228-230); one stream handles it
        if (!new_first && st == (K % NS) && new_ok) {
            const double un = rc_uniform(a, (unsigned)i, 0u);
            const double v = (log((double)(Ki + 1)) + a.r * a.log1mp) + rc_gumbel(un, T.flt);
            if (v > bestv || bestslot == -2) { bestv = v; bestpos = RC_NEWKEY; bestslot = -1; }
        }
    }
    RC_PE_STAMP(21)
#ifdef RC_PROF_EVAL
    if (mode == 1) { atomicMax(&T.misc[19], pe_[0]); atomicMax(&T.misc[20], pe_[1]); atomicMax(&T.misc[21], pe_[2]); }
#endif
    if (hot >= 0) __syncthreads();   // (the records of the hot slot alias the reduction scratch)
    // reduce over the candidate streams: the 64 / RC_PTS streams of each wave by shuffles, then the waves through LDS
#pragma unroll
    for (int off = RC_PTS; off < 64; off <<= 1) {
        const double ov = __shfl_xor(bestv, off);
        const int op = __shfl_xor(bestpos, off), os = __shfl_xor(bestslot, off);
        best_merge(bestv, bestpos, bestslot, ov, op, os);
    }
    if ((threadIdx.x & 63) < RC_PTS) {
        T.red_v[wave * RC_PTS + pt] = bestv;
        T.red_pos[wave * RC_PTS + pt] = bestpos;
        T.red_slot[wave * RC_PTS + pt] = bestslot;
    }
    __syncthreads();
    if (wave == 0) {
        // lane l: point l % RC_PTS, wave group l / RC_PTS (waves g, g + 64 / RC_PTS, ...); then the groups by shuffles
        const int grp = (int)(threadIdx.x & 63) >> RC_PTS_LOG2, half = grp;   // (half == 0: the lanes that end up holding a point's result)
        double bv = -INFINITY;
        int bp = 0x7fffffff, bs = -2;
        for (int w = grp; w < NW; w += 64 / RC_PTS) best_merge(bv, bp, bs, T.red_v[w * RC_PTS + pt], T.red_pos[w * RC_PTS + pt], T.red_slot[w * RC_PTS + pt]);
#pragma unroll
        for (int off = RC_PTS; off < 64; off <<= 1) {
            const double ov = __shfl_xor(bv, off);
            const int op = __shfl_xor(bp, off), os = __shfl_xor(bs, off);
            best_merge(bv, bp, bs, ov, op, os);
        }
        bool changed = false;
        int target = own;
        if (half == 0 && valid) {
            // -1: new cluster, label = smallest empty label once i is removed (mcmc.jl:199).  A singleton that draws "new
            // cluster" gets min(its own label, smallest empty label) — often its own label, i.e. no change at all; it is
            // announced as a changer all the same and batch_sim decides (deciding here, under the labels free at this
            // point's turn, makes every such singleton after a death a violation: 4 -> 14 rounds per sweep measured).
            target = (bs >= 0) ? bs : -1;
            changed = (target != own);
        }
        if (mode == 0) {
            if (half == 0 && valid && changed)
                __hip_atomic_store(rec + i, ((unsigned)own << 16) | (unsigned)(target + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u64 m = __ballot(changed) & ((1ull << RC_PTS) - 1ull);
            if (m) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the records are out before the mask that announces them
                if (threadIdx.x == 0) __hip_atomic_store(cword + chunk, ((u64)stamp << 32) | m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            // The guess this draw is checked against is what the batch says the point does: its entry (the slot it joins; "new
            // cluster" for a birth, a rename or a singleton that keeps its label; "stays" for a placeholder), or no entry = stays.
            if (half == 0 && valid) {
                int guess = own;
                if (j < nb && T.bx[j] == i) {
                    const int fl = T.bflag[j];
                    guess = (fl & RC_BF_STAY) ? own : (fl & (RC_BF_BIRTH | RC_BF_RENAME | RC_BF_NOOP)) ? -1 : (int)T.bb[j];
                }
                if (target != guess) atomicMin(T.blk_key, (u64)(unsigned)i);
                // ... and the draw itself is the point's guess in the next round (announced in the other parity's buffers): it
                // already accounts for the changers before the point, which a fresh draw under the committed state would not
                if (changed) __hip_atomic_store(rec_next + i, ((unsigned)own << 16) | (unsigned)(target + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // (points of this chunk beyond the batch's range: a batch that does not reach the end of the sweep is followed by a fresh
            // tentative pass, see resolve_body)
            const u64 m = __ballot(changed) & ((1ull << RC_PTS) - 1ull);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (threadIdx.x == 0) __hip_atomic_store(cword_next + chunk, ((u64)stamp_next << 32) | m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
}

// Grid barrier: monotonic arrival counter; lane 0 of each block arrives after its (returning) atomicMin
// on the round's key word, so every block's candidate is in the word before anyone leaves.  Bounded spin.
// The arrival must also publish what this block announced to the others this round — the V.rec records and the chunk
// word, written with agent-scope atomic stores by several waves.  __syncthreads() only waits for LDS traffic
// (s_waitcnt lgkmcnt(0); global accesses of one CU are kept in order by its L1, which is all a workgroup barrier needs),
// so every wave first waits for its own outstanding global stores: without this another block could see the arrival
// before a record and assemble a different batch (observed as diverging chains once the timing changed).
// Two-level arrival (RC_BAR_* words of the arrival buffer, each on its own 128-byte line): the blocks arrive on one of eight
// group counters (blockIdx & 7), the last of a group on the top counter, the last of all publishes the barrier number in
// the eight release words the groups poll.  Returning atomics on one address serialise at ≈25 ns each — 256 blocks on one
// counter were 5-6 µs per barrier, twice per round; now 32 + 8 in sequence, and the polls no longer queue behind the arrivals.
#define RC_BAR_GROUPS 8
#define RC_BAR_STRIDE 32                      // unsigneds per line
#define RC_BAR_WORDS ((2 * RC_BAR_GROUPS + 1) * RC_BAR_STRIDE)
__device__ __forceinline__ bool grid_barrier(const View &V, Tab &T, unsigned *arrive, int G, unsigned nbar, u64 my_key, u64 *key_word)
{
    int &sh_ok = T.misc[6];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (my_key != RC_KEY_NONE) {
            const u64 old = atomicMin(key_word, my_key);
            asm volatile("" ::"v"((unsigned)old));  // data dependence: the min has been performed before we arrive
        }
        const int ngroups = min(G, RC_BAR_GROUPS), g = (int)blockIdx.x & (RC_BAR_GROUPS - 1);
        const unsigned members = (unsigned)((G - g + RC_BAR_GROUPS - 1) / RC_BAR_GROUPS);   // blocks b < G with b % 8 == g
        unsigned *grp = arrive + g * RC_BAR_STRIDE, *top = arrive + RC_BAR_GROUPS * RC_BAR_STRIDE;
        unsigned *rel = arrive + (RC_BAR_GROUPS + 1) * RC_BAR_STRIDE;
        if (__hip_atomic_fetch_add(grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == members * nbar)
            if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == (unsigned)ngroups * nbar)
                for (int q = 0; q < ngroups; ++q) __hip_atomic_store(rel + q * RC_BAR_STRIDE, nbar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        int ok = 1;
        while (__hip_atomic_load(rel + g * RC_BAR_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nbar) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > RC_SPIN_LIMIT) { ok = 0; atomicOr(&V.sc->err, RC_DERR_BARRIER); break; }
        }
        sh_ok = ok;
    }
    __syncthreads();
    return sh_ok != 0;
}

// T.act2 = the active slots with those whose cached scores are still valid first (T.misc[14] of them): not touched by a
// committed change of this sweep nor — with_batch, for the validation pass — by an entry of the current batch.  The order
// inside the two parts is arbitrary (ties between candidates are broken by label, not by position).  Ends synchronised.
__device__ __forceinline__ void tab_partition(const View &V, Tab &T, bool with_batch)
{
    if (threadIdx.x == 0) { T.misc[14] = 0; T.misc[15] = 0; }
    __syncthreads();
    const int K = T.misc[0];
    for (int pos = threadIdx.x; pos < K; pos += blockDim.x) {
        const int k = T.act[pos];
        const bool d = T.dirty[k] || (with_batch && (k ? T.seg[k - 1] : 0) != T.seg[k]);
        if (d) T.act2[K - 1 - atomicAdd(&T.misc[15], 1)] = (short)k;
        else T.act2[atomicAdd(&T.misc[14], 1)] = (short)k;
    }
    __syncthreads();
}

// smallest label > lab (1-based) whose bit is clear in the occupancy bitset; n + 1 if none
__device__ __forceinline__ int next_empty_label(const Tab &T, int n, int lab)
{
    const int nw = (n + 31) / 32;
    int w = lab >> 5;                              // bit index of label lab+1 is lab
    if (w >= nw) return n + 1;
    unsigned inv = ~T.used[w] & ~((1u << (lab & 31)) - 1u);
    while (!inv) {
        if (++w >= nw) return n + 1;
        inv = ~T.used[w];
    }
    const int r = w * 32 + __ffs((int)inv);       // 1-based label
    return r <= n ? r : n + 1;
}

// What every changer of the batch does when the batch is applied in order (identical in every block): moves between
// clusters, deaths, births (slot = next free slot of the committed table, label = smallest empty label at that moment,
// mcmc.jl:199), singletons that take a smaller label or keep theirs.  Fills bb (real target slot), blab / bflag /
// bK and the list of births; the batch is cut before an entry whose target cluster an earlier entry emptied (that
// point has to be drawn again) or that finds no free slot.
// Run by wave 0 with all lanes in step: 64 entries at a time are fetched lane-parallel (source, target, the source's
// label, whether either cluster can become empty inside the batch) and only the entries that need the running state —
// births, deaths, singletons, anything touching a cluster that could die — are visited one after the other, from
// registers plus one LDS round trip for the current sizes of their two slots.
// T.size is used in place (the caller restores it from the entries), T.used is scratch (the caller builds it).
// misc: [3] entries kept, [4] last point covered, [5] capacity failure, [8] births, [9] entries that change something,
// [10] index of the first of them.
__device__ __forceinline__ void batch_sim(const View &V, Tab &T, int total, int cap)
{
    const int lane = threadIdx.x & 63;
    const int nb0 = min(total, cap);
    int K = T.misc[0], se = T.misc[1], fcur = 0, nbirth = 0, neff = 0, first_eff = -1;
    int nb = nb0, hi = (total > cap) ? T.misc[2] - 1 : V.n - 1, fail = 0, nvisited = 0;
    bool stop = false;
    // Register-held running state (kcap < 2048: some label <= 2048 is always empty, so the smallest empty one lies in the
    // first 64 words of the bitset): lane w holds word w of the label bitset and the free-slot mask of slots 64w..64w+63.
    // The entries that need it are applied by ONE wave, one after the other: what counts is the number of dependent
    // instructions per entry, and a label taken or freed is two instructions here against LDS round trips (a birth cost
    // ~1.2 µs: up to four 64-slot steps to the first free slot, two read-modify-writes of the bitset and a word-by-word
    // walk to the next empty label).
#ifdef RC_PROF_SIM
    const long long pf0_ = __builtin_amdgcn_s_memrealtime();
    long long pf_pro_ = 0, pf_loop_ = 0, pf_epi_ = 0, pf_init_ = 0, pf_ser_ = 0, pf_b_ = 0, pf_d_ = 0, pf_ch_ = 0;
    long long pc_n_[3] = {0, 0, 0}, pc_t_[3] = {0, 0, 0}, pc_pre_ = 0, pc_it_ = 0, pc_g_[6] = {0, 0, 0, 0, 0, 0};   // per path (0 rename, 1 certain death, 2 general): entries, shader cycles; the iteration's preamble
#endif
    const bool regs = V.kcap < 2048;
    unsigned U = 0xffffffffu;
    u64 myfree = 0;
    if (regs) {
        if (lane < (V.n + 31) / 32) U = T.used[lane];
        for (int w = 0; w * 64 < V.kcap; ++w) {
            const int k = w * 64 + lane;
            const u64 m = __ballot(k < V.kcap && T.label[k] == 0);
            if (lane == w) myfree = m;
        }
    }
#ifdef RC_PROF_SIM
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); pf_init_ = __builtin_amdgcn_s_memrealtime() - pf0_;
#endif
    for (int q0 = 0; q0 < nb0 && !stop; q0 += 64) {
#ifdef RC_PROF_SIM
        const long long pc0_ = __builtin_amdgcn_s_memrealtime(); ++pf_ch_;
#endif
        int va = 0, vt = -1, vla = 0, vcda = 0, vcdt = 0;
        bool vfast = false, vsafe = false, vstay = false, vdeath = false, vbirth = false;
        if (q0 + lane < nb0) {
            va = T.ba[q0 + lane]; vt = T.bb[q0 + lane]; vla = T.label[va];
            // target == source: the placeholder of a carried-over guess whose target cluster is gone (the assembly wrote it so)
            vstay = (vt == va);
            vcda = T.candie[va]; vcdt = vt >= 0 ? (int)T.candie[vt] : 0;
            const bool lone_ = T.size[va] == 1 && !T.joined[va];   // a singleton nobody in the batch joins: alone at its turn whatever happened before
            // ... drawing "new cluster"
            vfast = vt < 0 && lone_;
            // ... joining a cluster that cannot become empty inside the batch: a death whatever the order — what depends on the order is
            // only what it does to the running state (one cluster less, its label free), see the short path below
            vdeath = vt >= 0 && !vstay && lone_ && !vcdt;
            // ... a point of a cluster that cannot become empty inside the batch, drawing "new cluster": a birth whatever the order
            vbirth = vt < 0 && !lone_ && !vcda;
            // a move between two clusters neither of which can become empty inside the batch: a plain move whatever the
            // order, nothing to simulate (the sizes of such clusters are not tracked here at all)
            vsafe = vt >= 0 && !vstay && !vcda && !vcdt;
        }
        const u64 safemask = __ballot(vsafe), staymask = __ballot(vstay), fastmask = __ballot(vfast), deathmask = __ballot(vdeath), birthmask = __ballot(vbirth);
        // results of entry q0 + lane (stored after the chunk).  The plain moves are not visited at all: their target is the
        // tentative one, and cluster count / smallest empty label are those left by the last visited entry before them
        int ob = vt, olab = 0, oflag = vstay ? (RC_BF_NOOP | RC_BF_STAY) : 0, oK = K;
        const int cnt = min(64, nb0 - q0);
        int done = cnt;
        u64 todo = ~safemask & ~staymask & (cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull));
#ifdef RC_PROF_SIM
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long psim0_ = __builtin_amdgcn_s_memrealtime(); pf_pro_ += psim0_ - pc0_;
#endif
        while (todo) {
#ifdef RC_PROF_SIM
            const long long pit0_ = __builtin_amdgcn_s_memtime(); ++pc_it_;
#endif
            // the lone singletons that keep their label under the current smallest empty label change nothing: all of them up to
            // the next entry that needs the serial path are settled at once (a hundred of them per round in the moving regime)
            const u64 ser = todo & ~__ballot(vfast && se >= vla);
            const u64 nop = todo & (ser ? ((ser & (0ull - ser)) - 1ull) : ~0ull);
            if (nop) {
                if ((nop >> lane) & 1) { ob = va; oflag = RC_BF_NOOP; oK = K; }
#ifndef RC_PROF_SIM
                nvisited += __popcll(nop);
#endif
                todo &= ~nop;
            }
            if (!ser) break;
            const int e = __ffsll((long long)ser) - 1;
            todo &= ~(1ull << e);
            ++nvisited;
            const int q = q0 + e;
#ifdef RC_PROF_SIM
            const long long pit1_ = __builtin_amdgcn_s_memtime(); pc_pre_ += pit1_ - pit0_;
#endif
            if (regs && ((fastmask >> e) & 1ull)) {
                // A lone singleton that draws "new cluster" and is still here: the smallest empty label is smaller than its own
                // (the others were settled above), so it takes that label and frees its own — slot, sizes and cluster count stay.
                // These renames are most of the serial entries of the moving regime (twenty per batch); on this short path an
                // entry is ~45 instructions of one wave instead of ~200 through the general case below.
                se = __builtin_amdgcn_readfirstlane(se); neff = __builtin_amdgcn_readfirstlane(neff); first_eff = __builtin_amdgcn_readfirstlane(first_eff);
                const int la_ = __builtin_amdgcn_readlane(vla, e), lab_ = se;
                if (lane == ((lab_ - 1) >> 5)) U |= 1u << ((lab_ - 1) & 31);
                if (lane == ((la_ - 1) >> 5)) U &= ~(1u << ((la_ - 1) & 31));   // (a label beyond 2048 has no lane: never the smallest empty one)
                {   // smallest empty label above lab_: its bit index is >= lab_ (the freed label la_ > lab_ is among the candidates)
                    const int w0 = lab_ >> 5;
                    unsigned inv = (lane >= w0) ? ~U : 0u;
                    if (lane == w0) inv &= ~((1u << (lab_ & 31)) - 1u);
                    const u64 anyw = __ballot(inv != 0u);
                    se = V.n + 1;
                    if (anyw) {
                        const int fl = __ffsll((long long)anyw) - 1;
                        const int r = fl * 32 + __ffs(__builtin_amdgcn_readlane((int)inv, fl));   // 1-based label
                        if (r <= V.n) se = r;
                    }
                    if (la_ < se) se = la_;   // (labels beyond the bitset's 64 words are not seen by the scan)
                }
                if (lane == e) { ob = va; olab = lab_; oflag = RC_BF_RENAME; }
                if (first_eff < 0) first_eff = q;
                ++neff;
#ifdef RC_PROF_SIM
                ++pf_ser_; ++pc_n_[0]; pc_t_[0] += __builtin_amdgcn_s_memtime() - pit1_;
#endif
                continue;
            }
            if (regs && ((deathmask >> e) & 1ull)) {
                // A lone singleton that joins a cluster which cannot die in this batch (most of the structural entries of the moving
                // regime: twenty per batch): its cluster dies for certain — target, flag and sizes do not depend on the entries before
                // it; the running state loses a cluster and gains a free label.  ~15 instructions and no LDS round trip instead of ~70
                // through the general case below (0.55 us per entry on the one wave that applies them).
                se = __builtin_amdgcn_readfirstlane(se); K = __builtin_amdgcn_readfirstlane(K);
                neff = __builtin_amdgcn_readfirstlane(neff); first_eff = __builtin_amdgcn_readfirstlane(first_eff);
                const int la_ = __builtin_amdgcn_readlane(vla, e), a_ = __builtin_amdgcn_readlane(va, e);
                if (lane == ((la_ - 1) >> 5)) U &= ~(1u << ((la_ - 1) & 31));   // (a label beyond 2048 has no lane: never the smallest empty one)
                if (la_ < se) se = la_;
                K -= 1;
                if (lane == 0) T.size[a_] = 0;                                   // (the caller restores the simulated sizes from the entries)
                if (lane == e) { ob = vt; olab = 0; oflag = RC_BF_DEATH; }
                if (lane >= e) oK = K;
                if (first_eff < 0) first_eff = q;
                ++neff;
#ifdef RC_PROF_SIM
                ++pf_ser_; ++pf_d_; ++pc_n_[1]; pc_t_[1] += __builtin_amdgcn_s_memtime() - pit1_;
#endif
                continue;
            }
            if (regs && ((birthmask >> e) & 1ull) && __builtin_amdgcn_readfirstlane(nbirth) < RC_BIRTH_MAX && __ballot(myfree != 0ull) != 0ull) {
                // A certain birth (the source cluster cannot die in this batch) with a free slot at hand and the batch's births not
                // used up — the other cases (no slot: capacity failure; too many births: cut) go through the general path below.
                // Slot = lowest free slot of the committed table, label = smallest empty label (mcmc.jl:199).
                se = __builtin_amdgcn_readfirstlane(se); K = __builtin_amdgcn_readfirstlane(K); nbirth = __builtin_amdgcn_readfirstlane(nbirth);
                neff = __builtin_amdgcn_readfirstlane(neff); first_eff = __builtin_amdgcn_readfirstlane(first_eff);
                const u64 anyfree = __ballot(myfree != 0ull);
                const int fw = __ffsll((long long)anyfree) - 1;
                const u64 word = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(myfree >> 32), fw) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)myfree, fw);
                const int bitpos = __ffsll((long long)word) - 1, f = fw * 64 + bitpos, lab_ = se;
                if (lane == fw) myfree &= ~(1ull << bitpos);
                if (lane == ((lab_ - 1) >> 5)) U |= 1u << ((lab_ - 1) & 31);
                {   // smallest empty label above lab_
                    const int w0 = lab_ >> 5;
                    unsigned inv = (lane >= w0) ? ~U : 0u;
                    if (lane == w0) inv &= ~((1u << (lab_ & 31)) - 1u);
                    const u64 anyw = __ballot(inv != 0u);
                    se = V.n + 1;
                    if (anyw) {
                        const int fl = __ffsll((long long)anyw) - 1;
                        const int r = fl * 32 + __ffs(__builtin_amdgcn_readlane((int)inv, fl));   // 1-based label
                        if (r <= V.n) se = r;
                    }
                }
                K += 1;
                if (lane == 0) { T.size[f] = 1; T.birth[nbirth] = (short)q; }
                ++nbirth;
                if (lane == e) { ob = f; olab = lab_; oflag = RC_BF_BIRTH; }
                if (lane >= e) oK = K;
                if (first_eff < 0) first_eff = q;
                ++neff;
#ifdef RC_PROF_SIM
                ++pf_ser_; ++pf_b_; ++pc_n_[1]; pc_t_[1] += __builtin_amdgcn_s_memtime() - pit1_;
#endif
                continue;
            }
            const int a = __builtin_amdgcn_readlane(va, e), la = __builtin_amdgcn_readlane(vla, e);
            // the running state is the same in every lane: keep it in scalar registers (one wave applies these entries one after
            // the other, ~10 cycles per dependent instruction; with the state in vector registers every decision below was a
            // compare + exec-mask save / restore: 150 instructions and 0.7 µs per entry)
            se = __builtin_amdgcn_readfirstlane(se); K = __builtin_amdgcn_readfirstlane(K); fcur = __builtin_amdgcn_readfirstlane(fcur);
            nbirth = __builtin_amdgcn_readfirstlane(nbirth); neff = __builtin_amdgcn_readfirstlane(neff); first_eff = __builtin_amdgcn_readfirstlane(first_eff);
            int b, flag = 0, lab = 0, old = 0;
            {
                const int tgt = __builtin_amdgcn_readlane(vt, e);
                // (everything here is the same in every lane; saying so — readfirstlane — turns the decisions below into scalar
                // branches instead of compares, exec masks and their restores: the loop is bound by its instruction count)
                // The could-die flags come from the chunk's prefetch; the running sizes — kept only for slots that could die — are
                // the one LDS round trip of an entry, and only if one of its two slots is such a slot.
                const bool cda = __builtin_amdgcn_readlane(vcda, e) != 0, cdt = __builtin_amdgcn_readlane(vcdt, e) != 0;
                // a lone singleton that draws "new cluster" (nobody in the batch joins it): alone at its turn, no size to look up —
                // it takes the smallest empty label if that is smaller than its own.  These are most of the serial entries of the
                // moving regime (twenty renames per batch); without the LDS round trip for the sizes an entry is ~40 instructions.
                const bool lone = ((fastmask >> e) & 1ull) != 0ull;
                int sza = 0, szt = 0;
                if (!lone && (cda || cdt)) {
                    const int x_ = T.size[a], y_ = T.size[tgt >= 0 ? tgt : a];   // (both issued before either is waited for)
                    sza = __builtin_amdgcn_readfirstlane(x_); szt = __builtin_amdgcn_readfirstlane(y_);
                }
                b = tgt;
                if (lone) {
                    b = a;
                    if (se < la) { flag = RC_BF_RENAME; lab = se; old = la; }
                    else flag = RC_BF_NOOP;
                } else if (tgt >= 0 && cdt && szt == 0) {
                    // the target cluster was emptied by an earlier entry: the guess is void.  The entry stays as a placeholder that
                    // does nothing ("stays" is the new guess; the validation draws the point under the state as it is at its turn
                    // and, if it does move, the round ends there with that draw as the next guess — cutting the batch here instead
                    // ended the round for certain and left every point behind it with a guess nobody had checked)
                    b = a; flag = RC_BF_NOOP | RC_BF_STAY;
                } else if (tgt >= 0) {
                    if (cda && sza == 1) { flag = RC_BF_DEATH; old = la; K -= 1; }
                } else if (cda && sza == 1) {
                    b = a;
                    if (se < la) { flag = RC_BF_RENAME; lab = se; old = la; }
                    else flag = RC_BF_NOOP;
                } else {
                    // next free slot of the committed table (none is reused inside a batch)
                    int f = -1;
                    if (regs) {
                        const u64 anyfree = __ballot(myfree != 0ull);
                        if (anyfree) {
                            const int fw = __ffsll((long long)anyfree) - 1;
                            const u64 word = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(myfree >> 32), fw) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)myfree, fw);
                            const int bitpos = __ffsll((long long)word) - 1;
                            f = fw * 64 + bitpos;
                            if (lane == fw) myfree &= ~(1ull << bitpos);
                        }
                    } else {
                        while (fcur < V.kcap) {   // 64 slots per step
                            const int k = fcur + lane;
                            const u64 fr = __ballot(k < V.kcap && T.label[k] == 0);
                            if (fr) { f = fcur + __ffsll((long long)fr) - 1; break; }
                            fcur += 64;
                        }
                    }
                    if (f < 0) { fail = (q == 0); nb = q; hi = T.bx[q] - 1; stop = true; done = e; break; }
                    // births are the expensive entries of the simulation and a batch with dozens of them rarely survives validation
                    // whole: cut (first sweep from random labels: 29 -> 22 ms at kcap = 512; no effect at equilibrium)
                    if (nbirth >= RC_BIRTH_MAX) { nb = q; hi = T.bx[q] - 1; stop = true; done = e; break; }
                    fcur = f + 1;   // (a partial step is re-read from f + 1 on: harmless)
                    b = f; flag = RC_BF_BIRTH; lab = se; K += 1;
                }
                if (regs) {
                    if (lab && lane == ((lab - 1) >> 5)) U |= 1u << ((lab - 1) & 31);
                    if (old && lane == ((old - 1) >> 5)) U &= ~(1u << ((old - 1) & 31));   // (a label beyond 2048 has no lane: never the smallest empty one)
                    if (lab) {   // smallest empty label above lab: its bit index is >= lab
                        const int w0 = lab >> 5;
                        unsigned inv = (lane >= w0) ? ~U : 0u;
                        if (lane == w0) inv &= ~((1u << (lab & 31)) - 1u);
                        const u64 anyw = __ballot(inv != 0u);
                        se = V.n + 1;
                        if (anyw) {
                            const int fl = __ffsll((long long)anyw) - 1;
                            const int r = fl * 32 + __ffs(__builtin_amdgcn_readlane((int)inv, fl));   // 1-based label
                            if (r <= V.n) se = r;
                        }
                    } else if (old && old < se) se = old;
                } else {
                    // (one lane writes: 64 lanes storing to one LDS address are serialised)
                    if (lane == 0) {
                        if (lab) T.used[(lab - 1) >> 5] |= 1u << ((lab - 1) & 31);
                        if (old) T.used[(old - 1) >> 5] &= ~(1u << ((old - 1) & 31));
                    }
                    if (lab) se = __builtin_amdgcn_readfirstlane(next_empty_label(T, V.n, lab));
                    else if (old && old < se) se = old;
                }
                if (lane == 0) {
                    if (a != b) {
                        if (cda) T.size[a] = sza - 1;
                        if (flag & RC_BF_BIRTH) T.size[b] = 1; else if (cdt) T.size[b] = szt + 1;
                    }
                    if (flag & RC_BF_BIRTH) T.birth[nbirth] = (short)q;
                }
                if (flag & RC_BF_BIRTH) ++nbirth;
                if (!(flag & RC_BF_NOOP)) { if (first_eff < 0) first_eff = q; ++neff; }
            }
            if (lane == e) { ob = b; olab = lab; oflag = flag; }
            if (lane >= e) oK = K;
#ifdef RC_PROF_SIM
            ++pf_ser_; pf_b_ += (flag & RC_BF_BIRTH) ? 1 : 0; pf_d_ += (flag & RC_BF_DEATH) ? 1 : 0; ++pc_n_[2]; pc_t_[2] += __builtin_amdgcn_s_memtime() - pit1_;
            pc_g_[(flag & RC_BF_BIRTH) ? 0 : (flag & RC_BF_DEATH) ? 1 : (flag & RC_BF_RENAME) ? 2 : (flag & RC_BF_STAY) ? 3 : (flag & RC_BF_NOOP) ? 4 : 5] += 1;
#endif
        }
#ifdef RC_PROF_SIM
        const long long psim1_ = __builtin_amdgcn_s_memrealtime(); pf_loop_ += psim1_ - psim0_;
#endif
        {   // the plain moves before the cut count as effective entries
            const u64 kept = safemask & (done == 64 ? ~0ull : ((1ull << done) - 1ull));
            if (kept) {
                const int fs = q0 + __ffsll((long long)kept) - 1;
                if (first_eff < 0 || fs < first_eff) first_eff = fs;
                neff += __popcll(kept);
            }
        }
        if (lane < done) {
            const int q = q0 + lane;
            T.bb[q] = (short)ob; T.blab[q] = olab; T.bflag[q] = (unsigned char)oflag; T.bK[q] = (short)oK;
        }
#ifdef RC_PROF_SIM
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); pf_epi_ += __builtin_amdgcn_s_memrealtime() - psim1_;
#endif
    }
#ifdef RC_PROF_SIM
    if (blockIdx.x == 0 && lane == 0) {
        long long *dbg_ = (long long *)((char *)V.work[0] + 64);
        dbg_[0] += 1; dbg_[1] += pf_init_; dbg_[2] += pf_pro_; dbg_[3] += pf_loop_; dbg_[4] += pf_epi_; dbg_[5] += pf_ser_; dbg_[6] += pf_b_; dbg_[7] += pf_d_; dbg_[8] += pf_ch_;
        dbg_[9] += nb0; dbg_[10] += __builtin_amdgcn_s_memrealtime() - pf0_;
        for (int c_ = 0; c_ < 3; ++c_) { dbg_[11 + 2 * c_] += pc_n_[c_]; dbg_[12 + 2 * c_] += pc_t_[c_]; }
        dbg_[17] += pc_it_; dbg_[18] += pc_pre_;
        for (int c_ = 0; c_ < 6; ++c_) dbg_[19 + c_] += pc_g_[c_];
    }
#endif
    if (lane == 0) { T.misc[3] = nb; T.misc[4] = hi; T.misc[5] = fail; T.misc[8] = nbirth; T.misc[9] = neff; T.misc[10] = first_eff; T.misc[13] = nvisited; T.misc[18] = 0; }
}

// The same simulation, reorganised around what is really sequential (round 4; kcap < 2048, the label bitset in registers).
// batch_sim above walks the entries that need the running state one after the other and does everything for each in turn —
// sizes from LDS, kind of entry, free slot, label bitset, smallest empty label, cluster count, flags — 500 to 2000 cycles of
// one wave per entry (profiles/r04: 28 such entries per batch in the moving regime, 20 of the 23 µs of a call, 55 µs per
// sweep: the largest phase of the resolver).  Only two things are chains:
//   (1) the running SIZES of the clusters that could become empty inside the batch, which decide what an entry IS — a move, a
//       death, a birth, a singleton drawing "new cluster", a placeholder.  Most entries are decided by the chunk prefetch alone
//       (lone singletons, clusters that cannot die); the others are walked in order with one LDS round trip each, and nothing else;
//   (2) the free-LABEL set, a sequence of "label freed" (death), "take the smallest" (birth) and both (a singleton drawing "new
//       cluster": it takes min(own label, smallest empty label), mcmc.jl:199).  One short scalar-driven loop over these events:
//       lane w holds the free labels 32w+1 .. 32w+32 as a bit mask, the smallest empty label is a scalar; singletons that keep their
//       label (own label below the smallest empty one) are skipped in bulk, as before.
// Everything else is computed for the 64 entries of a chunk at once from the ballots of the kinds: cluster count after every
// entry (prefix pop-counts), the births' slots (r-th free slot of the committed table), the list of births, the cut (a birth
// without a free slot or beyond RC_BIRTH_MAX), effective-entry count.  T.size is put back before returning (misc[18] = 1: the
// caller's restore is skipped).  Same results as batch_sim entry for entry (the randomised comparisons run both).
__device__ __forceinline__ void batch_sim_fast(const View &V, Tab &T, int total, int cap)
{
    const int lane = threadIdx.x & 63;
    const int nb0 = min(total, cap);
    int K = T.misc[0], se = T.misc[1], nbirth = 0, neff = 0, first_eff = -1;
    int nb = nb0, hi = (total > cap) ? T.misc[2] - 1 : V.n - 1, fail = 0, nvisited = 0;
    bool stop = false;
#ifdef RC_PROF_SIM
    const long long pf0_ = __builtin_amdgcn_s_memrealtime();
    long long pf_pro_ = 0, pf_loop_ = 0, pf_epi_ = 0, pf_init_ = 0, pf_b_ = 0, pf_d_ = 0, pf_ch_ = 0, pc_unc_ = 0, pc_unc_t_ = 0, pc_ev_ = 0, pc_ev_t_ = 0;
#endif
    // free labels (bit set = free): lane w holds labels 32w+1 .. 32w+32; free slots of the committed table: lane w holds slots 64w .. 64w+63
    unsigned F = 0u;
    if (lane < (V.n + 31) / 32) F = ~T.used[lane];
    u64 myfree = 0;
    int nfree = 0;
    for (int w = 0; w * 64 < V.kcap; ++w) {
        const int k = w * 64 + lane;
        const u64 m = __ballot(k < V.kcap && T.label[k] == 0);
        if (lane == w) myfree = m;
        nfree += __popcll(m);
    }
    nfree = __builtin_amdgcn_readfirstlane(nfree);
#ifdef RC_PROF_SIM
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); pf_init_ = __builtin_amdgcn_s_memrealtime() - pf0_;
#endif
    const u64 le = (2ull << lane) - 1ull, lt = le >> 1;          // lanes <= / < this one
    for (int q0 = 0; q0 < nb0 && !stop; q0 += 64) {
#ifdef RC_PROF_SIM
        const long long pc0_ = __builtin_amdgcn_s_memrealtime(); ++pf_ch_;
#endif
        const int cnt = min(64, nb0 - q0);
        const bool in = lane < cnt;
        int va = 0, vt = -1, vla = 0;
        bool vstay = false, cda = false, cdt = false, lone = false;
        if (in) {
            va = T.ba[q0 + lane]; vt = T.bb[q0 + lane]; vla = T.label[va];
            vstay = (vt == va);                                      // placeholder of a carried-over guess whose target is gone (assembly)
            cda = T.candie[va] != 0; cdt = vt >= 0 && T.candie[vt] != 0;
            lone = T.size[va] == 1 && !T.joined[va];                 // a singleton nobody in the batch joins: alone at its turn whatever happened before
        }
        // kinds the prefetch decides (see batch_sim): singleton drawing "new cluster"; certain death; certain birth; plain move
        const bool c_snew = in && vt < 0 && lone, c_death = in && vt >= 0 && !vstay && lone && !cdt;
        const bool c_birth = in && vt < 0 && !lone && !cda, c_safe = in && vt >= 0 && !vstay && !cda && !cdt;
        u64 Mstay = __ballot(in && vstay), Msnew = __ballot(c_snew), Mdeath = __ballot(c_death), Mbirth = __ballot(c_birth), Mmove = __ballot(c_safe);
        u64 uncm = __ballot(in && !vstay && !c_snew && !c_death && !c_birth && !c_safe);
        const u64 cdam = __ballot(cda), cdtm = __ballot(cdt);
        u64 decA = 0, incT = 0;                                      // entries that took a point out of a tracked source / put one into a tracked target
#ifdef RC_PROF_SIM
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long psim0_ = __builtin_amdgcn_s_memrealtime(); pf_pro_ += psim0_ - pc0_;
        const long long pu0_ = __builtin_amdgcn_s_memtime();
#endif
        // (1) the entries whose kind depends on the running sizes, in order: the two sizes are the one LDS round trip of an entry
        while (uncm) {
            const int e = __ffsll((long long)uncm) - 1;
            const u64 bit = 1ull << e;
            uncm &= ~bit;
            ++nvisited;
#ifdef RC_PROF_SIM
            ++pc_unc_;
#endif
            const int a = __builtin_amdgcn_readlane(va, e), t = __builtin_amdgcn_readlane(vt, e);
            const bool ca = (cdam & bit) != 0ull, ct = (cdtm & bit) != 0ull;
            const int x_ = T.size[a], y_ = T.size[t >= 0 ? t : a];   // (both issued before either is waited for)
            const int sza = __builtin_amdgcn_readfirstlane(x_), szt = __builtin_amdgcn_readfirstlane(y_);
            if (t >= 0) {
                if (ct && szt == 0) {
                    Mstay |= bit;                                     // the target was emptied by an earlier entry: a placeholder ("stays")
                } else {
                    if (ca && sza == 1) Mdeath |= bit; else Mmove |= bit;
                    if (ca) { if (lane == 0) T.size[a] = sza - 1; decA |= bit; }
                    if (ct) { if (lane == 0) T.size[t] = szt + 1; incT |= bit; }
                }
            } else if (ca && sza == 1) {
                Msnew |= bit;                                         // alone at its turn: keeps its label or takes a smaller one
            } else {
                Mbirth |= bit;
                if (ca) { if (lane == 0) T.size[a] = sza - 1; decA |= bit; }
            }
        }
#ifdef RC_PROF_SIM
        pc_unc_t_ += __builtin_amdgcn_s_memtime() - pu0_;
#endif
        // the cut: a birth that finds no free slot (capacity: the host grows the tables if it is the batch's first entry) or is the
        // batch's RC_BIRTH_MAX-th — the rest of the batch is announced again
        int done = cnt;
        const int nbc_all = __popcll(Mbirth), room = min(nfree, RC_BIRTH_MAX) - nbirth;
        if (nbc_all > room) {
            u64 m = Mbirth;
            for (int r = 0; r < room; ++r) m &= m - 1ull;
            const int e_cut = __ffsll((long long)m) - 1;
            if (nbirth + room >= nfree) fail = (q0 + e_cut == 0);
            nb = q0 + e_cut; hi = T.bx[nb] - 1; stop = true; done = e_cut;
            const u64 tail = ~((1ull << e_cut) - 1ull);
            // the sizes the entries at and behind the cut changed go back at once (those before it at the end, from their flag bits)
            if (((decA & tail) >> lane) & 1ull) atomicAdd(&T.size[va], 1);
            if (((incT & tail) >> lane) & 1ull) atomicSub(&T.size[vt], 1);
            Mstay &= ~tail; Msnew &= ~tail; Mdeath &= ~tail; Mbirth &= ~tail; Mmove &= ~tail; decA &= ~tail; incT &= ~tail;
        }
        // (2) the label events, in order
        int olab = 0;
        {
#ifdef RC_PROF_SIM
            const long long pe0_ = __builtin_amdgcn_s_memtime();
#endif
            const u64 Mdb = Mdeath | Mbirth;
            u64 todo = Mdb | Msnew;
            se = __builtin_amdgcn_readfirstlane(se);
            while (todo) {
                // singletons whose own label lies below the smallest empty one keep it: nothing changes, nothing to do
                const u64 ser = todo & (Mdb | __ballot(vla > se));
                if (!ser) break;
                const int e = __ffsll((long long)ser) - 1;
                const u64 bit = 1ull << e;
                todo &= ~((bit << 1) - 1ull);
                ++nvisited;
#ifdef RC_PROF_SIM
                ++pc_ev_;
#endif
                const int la = __builtin_amdgcn_readlane(vla, e);
                if (Mdeath & bit) {
                    if (lane == ((la - 1) >> 5)) F |= 1u << ((la - 1) & 31);        // (a label beyond 2048 has no lane: never the smallest empty one)
                    if (la < se) se = la;
                } else {
                    const int lab = se;                                              // birth / rename: the smallest empty label (mcmc.jl:199)
                    if (lane == ((lab - 1) >> 5)) F &= ~(1u << ((lab - 1) & 31));
                    const bool sn = (Msnew & bit) != 0ull;
                    if (sn && lane == ((la - 1) >> 5)) F |= 1u << ((la - 1) & 31);   // the singleton's old label is free now
                    const u64 anyw = __ballot(F != 0u);
                    se = V.n + 1;
                    if (anyw) {
                        const int fl = __ffsll((long long)anyw) - 1;
                        const int r = fl * 32 + __ffs(__builtin_amdgcn_readlane((int)F, fl));   // 1-based label
                        if (r <= V.n) se = r;
                    }
                    if (sn && la < se) se = la;                                      // (labels beyond the 64 words are not seen by the scan)
                    if (lane == e) olab = lab;
                }
            }
#ifdef RC_PROF_SIM
            pc_ev_t_ += __builtin_amdgcn_s_memtime() - pe0_;
#endif
        }
#ifdef RC_PROF_SIM
        const long long psim1_ = __builtin_amdgcn_s_memrealtime(); pf_loop_ += psim1_ - psim0_;
        pf_b_ += __popcll(Mbirth); pf_d_ += __popcll(Mdeath);
#endif
        // everything else, for the whole chunk at once
        const int nbc = __popcll(Mbirth);
        int slots = 0;                                               // lane r: the r-th free slot of the committed table not yet handed out
        for (int r = 0; r < nbc; ++r) {
            const u64 anyfree = __ballot(myfree != 0ull);             // (non-zero: births <= room)
            const int fw = __ffsll((long long)anyfree) - 1;
            const u64 word = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(myfree >> 32), fw) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)myfree, fw);
            const int bitpos = __ffsll((long long)word) - 1;
            if (lane == fw) myfree &= ~(1ull << bitpos);
            if (lane == r) slots = fw * 64 + bitpos;
        }
        const bool isb = ((Mbirth >> lane) & 1ull) != 0ull, issn = ((Msnew >> lane) & 1ull) != 0ull, isst = ((Mstay >> lane) & 1ull) != 0ull;
        const int brank = __popcll(Mbirth & lt);
        const int bslot = __shfl(slots, brank & 63);
        const int oK = K + __popcll(Mbirth & le) - __popcll(Mdeath & le);
        int ob = vt, oflag = 0;
        if (isst) { ob = va; oflag = RC_BF_NOOP | RC_BF_STAY; }
        else if (issn) { ob = va; oflag = olab ? RC_BF_RENAME : RC_BF_NOOP; }
        else if (isb) { ob = bslot; oflag = RC_BF_BIRTH; T.birth[nbirth + brank] = (short)(q0 + lane); }
        else if ((Mdeath >> lane) & 1ull) oflag = RC_BF_DEATH;
        {
            const u64 effm = Mmove | Mdeath | Mbirth | __ballot(issn && olab != 0);
            if (effm) {
                if (first_eff < 0) first_eff = q0 + __ffsll((long long)effm) - 1;
                neff += __popcll(effm);
            }
        }
        if (lane < done) {
            const int q = q0 + lane;
            T.bb[q] = (short)ob; T.blab[q] = olab; T.bK[q] = (short)oK;
            T.bflag[q] = (unsigned char)(oflag | (((decA >> lane) & 1ull) ? 0x20 : 0) | (((incT >> lane) & 1ull) ? 0x40 : 0));
        }
        K += nbc - __popcll(Mdeath);
        nbirth += nbc;
#ifdef RC_PROF_SIM
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); pf_epi_ += __builtin_amdgcn_s_memrealtime() - psim1_;
#endif
    }
    // the running sizes back to the committed ones (flag bits 0x20: left a tracked source, 0x40: joined a tracked target; bb of such
    // an entry is still its guessed target: a move or a death)
    __builtin_amdgcn_wave_barrier();
    for (int q = lane; q < nb; q += 64) {
        const int fl = T.bflag[q];
        if (fl & 0x60) {
            if (fl & 0x20) atomicAdd(&T.size[T.ba[q]], 1);
            if (fl & 0x40) atomicSub(&T.size[T.bb[q]], 1);
            T.bflag[q] = (unsigned char)(fl & 0x1f);
        }
    }
#ifdef RC_PROF_SIM
    if (blockIdx.x == 0 && lane == 0) {
        long long *dbg_ = (long long *)((char *)V.work[0] + 64);
        dbg_[0] += 1; dbg_[1] += pf_init_; dbg_[2] += pf_pro_; dbg_[3] += pf_loop_; dbg_[4] += pf_epi_; dbg_[5] += nvisited; dbg_[6] += pf_b_; dbg_[7] += pf_d_; dbg_[8] += pf_ch_;
        dbg_[9] += nb0; dbg_[10] += __builtin_amdgcn_s_memrealtime() - pf0_;
        dbg_[11] += pc_unc_; dbg_[12] += pc_unc_t_; dbg_[13] += pc_ev_; dbg_[14] += pc_ev_t_;
    }
#endif
    if (lane == 0) { T.misc[3] = nb; T.misc[4] = hi; T.misc[5] = fail; T.misc[8] = nbirth; T.misc[9] = neff; T.misc[10] = first_eff; T.misc[13] = nvisited; T.misc[18] = 1; }
}

// Commit of the first `nc` batch changers: sizes, labels, cluster count, per-slot constants, slot_of, and the S
// corrections S[a][i] -= x[i*,i], S[b][i] += x[i*,i] (exact) for every point this block owns, in the generation it is
// reading (plain read-modify-write: nobody else touches those words) and in the NEXT generation, which the concurrently
// running row reduction of the following sweep is filling from the pre-change labels (64-bit integer atomics commute
// with its own, so the sum is exact whatever the interleaving).  A row of a cluster that dies ends as exact zeros, a
// new cluster's row starts from zeros (invariant: rows of free slots are zero).  mcmc.jl:250-252 plus the bookkeeping.
// Returns the number of label changes among the nc entries.
__device__ __forceinline__ int commit_batch(const View &V, const SweepArgs &sa, Tab &T, int nc, int G, int own_gen, int next_gen)
{
#ifdef RC_PROF_COMMIT
    const long long pcA_ = __builtin_amdgcn_s_memrealtime();
#endif
    if (threadIdx.x == 0) { T.misc[11] = 0; T.misc[12] = 0; if (nc) T.misc[0] = T.bK[nc - 1]; }
    __syncthreads();
    // one thread per entry: a slot's label is written by at most one entry of a batch (a cluster dies, is born or is
    // relabelled at most once), sizes by LDS atomics
    // With fewer than 2048 slots the label bitset T.used is the committed one at this point (built before the round's simulation,
    // which only reads it): the commit then corrects it in place — the labels its entries free here, the labels they take after
    // the barrier (a label freed by one entry can be taken by a later one) — instead of rebuilding it from all the labels.
    const bool inc_used = V.kcap < 2048;
    for (int q = threadIdx.x; q < nc; q += blockDim.x) {
        const int a = T.ba[q], b = T.bb[q], flag = T.bflag[q];
        if (a != b) { atomicSub(&T.size[a], 1); atomicAdd(&T.size[b], 1); }
        if (inc_used && (flag & (RC_BF_DEATH | RC_BF_RENAME))) { const int ol = T.label[a]; atomicAnd(&T.used[(ol - 1) >> 5], ~(1u << ((ol - 1) & 31))); }
        if (flag & RC_BF_DEATH) T.label[a] = 0;
        if (flag & RC_BF_BIRTH) { T.label[b] = T.blab[q]; atomicMax(&T.misc[7], b + 1); }
        if (flag & RC_BF_RENAME) T.label[a] = T.blab[q];
        if (flag & (RC_BF_DEATH | RC_BF_BIRTH | RC_BF_RENAME)) T.misc[11] = 1;
        if (!(flag & RC_BF_NOOP)) { atomicAdd(&T.misc[12], 1); T.dirty[a] = 1; T.dirty[b] = 1; }
    }
    __syncthreads();
#ifdef RC_PROF_COMMIT
    const long long pc0_ = __builtin_amdgcn_s_memrealtime();
#endif
    if (T.misc[11] && inc_used) tab_after_commit(V, sa, T, nc);
    else {
        if (T.misc[11]) tab_structural(V, T);
        tab_bases(V, sa, T);
    }
#ifdef RC_PROF_COMMIT
    const long long pcB_ = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { T.misc[13] = (int)(pcB_ - pc0_); T.misc[19] = (int)(pc0_ - pcA_); }
#endif
    const int nchunks = (V.n + RC_PTS - 1) / RC_PTS;
    const int pt = threadIdx.x & (RC_PTS - 1), st = threadIdx.x >> RC_PTS_LOG2, NS = blockDim.x >> RC_PTS_LOG2;
    long long *SDo = V.SD[own_gen], *SLo = V.SL[own_gen];
    long long *SDn = next_gen >= 0 ? V.SD[next_gen] : nullptr, *SLn = next_gen >= 0 ? V.SL[next_gen] : nullptr;
    const int hi_slots = T.misc[7];   // slots in use, births of this batch included
    const int hot = T.misc[16];        // the batch's largest group: its correction is shared by the streams of a point (hot_accumulate)
    long long *const hotacc = (long long *)T.red_v + 4 * pt;
    for (int c = blockIdx.x, m = 0; c < nchunks; c += G, ++m) {
        const int io = c * RC_PTS + pt;
        const bool act = io < V.n;
        const int i = act ? (T.cached ? T.cu[m * RC_PTS + pt] : V.pi[io]) : 0;
        if (hot >= 0) {
            if (threadIdx.x < 4 * RC_PTS) ((long long *)T.red_v)[threadIdx.x] = 0;
            __syncthreads();
            if (act) hot_accumulate(V, T, hot, i, nc, st, NS, hotacc);
            __syncthreads();
        }
        // Slot by slot (the groups of T.pairs): the net correction of the first nc entries, then ONE read-modify-write of the
        // generation being read (plain: every (slot row, point) belongs to one thread — stream slot mod NS — and this block
        // reads these rows again next round through its L1) and one atomic pair on the next generation, which the row reduction
        // of the following sweep is adding to concurrently.  (Entry by entry this was a chain of dependent round trips per
        // entry: 118 µs per round when a hundred changers joined one cluster.)
        for (int k = st; act && k < hi_slots; k += NS) {
            if ((k ? T.seg[k - 1] : 0) == T.seg[k]) continue;
            // (the words to correct are requested before the group is walked: their round trip then runs beside the matrix entries'
            // instead of behind it — the commit is a chain of dependent global round trips, not of instructions)
            const size_t ik = (size_t)k * V.ld + i;
            const long long oD = SDo[ik], oL = SLo[ik];
            long long dD = 0, dL = 0;
            int sz_ = 0, lab_ = 0;
            if (k == hot) { if (((const int *)hotacc)[5] == 0) continue; dD = hotacc[0]; dL = hotacc[1]; }
            else if (!batch_corr(V, T, k, i, nc, dD, dL, sz_, lab_)) continue;
            if (dD) { SDo[ik] = oD + dD; if (SDn) __hip_atomic_fetch_add((u64 *)(SDn + ik), (u64)dD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            if (dL) { SLo[ik] = oL + dL; if (SLn) __hip_atomic_fetch_add((u64 *)(SLn + ik), (u64)dL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
        if (hot >= 0) __syncthreads();
    }
#ifdef RC_PROF_COMMIT
    __syncthreads();
    if (threadIdx.x == 0) T.misc[20] = (int)(__builtin_amdgcn_s_memrealtime() - pcB_);
#endif
    for (int q = threadIdx.x; q < nc; q += blockDim.x)
        if (T.ba[q] != T.bb[q]) {
            V.slot_of[T.bu[q]] = T.bb[q];  // same values from every block
            if (T.cached) {                 // ... and this block's copy of the slots of its own points
                const int x = T.bx[q], cx = x >> RC_PTS_LOG2, d = cx - (int)blockIdx.x;
                if (d >= 0 && d % G == 0) T.cown[(d / G) * RC_PTS + (x & (RC_PTS - 1))] = T.bb[q];
            }
        }
    __syncthreads();
    return T.misc[12];
}

// ---------------------------------------------------------------------------------------------------
// k_resolve — one persistent launch per sweep.  Each round:
//   1. (round 0, and after a batch that did not reach the end of the sweep) every open point is scored and drawn in parallel
//      under the committed state: the guesses; barrier.  Otherwise the guesses are the draws of the previous round's step 3;
//   2. every block assembles the same ordered batch of guessed changers (up to the batch capacity) and simulates it in order
//      (batch_sim: moves, births, deaths, renames — label-keyed noise makes structural changes batchable);
//   3. every open point of the batch's range is drawn again under the committed state plus the batch entries that precede it;
//      the smallest point whose draw differs from what the batch says it does is the first violation; every draw is announced
//      as the point's guess for the next round;
//   4. barrier; the batch entries before the violation are committed at once.
// By induction over the point order this is exactly the sequential sweep: a point before the first violation saw
// precisely the changes of its predecessors.  (resolve_body says more about the guesses.)
// G blocks (G <= #CUs; they need not start together: the spin is bounded only by a generous timeout, and k_bulk
// never waits on this kernel).  Sweep t reads S generation own_gen, corrects own_gen and next_gen, uses key /
// chunk-word / barrier generation t%2 (and re-arms generation (t+1)%2), and leaves perm / snapshot generation t%2
// describing the labels after the sweep.
// ---------------------------------------------------------------------------------------------------
// The resolver proper (k_resolve is its launch wrapper).  smem: tab_bytes(kcap, n, blockDim.x / 64) bytes of LDS, at
// least 2·kcap ints.  All G blocks must be resident.
// Chaos builds (-DRC_CHAOS=sites): block-dependent pseudo-random delays at chosen points of the resolver's rounds, to shake
// out orderings between blocks that the usual lock-step timing hides (tools/chaos.sh).  Bit k of RC_CHAOS enables site k.
#ifdef RC_CHAOS
__device__ __forceinline__ void chaos_delay(int site, int round)
{
    if (!((RC_CHAOS >> site) & 1)) return;
    unsigned h = (unsigned)blockIdx.x * 2654435761u ^ (unsigned)round * 40503u ^ (unsigned)site * 97u;
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    const unsigned reps = h & 63u;                       // 0..63 x ~0.5 µs
    for (unsigned q = 0; q < reps; ++q) __builtin_amdgcn_s_sleep(16);
}
#define RC_CHAOS_AT(site) chaos_delay(site, round)
#else
#define RC_CHAOS_AT(site)
#endif

__device__ __forceinline__ void resolve_body(const View &V, const SweepArgs &sa, int G, char *smem)
{
    // The resolver is the latency-critical part of a sweep and shares its SIMDs with the waves of the other stream's row
    // reduction (three of those and one of these per SIMD): it takes instruction-issue priority over them — without it a
    // stationary pass that takes 11 µs alone took 40 µs beside the reduction.
#ifndef RC_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    RC_PF(long long ps[16]; for (int q_ = 0; q_ < 16; ++q_) ps[q_] = 0; ps[0] = __builtin_amdgcn_s_memrealtime(); long long pt_ = ps[0];)
#ifdef RC_PROF_ROUND   // absolute stamps of ONE round (number RC_PROF_ROUND) instead of per-phase sums over all rounds
#define RC_PHASE(k) RC_PF({ if (round == RC_PROF_ROUND) ps[k] = __builtin_amdgcn_s_memrealtime(); })
#else
#define RC_PHASE(k) RC_PF({ const long long now_ = __builtin_amdgcn_s_memrealtime(); ps[k] += now_ - pt_; pt_ = now_; })
#endif
    // (the error word is requested together with the first wave of the prologue's loads and tested before anything is stored:
    // a test of its own was a global round trip — ~3 us beside the streaming row reduction — in front of everything else)
    const int err_word = __hip_atomic_load(&V.sc->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int sc_changes = V.sc->n_changes, sc_last = V.sc->last_change_sweep, sc_rounds = V.sc->n_rounds;   // (read before the first barrier: the epilogue rewrites them)
    Tab T = tab_carve(smem, V.kcap, V.n, blockDim.x >> 6, V.maxb);
    if (V.n > RC_USED_LDS_MAX_N) T.used = V.used_scratch + (size_t)blockIdx.x * (size_t)((V.n + 31) / 32);
    const int t = sa.t, own_gen = sa.own_gen, next_gen = sa.next_gen, kg = t & 1;
    const long long *SD = V.SD[own_gen], *SL = V.SL[own_gen];
    u64 *keys = V.keys[kg];
    u64 *const cword_gen = V.cword[kg];
    unsigned *arrive = V.arrive[kg];
    {
        // Prologue: the slot tables (tab_load), their score constants (tab_bases) and — this block's chunks being c = blockIdx.x + m G,
        // few per block — the internal indices and slots of its points, in ONE loop whose global loads go out in two waves: sizes,
        // labels, act list and pi[i] first (with the error word and the previous sweep's counts), then what depends on them
        // (A[size], A[size - 1], slot_of[pi[i]]).  As separate passes these were six dependent round trips: 9-10 us of a 33 us
        // stationary resolver.
        const int nchunks_ = (V.n + RC_PTS - 1) / RC_PTS, cpb = (nchunks_ + G - 1) / G;
        T.cached = V.cu_cache != 0 && cpb <= RC_CPB_LDS && V.kcap < 2048;
        const int ncu = T.cached ? cpb * RC_PTS : 0, kend = max(V.kcap, ncu);
        int hK = 0, hse = 0, hhi = 0;
        if (threadIdx.x == 0) { hK = V.sc->K; hse = V.sc->smallest_empty; hhi = V.sc->slot_hi; }
        double2 fl_ = {0.0, 0.0};
        if (threadIdx.x < 128) fl_ = V.flt[threadIdx.x];
        bool first = true;
        for (int k = threadIdx.x; first || k < kend; k += blockDim.x) {
            const bool isk = k < V.kcap;
            int sz = 0, lb = 0, ac = 0, u_ = 0, o_ = 0, i_ = V.n;
            if (isk) { sz = V.slot_size[k]; lb = V.slot_label[k]; ac = V.slot_act[k]; }
            if (k < ncu) {
                i_ = ((int)blockIdx.x + (k >> RC_PTS_LOG2) * G) * RC_PTS + (k & (RC_PTS - 1));
                if (i_ < V.n) u_ = V.pi[i_];
            }
            // A sweep that ran out of slots (or whose barrier timed out) left the error bit set: the sweeps enqueued behind it must not
            // touch the state — the host grows the tables, resumes that sweep and replays these (recover_capacity).  The bit was set by
            // a previous launch (resolvers are chained), so every block of this launch reads the same value.  Nothing was stored yet.
            if (first && err_word != 0) return;
            first = false;
            double a1 = 0.0, a2 = 0.0, a3 = 0.0;
            if (isk && lb > 0) { a1 = V.A[sz]; if (sz >= 2) a2 = V.A[sz - 1]; if (V.kcap < 2048 && sz + 1 <= V.n) a3 = V.A[sz + 1]; }
            if (i_ < V.n) o_ = V.slot_of[u_];
            if (isk) {
                T.size[k] = sz; T.label[k] = lb; T.act[k] = (short)ac; T.dirty[k] = 0;
                if (lb > 0) {   // (tab_base: A[s] + (log p + log(s - 1 + r)))
                    T.base_o[k] = a1 + (sa.logp + log((double)sz - 1.0 + sa.r));
                    T.base_s[k] = (sz >= 2) ? a2 + (sa.logp + log((double)(sz - 1) - 1.0 + sa.r)) : 0.0;
                    if (V.kcap < 2048) T.base_p[k] = (sz + 1 <= V.n) ? a3 + (sa.logp + log((double)(sz + 1) - 1.0 + sa.r)) : 0.0;
                }
            }
            if (k < ncu) { T.cu[k] = u_; T.cown[k] = (short)o_; }
        }
        if (threadIdx.x < 128) T.flt[threadIdx.x] = fl_;
        if (V.kcap < 2048 && threadIdx.x == 128 % blockDim.x) T.base_p[V.kcap] = tab_base(V, sa, 1);   // a cluster of one: the births of a batch as candidates
        if (threadIdx.x == 0) {
            T.misc[0] = hK; T.misc[1] = hse; T.misc[7] = hhi;
            *T.blk_key = RC_KEY_NONE;
#ifdef RC_PROF_EVAL
            T.misc[19] = 0; T.misc[20] = 0; T.misc[21] = 0;
#endif
        }
        __syncthreads();
    }
    // Score cache: filling it costs 8 B per (point, cluster) in the first pass — 1.5 % of a stationary sweep at n = 8192, which
    // has no later pass to profit from it — so it is filled only when the previous sweep changed labels.  Same results either way.
    const bool use_wc = V.wc != nullptr && sa.dbg == 0 && (V.wc_always || sc_changes > 0);
    const int last = sc_last;
    const int prev_rounds = sc_rounds;     // rounds of the previous sweep = the key words it used (its generation is re-armed in the epilogue)
    RC_PF(ps[1] = __builtin_amdgcn_s_memrealtime();)
    const int nchunks = (V.n + RC_PTS - 1) / RC_PTS;
    int after = sa.after0, round = 0, changes = sa.changes0, nbar = 0;
    int cap = V.maxb;    // changers taken into the next batch (adaptive, identical in every block)
    bool ok = true;
    // Guesses.  A round validates GUESSES of what every open point does — any guess will do for correctness: a point is final
    // only once its draw under the exact sequential state (committed state + the batch entries before it, all of them validated)
    // equals its guess.  Round 0 guesses by drawing every point under the committed state (the tentative pass).  A later round
    // takes the draws of the previous round's VALIDATION as guesses, provided that batch reached the end of the sweep: they
    // already account for the changers before each point (all but the one that was violated), so they are better guesses than
    // a fresh tentative pass — the round after a violation usually goes through whole — and they cost nothing: no tentative
    // pass and no barrier before the batch is assembled, one grid barrier per round instead of two.
    // The announcements (chunk words, changer records) are double-buffered: a round reads buffer `cur` (words stamped `gstamp`)
    // and its validation writes the other one, so a block that is ahead never overwrites what a block behind still reads.
    // `exact`: every guess after `after` was drawn under the committed state as it is now (true after the tentative pass and
    // after a validation under a batch without effective entries): then the points up to the first effective changer are
    // final as they are, and a batch without effective entries ends the sweep.
    int cur = 0;
    unsigned gstamp = 0u;
    bool exact = true, used_current = false;
    // `redraw`: the guesses of this round are drawn afresh under the committed state (round 0; and after a batch that did not reach
    // the end of the sweep — more changers than a batch takes, or a cut: the points behind it carry guesses nobody has validated
    // since the state they were drawn under, and in the sweeps where this happens round after round — the first ones from a poor
    // labelling, thousands of changers — such guesses are nearly always wrong: 203 rounds instead of 102 for the first sweep from
    // random labels when they were kept).
    bool redraw = true, first_pass = true;
    while (ok) {
        u64 *const cword = cword_gen + (size_t)cur * (size_t)(nchunks + 1), *const cword_next = cword_gen + (size_t)(cur ^ 1) * (size_t)(nchunks + 1);
        unsigned *const rec = V.rec + (size_t)cur * (size_t)V.n, *const rec_next = V.rec + (size_t)(cur ^ 1) * (size_t)V.n;
        if (redraw) {
            RC_CHAOS_AT(0);
#if defined(RC_PROF_COMMIT) || defined(RC_PROF_EVAL)
            RC_PF(pt_ = __builtin_amdgcn_s_memrealtime();)
#else
            RC_PF(pt_ = __builtin_amdgcn_s_memrealtime(); ps[13] += 1;)
#endif
            gstamp += 1u;   // (words carried into this buffer are void: a chunk without changers writes none)
            // with the score cache a later pass computes only the slots a commit touched, and the cache is current again after it
            if (use_wc && !first_pass) tab_partition(V, T, false);
            for (int c = blockIdx.x, m = 0; c < nchunks; c += G, ++m)
                if (c * RC_PTS + RC_PTS - 1 > after) eval_chunk(V, sa, T, SD, SL, c, m, after, V.n, 0, 0, cword, rec, gstamp, use_wc ? (first_pass ? 1 : 2) : 0);
            if (use_wc && !first_pass) { for (int k = threadIdx.x; k < V.kcap; k += blockDim.x) T.dirty[k] = 0; }
            if (sa.dbg & 2) break;
            RC_PHASE(6)
#ifndef RC_PROF_COMMIT
            RC_PF(if (first_pass) ps[2] = __builtin_amdgcn_s_memrealtime();)
#endif
            ok = grid_barrier(V, T, arrive, G, (unsigned)(++nbar), RC_KEY_NONE, keys + round);
            RC_PF(if (first_pass) ps[3] = __builtin_amdgcn_s_memrealtime();)
            RC_PHASE(7)
            if (!ok) break;
            exact = true; redraw = false; first_pass = false;
        }
        RC_CHAOS_AT(1);
        RC_PF(pt_ = __builtin_amdgcn_s_memrealtime();)
        // 1. the ordered batch of guessed changers after `after` (the bits of the points up to `after` in its chunk are history)
        int any = 0;
        for (int c = threadIdx.x; c <= nchunks; c += blockDim.x) {
            int cnt = 0;
            if (c < nchunks && c * RC_PTS + RC_PTS - 1 > after) {
                const u64 w = __hip_atomic_load(cword + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(w >> 32) == gstamp) {
                    unsigned m = (unsigned)w;
                    if (c * RC_PTS <= after) m &= ~0u << (after - c * RC_PTS + 1);
                    cnt = __popc(m);
                }
            }
            T.ccnt[c] = (unsigned short)cnt;
            any |= cnt;
        }
        for (int k = threadIdx.x; k <= V.kcap; k += blockDim.x) { T.seg[k] = 0; if (k < V.kcap) T.joined[k] = 0; }
        if (threadIdx.x == 0) T.misc[2] = 0;
        __syncthreads();
        if (any) T.misc[2] = 1;
        __syncthreads();
        const int any_all = T.misc[2];
        __syncthreads();
        if (!any_all && exact) break;  // no point wants to move, and every one of them drew under the state as it is: the sweep is complete (the stationary fast path)
        // exclusive offsets by wave 0: every lane scans its run of chunks, the lanes' totals by shuffles (a scan by one
        // thread costs two LDS round trips per chunk: 14 µs at n = 8192, half of a stationary resolver pass)
        if (threadIdx.x < 64) {
            const int per = (nchunks + 1 + 63) / 64, c0_ = (int)threadIdx.x * per, c1_ = min(c0_ + per, nchunks + 1);
            int sum = 0;
            for (int c = c0_; c < c1_; ++c) sum += T.ccnt[c];
            int incl = sum;
#pragma unroll
            for (int d_ = 1; d_ < 64; d_ <<= 1) {
                const int up = __shfl_up(incl, d_);
                if ((int)threadIdx.x >= d_) incl += up;
            }
            int o = incl - sum;
            for (int c = c0_; c < c1_; ++c) { const int x = T.ccnt[c]; T.ccnt[c] = (unsigned short)min(o, 65535); o += x; }
        }
        __syncthreads();
        const int total = T.ccnt[nchunks];
        // one thread per batch entry (and one for the first changer that does not fit): its chunk by bisection of the offsets, its
        // point from the rank of its bit in the chunk's mask, then ONE record load — all loads of a batch in flight together (a
        // thread per chunk fetched the records of its up to 32 changers one after the other: 4.6 µs per round in the moving
        // regime, 11.5 when every point moves)
        for (int o = threadIdx.x; o < min(total, cap + 1); o += blockDim.x) {
            int lo_ = 0, hi_ = nchunks;                      // the largest c with ccnt[c] <= o (then ccnt[c + 1] > o)
            while (hi_ - lo_ > 1) { const int mid = (lo_ + hi_) >> 1; if ((int)T.ccnt[mid] <= o) lo_ = mid; else hi_ = mid; }
            const int c = lo_;
            unsigned m = (unsigned)__hip_atomic_load(cword + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c * RC_PTS <= after) m &= ~0u << (after - c * RC_PTS + 1);
            for (int r_ = o - (int)T.ccnt[c]; r_ > 0; --r_) m &= m - 1;
            const int x = c * RC_PTS + __ffs((int)m) - 1;
            if (o < cap) {
                const unsigned rc = __hip_atomic_load(rec + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int own_ = (int)(rc >> 16), tgt_ = (int)(rc & 0xFFFFu) - 1;
                T.bx[o] = x; T.bu[o] = V.pi[x]; T.ba[o] = (short)own_;
                if (tgt_ >= 0 && T.size[tgt_] == 0) {
                    // a carried-over guess whose target is not a cluster of the committed state (born by an entry that was not
                    // committed, or emptied since): the entry stays as a placeholder that does nothing, the guess becomes "stays"
                    T.bb[o] = (short)own_;
                } else {
                    T.bb[o] = (short)tgt_;
                    if (tgt_ >= 0) T.joined[tgt_] = 1;
                    atomicAdd(&T.seg[own_], 1);   // leaves its cluster
                }
            } else {
                T.misc[2] = x;  // first changer that does not fit into the batch
            }
        }
        __syncthreads();
        // what every changer does when the batch is applied in order.  The label bitset (for the labels of births) is
        // scratch of batch_sim: rebuilt from the committed labels every time
        if (total > 0 && !used_current) {
            const int nw_ = (V.n + 31) / 32;
            for (int w = threadIdx.x; w < nw_; w += blockDim.x) T.used[w] = 0u;
            __syncthreads();
            for (int k = threadIdx.x; k < T.misc[7]; k += blockDim.x) {
                const int lab = T.label[k];
                if (lab > 0) atomicOr(&T.used[(lab - 1) >> 5], 1u << ((lab - 1) & 31));
            }
            __syncthreads();
        }
        // (with the simulation's state in registers — kcap < 2048 — the bitset is only read there, and a commit that changes a
        // label rebuilds it from the committed labels, tab_structural: it stays current for the rest of the sweep)
        if (total > 0) used_current = V.kcap < 2048;
        RC_PHASE(8)
        // clusters that could become empty inside the batch (more leavers than would leave one member): only their sizes matter
        for (int k = threadIdx.x; k < V.kcap; k += blockDim.x) T.candie[k] = (T.size[k] - T.seg[k] < 1);
        __syncthreads();
#ifdef RC_SIM_OLD   // (comparison builds: the one-entry-at-a-time simulation everywhere)
        if (threadIdx.x < 64) batch_sim(V, T, total, cap);
#else
        if (threadIdx.x < 64) { if (V.kcap < 2048) batch_sim_fast(V, T, total, cap); else batch_sim(V, T, total, cap); }
#endif
        if (threadIdx.x == 0) *T.blk_key = RC_KEY_NONE;
        __syncthreads();
        RC_PHASE(14)
#if !defined(RC_PROF_COMMIT) && !defined(RC_PROF_EVAL)
        RC_PF(ps[15] += T.misc[13];)
#endif
        // the simulated sizes back to the committed ones (only clusters that could die, and new ones, were simulated), and
        // the entries that touch each slot, grouped by slot: count, offsets, scatter, sort within a slot
        {
            const int nbk = T.misc[3];
            const bool sizes_restored = T.misc[18] != 0;   // (batch_sim_fast puts the sizes back itself)
            for (int k = threadIdx.x; k <= V.kcap; k += blockDim.x) T.seg[k] = 0;
            __syncthreads();
            for (int q = threadIdx.x; q < nbk; q += blockDim.x) {
                const int a_ = T.ba[q], b_ = T.bb[q], fl = T.bflag[q];
                if (a_ != b_ && !sizes_restored) {
                    if (T.candie[a_]) atomicAdd(&T.size[a_], 1);
                    if ((fl & RC_BF_BIRTH) || T.candie[b_]) atomicSub(&T.size[b_], 1);
                }
                if (!(fl & RC_BF_NOOP)) { atomicAdd(&T.seg[a_], 1); if (b_ != a_) atomicAdd(&T.seg[b_], 1); }
            }
            __syncthreads();
            if (threadIdx.x < 64) {   // exclusive offsets over the slots (wave 0: per-lane runs + shuffle scan)
                const int per = (V.kcap + 1 + 63) / 64, c0_ = (int)threadIdx.x * per, c1_ = min(c0_ + per, V.kcap + 1);
                int sum = 0;
                for (int k = c0_; k < c1_; ++k) sum += T.seg[k];
                int incl = sum;
#pragma unroll
                for (int d_ = 1; d_ < 64; d_ <<= 1) {
                    const int up = __shfl_up(incl, d_);
                    if ((int)threadIdx.x >= d_) incl += up;
                }
                int o = incl - sum;
                for (int k = c0_; k < c1_; ++k) { const int x = T.seg[k]; T.seg[k] = o; o += x; }
            }
            __syncthreads();
            for (int q = threadIdx.x; q < nbk; q += blockDim.x) {
                const int a_ = T.ba[q], b_ = T.bb[q];
                if (!(T.bflag[q] & RC_BF_NOOP)) {
                    T.pairs_tmp[atomicAdd(&T.seg[a_], 1)] = (short)q;
                    if (b_ != a_) T.pairs_tmp[atomicAdd(&T.seg[b_], 1)] = (short)q;
                }
            }
            if (threadIdx.x == 0) T.misc[17] = 0;
            __syncthreads();   // now seg[k] = end of slot k's group, seg[k-1] (0 for k = 0) its begin
            for (int k = threadIdx.x; k < V.kcap; k += blockDim.x) {   // the largest group, if it is worth sharing (hot_accumulate)
                const int sz_ = T.seg[k] - (k ? T.seg[k - 1] : 0);
                if (sz_ >= RC_HOT_MIN) atomicMax(&T.misc[17], (sz_ << 16) | k);
            }
            __syncthreads();
            if (threadIdx.x == 0) T.misc[16] = T.misc[17] ? (T.misc[17] & 0xFFFF) : -1;
            // order every group by entry index: each entry ranks itself inside the groups of its two slots (a group can hold
            // hundreds of entries when the changers of a batch share a cluster: a one-thread-per-slot sort took 335 µs there)
            for (int q = threadIdx.x; q < nbk; q += blockDim.x) {
                if (T.bflag[q] & RC_BF_NOOP) continue;
                const int a_ = T.ba[q], b_ = T.bb[q];
                for (int side = 0; side < 2; ++side) {
                    const int k = side ? b_ : a_;
                    if (side && b_ == a_) break;
                    const int e0 = k ? T.seg[k - 1] : 0, e1 = T.seg[k];
                    int rank = 0;
                    for (int e = e0; e < e1; ++e) rank += (T.pairs_tmp[e] < q);
                    T.pairs[e0 + rank] = (short)q;
                }
            }
        }
        __syncthreads();
        const int nb = T.misc[3], hi = T.misc[4], neff = T.misc[9];
        RC_PHASE(9)
        RC_CHAOS_AT(2);
#ifdef RC_TRACE_RESOLVE   // diagnostic builds: per-round record of block RC_TRACE_BLOCK (default 0) behind the work counter
        if ((int)blockIdx.x == (sa.dbg >> 8) && threadIdx.x == 0 && round < 120) {   // kept in LDS until the sweep is over
            int *tr = (int *)(smem + tab_bytes_dev(V.kcap, V.n, blockDim.x >> 6, V.maxb)) + (size_t)round * 8;
            tr[0] = round; tr[1] = total; tr[2] = nb; tr[3] = hi; tr[4] = nb ? T.bx[0] : -1; tr[5] = after; tr[6] = neff; tr[7] = T.misc[8] | (exact ? 0x10000 : 0);
        }
#endif
        if (T.misc[5]) {   // the first changer needs a slot and every slot is taken
            // Everything up to `after` is final — with exact guesses also the points between it and this changer (they drew their
            // own labels under the committed state) — and the state is consistent: the host grows the slot tables and resumes
            // this sweep behind that point (with a tentative pass).
            if (threadIdx.x == 0 && blockIdx.x == 0) {
                const int ra = exact ? T.bx[0] - 1 : after;
                V.sc->fail_t = t; V.sc->resume_after = ra; V.sc->fail_changes = changes; V.sc->fail_rounds = sa.rounds0 + round + 1;
                V.hsum->fail_t = t; V.hsum->resume_after = ra; V.hsum->fail_changes = changes; V.hsum->fail_rounds = sa.rounds0 + round + 1;
                atomicOr(&V.sc->err, RC_DERR_CAPACITY);
            }
            ok = false;
            break;
        }
        if (neff == 0 && exact) {
            // only singletons that drew "new cluster" and keep their labels: nothing to validate or commit, and the guesses
            // behind the batch are still exact (nothing changed)
            if (hi == V.n - 1) break;
            after = hi;
            ++round;
            continue;
        }
        // 2. validation of the open points of the batch's range, each under the changers that precede it.  With exact guesses the
        // points up to the first effective changer saw no change at all and are final as they are.
        const int vlo = (exact && neff > 0) ? T.bx[T.misc[10]] : after;
        if (use_wc) tab_partition(V, T, true);
        for (int c = blockIdx.x, m = 0; c < nchunks; c += G, ++m) {
            if (c * RC_PTS + RC_PTS - 1 > vlo && c * RC_PTS <= hi)
                eval_chunk(V, sa, T, SD, SL, c, m, vlo, hi, 1, nb, cword, rec, gstamp, use_wc ? 2 : 0, cword_next, rec_next, gstamp + 1u);
        }
        __syncthreads();
#ifdef RC_PROF_EVAL   // (this build: columns 2 / 13 / 15 = longest thread in the cached-slot part / computed part / rest of the validation passes)
        RC_PF(ps[2] += T.misc[19]; ps[13] += T.misc[20]; ps[15] += T.misc[21];)
        __syncthreads();
        if (threadIdx.x == 0) { T.misc[19] = 0; T.misc[20] = 0; T.misc[21] = 0; }
#endif
        RC_PHASE(10)
        const u64 mine = *T.blk_key;
        ok = grid_barrier(V, T, arrive, G, (unsigned)(++nbar), mine, keys + round);
        RC_PHASE(11)
        if (!ok) break;
        RC_CHAOS_AT(3);
        // 3. commit the changers before the first violation
        const u64 vk = __hip_atomic_load(keys + round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int limit = (vk == RC_KEY_NONE) ? hi + 1 : (int)vk;   // points < limit are final
        int nc = 0;
        {
            int lo_ = 0, hi_ = nb;
            while (lo_ < hi_) { const int mid = (lo_ + hi_) >> 1; if (T.bx[mid] < limit) lo_ = mid + 1; else hi_ = mid; }
            nc = lo_;
        }
        changes += commit_batch(V, sa, T, nc, G, own_gen, next_gen);
        RC_PHASE(12)
#ifdef RC_PROF_COMMIT   // (this build: columns 2 / 13 / 15 = table rebuild / entry loop / row-sum corrections inside the commit)
        RC_PF(ps[2] += T.misc[13]; ps[13] += T.misc[19]; ps[15] += T.misc[20];)
#endif
        // Validation costs (points covered) x (changers before them); behind the first violation it only yields the next round's
        // guesses.  After a round that hit a violation the next batch takes one and a half times what was committed, at least
        // RC_CAP_FLOOR entries; after a round without one it doubles.  The floor is above the ~170 changers a sweep of the moving
        // regime of bench.py announces, so that its batches reach the end of the sweep and the guesses carry over (floor 128 / 256 /
        // 512: 4.73 / 4.84 / 4.86 k sweeps/s there, and 15.4 / 15.3 / 17.0 ms for the first sweep from random labels).
#ifndef RC_CAP_FLOOR
#define RC_CAP_FLOOR 256
#endif
        cap = (nc < nb) ? min(V.maxb, max(RC_CAP_FLOOR, nc + nc / 2)) : min(V.maxb, 2 * cap);
        after = limit - 1;
        cur ^= 1;
        gstamp += 1u;
        // the new guesses were drawn under the committed state plus the batch: exact if the batch changed nothing and covered every point
        exact = (neff == 0 && hi == V.n - 1);
        ++round;
        if (limit >= V.n) break;          // every point validated: the sweep is complete
        redraw = (hi < V.n - 1);          // the batch did not cover the rest of the sweep
        if (round > 2 * V.n + 4) break;   // cannot happen: the first open point's guess is exact at the latest in the round after it was violated
    }
#ifdef RC_TRACE_RESOLVE
    __syncthreads();
    if ((int)blockIdx.x == (sa.dbg >> 8)) {
        const int *tr = (const int *)(smem + tab_bytes_dev(V.kcap, V.n, blockDim.x >> 6, V.maxb));
        long long *out = (long long *)((char *)V.work[kg] + 64);
        for (int q = threadIdx.x; q < 120 * 8; q += blockDim.x) out[q] = (q / 8 <= round) ? tr[q] : -7;
    }
    __syncthreads();
#endif
    RC_PF(ps[4] = __builtin_amdgcn_s_memrealtime();)
    if (sa.zero_gen >= 0 && (blockIdx.x > 0 || G == 1)) {
        // The blocks that have no epilogue work clear the S generation that held the sums of the labels before this
        // sweep: nobody reads it any more, the row reduction of sweep t+2 fills it and k_resolve(t+1) adds its
        // corrections to it (both ordered after this launch).  Rows >= slot_hi are zero by invariant.
        const size_t total2 = (size_t)T.misc[7] * (size_t)V.ld / 2;
        const size_t nthreads = (size_t)(G > 1 ? G - 1 : 1) * blockDim.x;
        const size_t me = (size_t)(G > 1 ? blockIdx.x - 1 : 0) * blockDim.x + threadIdx.x;
        ll2 *zd = (ll2 *)V.SD[sa.zero_gen], *zl = (ll2 *)V.SL[sa.zero_gen];
        const ll2 z = {0, 0};
        for (size_t q = me; q < total2; q += nthreads) { zd[q] = z; zl[q] = z; }
    }
    // Epilogue.  Every block holds the same tables and has written the same slot_of, so the jobs are spread: re-arming, the
    // slot tables with the host summary, and the run count go to three different blocks, the label snapshot and the point
    // order of the next layout are built by all blocks together (one block doing everything in turn was a 42 µs tail on
    // every sweep that changed labels, 32 of them the order build).
    const int b_rearm = G > 1 ? 1 : 0, b_snap = G > 2 ? 2 : 0;
    __syncthreads();
    if ((int)blockIdx.x == b_rearm) {
        // the row reduction of this sweep is complete (stream order): re-arm its work counter for sweep t+2
        if (threadIdx.x == 0) *V.work[kg] = 0;
        // re-arm the other key / chunk-word / barrier generation for the next sweep (its last user, sweep t-1, is done)
        for (int q = threadIdx.x; q < min(prev_rounds + 2, 2 * V.n + 8); q += blockDim.x) V.keys[kg ^ 1][q] = RC_KEY_NONE;   // (the words sweep t-1 used)
        for (int q = threadIdx.x; q < 2 * (nchunks + 1); q += blockDim.x) V.cword[kg ^ 1][q] = 0;
        for (int q = threadIdx.x; q < RC_BAR_WORDS; q += blockDim.x) V.arrive[kg ^ 1][q] = 0u;
    }
    if (blockIdx.x == 0) {
        if (changes || t < 2) tab_store(V, T);   // (nothing committed: the tables in global memory are the ones that were loaded)
        if (threadIdx.x == 0) {
            V.sc->n_changes = changes;
            V.sc->n_rounds = sa.rounds0 + round + 1;
            if (changes) V.sc->last_change_sweep = t;
        }
        __syncthreads();
    }
    // label snapshot of this generation: k_bulk_sym of sweep t+2 reads it (as the perm generation below)
    if (changes || last == t - 1 || t < 2) {
        snapshot_copy_grid(V, kg, G);
        if ((int)blockIdx.x == b_snap) snapshot_runs(V, &T.misc[2]);
    }
    if (blockIdx.x == 0) write_summary(V, changes, sa.rounds0 + round + 1, changes != 0 || t < 2);
    // perm generation t%2 must describe the labels after this sweep (k_bulk of sweep t+2 reads it): all blocks
    if (changes && ok) {
        __syncthreads();
        build_perm_grid(V, kg, T.size, T.misc[7], G, (int *)smem);   // (the tables in LDS are no longer needed: T.size lies behind the 16·kcap bytes this uses)
    } else if (last == t - 1 && t >= 1) {
        for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < V.n; p += G * blockDim.x) { V.perm[kg][p] = V.perm[kg ^ 1][p]; V.pslot[kg][p] = V.pslot[kg ^ 1][p]; }
    }
    RC_PF(if (threadIdx.x == 0 && blockIdx.x < 256) { long long *o = (long long *)((char *)V.work[kg] + 64) + (size_t)(8192 - 256 + blockIdx.x) * 16;
                                                       ps[5] = __builtin_amdgcn_s_memrealtime(); for (int q = 0; q < 16; ++q) o[q] = ps[q]; })
}

// 128 VGPRs (four waves per SIMD): a 256-thread resolver block — one wave per SIMD — then fits on a CU beside THREE blocks
// of the wave-autonomous row reduction (3 x 128 + 128 = the SIMD's 512 registers; 3 x 40 KiB + 26 KiB of LDS), so the
// resolver of sweep t is resident at once and runs beside the row reduction of sweep t+1 instead of waiting for its
// blocks to retire (four reduction blocks take the whole CU: the two then serialised).
#ifndef RC_RES_MINWAVES
#define RC_RES_MINWAVES 4
#endif
__global__ __launch_bounds__(RC_RES_THREADS_MAX, RC_RES_MINWAVES) void k_resolve(View V, SweepArgs sa, int G)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    resolve_body(V, sa, G, smem);
}

// ===================================================================================================
// Wide contexts: more than RC_MAX_KCAP clusters.  The reference's state has room for n clusters (clustsizes of length n,
// src/types.jl:131-137; a new cluster is offered whenever maxK allows, src/mcmc.jl:198-199).  The resolver keeps its per-slot
// tables in LDS, which ends at 4096 slots; beyond, the context is WIDE: the slot tables stay in global memory, the row-sum table
// has one generation that is maintained in place (as in the incremental mode), and the sweep is the loop of the reference itself
// — one point after the other, src/mcmc.jl:192-252 — on ONE workgroup: its 1024 threads share the candidates of the point, the
// arg-max goes through shuffles and LDS, a move corrects the two S rows it touches (exact integers).  Slow — tens to hundreds of
// milliseconds per sweep — and far outside what the sampler is for (a chain among thousands of clusters is resolving births and
// deaths all the time), but the state space of the reference is covered up to RC_WIDE_MAX_KCAP clusters instead of ending at an
// error.  Same draws as every other path: the same score arithmetic (tab_base, log1p form), the same label-keyed uniforms, the
// same tie rule.
// ===================================================================================================
__device__ __forceinline__ int wide_block_min(int v, int *lds /* [17] */)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        int m = lds[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = min(m, lds[w]);
        lds[16] = m;
    }
    __syncthreads();
    return lds[16];
}

// smallest label > lab (1-based) whose bit is clear in the global bitset; n + 1 if none.  All threads.
__device__ __forceinline__ int wide_next_empty(const unsigned *used, int n, int lab, int *lds)
{
    const int nw = (n + 31) / 32;
    int best = n + 1;
    for (int w = (lab >> 5) + (int)threadIdx.x; w < nw; w += blockDim.x) {
        unsigned inv = ~used[w];
        if (w == (lab >> 5)) inv &= ~((1u << (lab & 31)) - 1u);
        if (inv) { const int r = w * 32 + __ffs((int)inv); if (r <= n) best = min(best, r); break; }   // (a thread's words ascend: its first hit is its smallest)
    }
    return wide_block_min(best, lds);
}

// After rc_set_state / apply_labels in a wide context: label bitset and smallest empty label, the label snapshot and run count,
// the host summary, and (rebuild_perm) the rows grouped by slot for the row reduction that fills the table.  One block.
__global__ __launch_bounds__(1024) void k_derive_wide(View V, int rebuild_perm)
{
    __shared__ int red[17];
    __shared__ int scan[1024];
    unsigned *used = V.wide_scratch;
    int *off = (int *)(V.wide_scratch + (V.n + 31) / 32), *cur = off + V.kcap + 1;
    const int nw = (V.n + 31) / 32, hi = V.sc->slot_hi;
    for (int w = threadIdx.x; w < nw; w += blockDim.x) used[w] = 0u;
    __threadfence_block(); __syncthreads();
    for (int k = threadIdx.x; k < hi; k += blockDim.x) {
        const int lab = V.slot_label[k];
        if (lab > 0) atomicOr(&used[(lab - 1) >> 5], 1u << ((lab - 1) & 31));
    }
    __threadfence_block(); __syncthreads();
    const int se = wide_next_empty(used, V.n, 0, red);
    if (threadIdx.x == 0) V.sc->smallest_empty = se;
    __syncthreads();
    snapshot_labels(V, 0, &red[0]);
    for (int i = threadIdx.x; i < V.n; i += blockDim.x) V.snap[1][i] = V.snap[0][i];
    __syncthreads();
    write_summary(V, 0, 0);
    __syncthreads();
    if (!rebuild_perm) return;
    // rows grouped by slot (k_bulk): counts, exclusive offsets (every thread scans a run of slots, the runs' totals through LDS), scatter
    for (int k = threadIdx.x; k <= V.kcap; k += blockDim.x) cur[k] = 0;
    __threadfence_block(); __syncthreads();
    for (int i = threadIdx.x; i < V.n; i += blockDim.x) atomicAdd(&cur[V.slot_of[i]], 1);
    __threadfence_block(); __syncthreads();
    const int per = (V.kcap + (int)blockDim.x - 1) / (int)blockDim.x, k0 = (int)threadIdx.x * per, k1 = min(k0 + per, V.kcap);
    int sum = 0;
    for (int k = k0; k < k1; ++k) sum += cur[k];
    scan[threadIdx.x] = sum;
    __threadfence_block(); __syncthreads();
    if (threadIdx.x == 0) { int o = 0; for (int q = 0; q < (int)blockDim.x; ++q) { const int x = scan[q]; scan[q] = o; o += x; } }
    __threadfence_block(); __syncthreads();
    int o = scan[threadIdx.x];
    for (int k = k0; k < k1; ++k) { off[k] = o; o += cur[k]; }
    __threadfence_block(); __syncthreads();
    for (int k = threadIdx.x; k <= V.kcap; k += blockDim.x) cur[k] = 0;
    __threadfence_block(); __syncthreads();
    int *perm = V.perm[0], *pslot = V.pslot[0];
    for (int i = threadIdx.x; i < V.n; i += blockDim.x) {
        const int sl = V.slot_of[i];
        const int p_ = off[sl] + atomicAdd(&cur[sl], 1);
        perm[p_] = i; pslot[p_] = sl;
    }
    __threadfence_block(); __syncthreads();
    for (int p_ = threadIdx.x; p_ < V.n; p_ += blockDim.x) { V.perm[1][p_] = perm[p_]; V.pslot[1][p_] = pslot[p_]; }
}

// One Gibbs sweep of a wide context (see above): src/mcmc.jl:192-252 point by point.  gen: the S generation that holds the row sums
// of the current labels (corrected in place).  One block of 1024 threads; the state lives in global memory and is shared between
// the threads of the block through it: every hand-over is a workgroup fence + barrier (RC_WIDE_SYNC — __syncthreads alone does
// not wait for global stores on this target).
#define RC_WIDE_SYNC() do { __threadfence_block(); __syncthreads(); } while (0)
__global__ __launch_bounds__(1024) void k_sweep_wide(View V, SweepArgs sa, int gen)
{
    __shared__ int red[17];
    __shared__ double rv[16];
    __shared__ int rk[16], rs[16];
    __shared__ int sh[2];
    if (__hip_atomic_load(&V.sc->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;   // (a sweep before this one ran out of slots: the host grows the tables and replays)
    unsigned *used = V.wide_scratch;
    long long *SD = V.SD[gen], *SL = V.SL[gen];
    const size_t ld = (size_t)V.ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
    int K = V.sc->K, se = V.sc->smallest_empty, hi = V.sc->slot_hi, changes = sa.changes0, fcur = 0;
    bool failed = false;
    for (int i = sa.after0 + 1; i < V.n && !failed; ++i) {
        // (everything read here was written before the last RC_WIDE_SYNC of the previous point)
        const int u = V.pi[i], own = V.slot_of[u];
        const int so = V.slot_size[own], la_own = V.slot_label[own], single = (so == 1);
        const int Ki = K - single;
        const long long dg = V.diagq[u];
        double bestv = -INFINITY;
        int bestkey = 0x7fffffff, bestslot = -2;
        for (int k = threadIdx.x; k < hi; k += blockDim.x) {
            const int lab = V.slot_label[k];
            if (lab == 0) continue;
            const int isown = (k == own);
            const int s_ = V.slot_size[k] - isown;
            if (s_ == 0) continue;                                               // its own singleton cluster (mcmc.jl:193-196)
            long long sd = SD[(size_t)k * ld + u], sl = SL[(size_t)k * ld + u];
            sd -= (isown ? dg : 0);                                              // i itself excluded (clusts[i] = -1)
            const double SDr = (double)sd * V.scD, SLr = (double)sl * V.scL;
            const double base = tab_base(V, sa, s_);
            double lik = V.cL * SLr - (V.alpha + V.delta1 * (double)s_) * rc_flog1p(SDr / V.beta, V.flt);
            if (V.repulsion) lik += (V.zeta + V.delta2 * (double)s_) * rc_flog1p(SDr / V.gamma, V.flt);
            double v = base + lik;
            const double un = rc_uniform(sa, (unsigned)i, (unsigned)lab);
            v = v + rc_gumbel(un, V.flt);
            if (bestslot == -2 || v > bestv || (v == bestv && lab < bestkey)) { bestv = v; bestkey = lab; bestslot = k; }
        }
        const bool new_ok = (V.maxK == 0 || (long long)Ki < V.maxK) && Ki < V.n;
        if (threadIdx.x == 0 && new_ok) {                                        // mcmc.jl:198-203, 228-230: last in the candidate order
            const double un = rc_uniform(sa, (unsigned)i, 0u);
            const double v = (log((double)(Ki + 1)) + sa.r * sa.log1mp) + rc_gumbel(un, V.flt);
            if (v > bestv || bestslot == -2) { bestv = v; bestkey = RC_NEWKEY; bestslot = -1; }
        }
#pragma unroll
        for (int off_ = 32; off_ > 0; off_ >>= 1) {
            const double ov = __shfl_xor(bestv, off_);
            const int ok_ = __shfl_xor(bestkey, off_), os = __shfl_xor(bestslot, off_);
            best_merge(bestv, bestkey, bestslot, ov, ok_, os);
        }
        __syncthreads();                                                         // (the previous point's readers of rv / sh are done)
        if (lane == 0) { rv[wave] = bestv; rk[wave] = bestkey; rs[wave] = bestslot; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double bv = rv[0]; int bk = rk[0], bs = rs[0];
            for (int w = 1; w < NW; ++w) best_merge(bv, bk, bs, rv[w], rk[w], rs[w]);
            sh[0] = bs;
        }
        __syncthreads();
        const int target = sh[0];      // a slot, -1 = new cluster, -2 = no candidate at all (alone and maxK forbids a new cluster): stays
        if (target == own || target == -2) continue;
        int b = target;
        if (target == -1 && single) {
            // a singleton that draws "new cluster": the smallest empty label once i is removed is min(its own, the smallest empty one)
            if (se < la_own) {                                                   // it takes the smaller label and frees its own (mcmc.jl:199)
                if (threadIdx.x == 0) {
                    V.slot_label[own] = se;
                    used[(se - 1) >> 5] |= 1u << ((se - 1) & 31);
                    used[(la_own - 1) >> 5] &= ~(1u << ((la_own - 1) & 31));
                }
                RC_WIDE_SYNC();
                se = wide_next_empty(used, V.n, se, red);                        // (the freed label la_own > se is among the candidates)
                ++changes;
            }
            continue;
        }
        if (target == -1) {
            // birth: the lowest free slot, label = the smallest empty label
            for (;;) {
                int mine = 0x7fffffff;
                const int k = fcur + (int)threadIdx.x;
                if (k < V.kcap && V.slot_label[k] == 0) mine = k;
                const int f = wide_block_min(mine, red);
                if (f != 0x7fffffff) { b = f; break; }
                fcur += (int)blockDim.x;
                if (fcur >= V.kcap) { b = -1; break; }
            }
            if (b < 0) {
                // every slot is taken: the points before i are final, the state is consistent — the host grows the tables and
                // resumes this sweep behind point i - 1 (recover_capacity)
                if (threadIdx.x == 0) {
                    V.sc->fail_t = sa.t; V.sc->resume_after = i - 1; V.sc->fail_changes = changes; V.sc->fail_rounds = sa.rounds0 + 1;
                    V.hsum->fail_t = sa.t; V.hsum->resume_after = i - 1; V.hsum->fail_changes = changes; V.hsum->fail_rounds = sa.rounds0 + 1;
                    atomicOr(&V.sc->err, RC_DERR_CAPACITY);
                }
                failed = true;
                continue;
            }
            if (threadIdx.x == 0) {
                V.slot_label[b] = se; V.slot_size[b] = 0;
                used[(se - 1) >> 5] |= 1u << ((se - 1) & 31);
            }
            RC_WIDE_SYNC();
            se = wide_next_empty(used, V.n, se, red);
            K += 1;
            hi = max(hi, b + 1);
        }
        // move u: own -> b.  S[own][j] -= X[u][j], S[b][j] += X[u][j] for every j (exact integers; the row of a cluster that dies
        // ends as exact zeros, a new cluster's row starts from zeros)
        for (int j = threadIdx.x; j < V.n; j += blockDim.x) {
            const size_t e = (size_t)u * ld + j;
            const long long xd = (V.bits == 64) ? ((const long long *)V.Dq)[e] : (long long)((const int *)V.Dq)[e];
            const long long xl = rc_load_L(V, u, j, xd);
            SD[(size_t)own * ld + j] -= xd; SD[(size_t)b * ld + j] += xd;
            SL[(size_t)own * ld + j] -= xl; SL[(size_t)b * ld + j] += xl;
        }
        if (threadIdx.x == 0) {
            V.slot_size[own] = so - 1; V.slot_size[b] += 1;
            V.slot_of[u] = b;
            if (single) {                                                        // the cluster dies: its label is free
                V.slot_label[own] = 0;
                used[(la_own - 1) >> 5] &= ~(1u << ((la_own - 1) & 31));
            }
        }
        if (single) { K -= 1; if (la_own < se) se = la_own; if (own < fcur) fcur = own - (own % (int)blockDim.x); }
        ++changes;
        RC_WIDE_SYNC();
    }
    RC_WIDE_SYNC();
    if (threadIdx.x == 0 && !failed) {
        V.sc->K = K; V.sc->smallest_empty = se; V.sc->slot_hi = hi;
        V.sc->n_changes = changes; V.sc->n_rounds = sa.rounds0 + 1;
        if (changes) V.sc->last_change_sweep = sa.t;
    }
    RC_WIDE_SYNC();
    if (failed) {
        // the host learns of the failure from the mapped summary (sync_and_check reads hsum->err, which only write_summary sets):
        // without this a wide context that ran out of slots AGAIN — it was grown to twice its capacity, not to n — dropped the sweep
        // and every sweep behind it silently (found by the large leg of the randomised checks, seed 69000: 2432 -> 4864 slots, then
        // K = 4924)
        if (threadIdx.x == 0) {
            V.hsum->err = __hip_atomic_load(&V.sc->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            V.hsum->seq += 1;
        }
        return;
    }
    snapshot_labels(V, sa.t & 1, &red[0]);
    write_summary(V, changes, sa.rounds0 + 1);
}

// ---------------------------------------------------------------------------------------------------
// loglik block sums (src/mcmc.jl:26-53): B[k][t] = Σ_{i in slot k} S[t][i], accumulated as (hi, lo) halves
// so that n² terms cannot overflow 64 bits.  One block per slot t; LDS bins per slot k.
// out[(t*hi + k)*4 + {0,1,2,3}] = D_hi, D_lo, L_hi, L_lo   (hi = slot high-water mark; slots >= hi are free)
// ---------------------------------------------------------------------------------------------------
#define RC_LO_BITS 24
#define RC_BS_TILE 4096   // slots k per block (LDS bins: 32 B each); blockIdx.y = tile of k — more than one only in wide contexts
__global__ __launch_bounds__(256) void k_blocksums(View V, int gen, int hi, long long *out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64 *bins = (u64 *)smem;  // [min(hi, RC_BS_TILE)][4]
    const int t = blockIdx.x, k0 = (int)blockIdx.y * RC_BS_TILE, kn = min(hi - k0, RC_BS_TILE);
    if (V.slot_size[t] == 0) return;
    for (int q = threadIdx.x; q < kn * 4; q += blockDim.x) bins[q] = 0;
    __syncthreads();
    const long long mask = ((long long)1 << RC_LO_BITS) - 1;
    for (int i = threadIdx.x; i < V.n; i += blockDim.x) {
        const int k = V.slot_of[i] - k0;
        if (k < 0 || k >= kn) continue;
        const long long d = V.SD[gen][(size_t)t * V.ld + i], l = V.SL[gen][(size_t)t * V.ld + i];
        atomicAdd(&bins[k * 4 + 0], (u64)(d >> RC_LO_BITS));
        atomicAdd(&bins[k * 4 + 1], (u64)(d & mask));
        atomicAdd(&bins[k * 4 + 2], (u64)(l >> RC_LO_BITS));
        atomicAdd(&bins[k * 4 + 3], (u64)(l & mask));
    }
    __syncthreads();
    for (int q = threadIdx.x; q < kn * 4; q += blockDim.x) out[((size_t)t * hi + k0) * 4 + q] = (long long)bins[q];
}

// Block sums of ONE would-be cluster against a bucketing of all points, straight from the matrices:
// out[b] = Σ_{y in rows} Σ_{x: bucket[x] = b} X[y][x] for X = D and logD, as (hi, lo) halves like k_blocksums.  The chain
// loop evaluates split proposals with it on a SNAPSHOT of the labels (bucket = snapshot slot, the members of the split
// cluster bucketed by their proposed side) without touching the live state.  One block per row; indices are internal.
__global__ __launch_bounds__(256) void k_split_eval(View V, const unsigned short *__restrict__ bucket, const int *__restrict__ rows,
                                                    int nbuckets, long long *out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64 *bins = (u64 *)smem;  // [nbuckets][4]
    for (int q = threadIdx.x; q < nbuckets * 4; q += blockDim.x) bins[q] = 0;
    __syncthreads();
    const int u = rows[blockIdx.x];
    const long long mask = ((long long)1 << RC_LO_BITS) - 1;
    for (int x = threadIdx.x; x < V.n; x += blockDim.x) {
        const size_t e = (size_t)u * V.ld + x;
        const long long d = (V.bits == 64) ? ((const long long *)V.Dq)[e] : (long long)((const int *)V.Dq)[e];
        const long long l = rc_load_L(V, u, x, d);
        const int b = bucket[x];
        atomicAdd(&bins[b * 4 + 0], (u64)(d >> RC_LO_BITS));
        atomicAdd(&bins[b * 4 + 1], (u64)(d & mask));
        atomicAdd(&bins[b * 4 + 2], (u64)(l >> RC_LO_BITS));
        atomicAdd(&bins[b * 4 + 3], (u64)(l & mask));
    }
    __syncthreads();
    for (int q = threadIdx.x; q < nbuckets * 4; q += blockDim.x)
        if (bins[q]) atomicAdd((u64 *)out + q, bins[q]);
}

// ---------------------------------------------------------------------------------------------------
// Co-clustering counts: counts[i][j] += (c_i == c_j) per recorded sample (adjacencymatrix, src/utils.jl:59-63;
// the sum of src/mcmc.jl:560).  uint32 — exact.  A read-modify-write of the n×n matrix per sample would cost
// as much HBM traffic as half a sweep, so recorded label vectors are queued as 16-bit slot
// ids (RC_CC_BATCH of them) and added in ONE pass over the matrix: tile = 16 rows × 1024 columns per block,
// 64 register counters per thread, row labels broadcast from LDS.
// ---------------------------------------------------------------------------------------------------
#define RC_CC_BATCH 32
#define RC_CC_ROWS 16
__global__ __launch_bounds__(256) void k_snapshot(const int *__restrict__ slot_of, const int *__restrict__ pi, int n, int ldn,
                                                 unsigned short *__restrict__ snap)
{
    const int i = blockIdx.x * 256 + threadIdx.x;  // original point order: the count matrix is the caller's
    if (i < ldn) snap[i] = (i < n) ? (unsigned short)slot_of[pi[i]] : (unsigned short)0xFFFE;
}

__global__ __launch_bounds__(256) void k_cocluster_batch(const unsigned short *__restrict__ snap, int cnt, int n, int ldn,
                                                        int ldc, unsigned *__restrict__ counts)
{
    __shared__ unsigned short ci[RC_CC_BATCH][RC_CC_ROWS];
    const int i0 = blockIdx.y * RC_CC_ROWS;
    const int j0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    for (int q = threadIdx.x; q < cnt * RC_CC_ROWS; q += 256) {
        const int t = q / RC_CC_ROWS, r = q % RC_CC_ROWS;
        ci[t][r] = (i0 + r < n) ? snap[(size_t)t * ldn + i0 + r] : (unsigned short)0xFFFF;
    }
    __syncthreads();
    if (j0 >= ldn) return;
    unsigned acc[RC_CC_ROWS][4];
#pragma unroll
    for (int r = 0; r < RC_CC_ROWS; ++r) { acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0; }
    for (int t = 0; t < cnt; ++t) {
        const ushort4 cj = *(const ushort4 *)(snap + (size_t)t * ldn + j0);
#pragma unroll
        for (int r = 0; r < RC_CC_ROWS; ++r) {
            const unsigned short c = ci[t][r];
            acc[r][0] += (c == cj.x); acc[r][1] += (c == cj.y); acc[r][2] += (c == cj.z); acc[r][3] += (c == cj.w);
        }
    }
#pragma unroll
    for (int r = 0; r < RC_CC_ROWS; ++r) {
        if (i0 + r < n) {
            uint4 *row = (uint4 *)(counts + (size_t)(i0 + r) * ldc + j0);
            uint4 v = *row;
            v.x += acc[r][0]; v.y += acc[r][1]; v.z += acc[r][2]; v.w += acc[r][3];
            *row = v;
        }
    }
}

__global__ void k_cocluster_final(const unsigned *__restrict__ counts, int n, int ldc, double numsamples,
                                  double *__restrict__ out)
{
    const size_t total = (size_t)n * n;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t i = t / n, j = t % n;
        out[t] = (double)counts[i * ldc + j] / numsamples;  // ./ numsamples, mcmc.jl:560
    }
}


__global__ void k_nop() {}

// ===================================================================================================
// Host side
// ===================================================================================================
struct rc_ctx {
    int dev = 0;
    int n = 0, ld = 0, kcap = 0;
    int eD = 0, eL = 0;
    hipStream_t sA = nullptr;  // resolve + observables (high priority)
    hipStream_t sB = nullptr;  // row-bucket reduction (k_bulk) of even sweeps
    hipStream_t sB2 = nullptr; // ... of odd sweeps: consecutive row reductions do not depend on each other (the S generation a
                               // reduction fills is cleared by k_resolve two sweeps earlier), so they may overlap and no launch gap
                               // separates them
    void *Dq = nullptr, *Lq = nullptr;  // int64 or int32 fixed point, INTERNAL point order (what the kernels read)
    void *Dq48 = nullptr;               // derived mode: Dq packed to 48 bits per entry (k_pack48), what k_bulk_syml2's fast path streams
    void *Dq_src = nullptr, *Lq_src = nullptr;  // the same matrices in the caller's point order (source of every re-layout)
    long long *diagq = nullptr, *diag_src = nullptr;
    int *pi = nullptr, *ipi = nullptr;  // device: original -> internal, internal -> original
    std::vector<int> h_pi, h_ipi;       // host copies
    bool relayout = true;               // RC_NO_RELAYOUT=1 keeps the caller's point order
    int sym_variant = -1;               // RC_SYM_VARIANT: 2 k_bulk_syml (wave-private LDS transposition), 1 k_bulk_symw (DPP only), 0 block-tiled
                                        // k_bulk_sym; -1 (default): k_bulk_syml when logD is derived, k_bulk_sym when it is stored (measured best)
    int sw_coarse = 0;                   // RC_SW_COARSE: rows per unit of k_bulk_syml (8..128, multiple of 4); 0 = chosen by syml_geometry
    size_t syml_pad = 0;                // RC_SYML_PAD: unused dynamic LDS per k_bulk_syml block (bytes), caps the blocks per CU
    int symw_per_cu = 3;                // RC_SYMW_PER_CU: blocks of k_bulk_syml / k_bulk_symw per CU (LDS: 40 KiB per block; the fourth slot is the resolver's)
    int4 *ufast = nullptr, *uslow = nullptr;   // unit lists of k_bulk_syml2 (build_syml2_lists)
    int *wfast = nullptr, *wslow = nullptr;
    int nfast = 0, nslow = 0, syml2_blocks = 0, syml2_g = 0;
    // the same lists for two blocks per CU: used while labels move (the 512-thread resolver of the previous sweep then fits beside the reduction)
    struct S2Alt { int4 *ufast = nullptr, *uslow = nullptr; int *wfast = nullptr, *wslow = nullptr; int nfast = 0, nslow = 0, blocks = 0; } s2alt;
    bool derived = false;               // logD derived from Dq on the fly (rc_qlog), not stored
    double2 *ltab = nullptr;            // device table of rc_qlog
    double2 *flt = nullptr;             // device table of rc_flog
    int n_relayouts = 0;                // re-layouts done so far (rc_set_state + automatic ones)
    int bits = 64;
    long long *SD[3] = {nullptr, nullptr, nullptr}, *SL[3] = {nullptr, nullptr, nullptr};
    int *slot_of = nullptr, *slot_size = nullptr, *slot_label = nullptr;
    short *slot_pos = nullptr, *slot_act = nullptr;
    int *perm[2] = {nullptr, nullptr}, *pslot[2] = {nullptr, nullptr};
    int *lsnap[2] = {nullptr, nullptr}, *work[2] = {nullptr, nullptr};  // label snapshots / work counters of k_bulk_sym
    u64 *cword[2] = {nullptr, nullptr};
    unsigned *rec = nullptr;
    int bulk_kernel = -1;      // RC_BULK_KERNEL: -1 auto, 0 k_bulk (full read, any layout), 1 k_bulk_sym (upper triangle)
    int last_bulk_kernel = 0;  // what the last enqueue chose
    int sym_item_tiles = 8;
    long long relayout_gap = 128;    // sweeps between two automatic re-layouts while K is between n/64 and n/36 (sweep_enqueue)
    int sym32_tr = 16;               // rows per tile of k_bulk_sym32 (16: 34 KiB blocks, 32: 67 KiB blocks)
    int sym32_bpc = 3;               // its blocks per CU
    int res_threads = 0;       // k_resolve block size: 0 = adaptive, else forced by RC_RES_THREADS (256 or 512)
    double *A = nullptr;
    u64 *keys[2] = {nullptr, nullptr};
    unsigned *arrive[2] = {nullptr, nullptr};
    DevScalars *sc = nullptr;
    HostSummary *hsum = nullptr;      // pinned, host-mapped
    HostSummary *hsum_dev = nullptr;  // its device address
    long long *blocks = nullptr;  // k_blocksums output [kcap][kcap][4]
    unsigned *counts = nullptr;   // co-clustering counts [n][ldc]
    int ldc = 0;
    unsigned short *snap = nullptr;  // queued label snapshots [RC_CC_BATCH][ldc]
    int snap_cnt = 0;
    double *cc_out = nullptr;
    rc_params P{};
    bool have_params = false, have_state = false;
    int G = 256;
    int rows_per_split = 256;
    int num_cus = 256;
    // software pipeline
    bool registered = false;          // counted in g_res_contexts
    unsigned *used_scratch = nullptr; // label bitsets of the resolver blocks for large n
    double *wc = nullptr;             // score cache of the resolver, kcap x ldw (RC_SCORE_CACHE=0: none, =1: filled in every sweep)
    int wc_always = 0;
    int maxb = RC_MAXB;               // resolver batch capacity (finish_create: the largest that lets the resolver's LDS fit beside the row reduction)
    bool res_one_stream = false;      // small problems: every resolver on stream B (in order, no event between consecutive resolvers), every row reduction on B2
    hipStream_t s_res_last = nullptr; // stream of the last resolver launch (sB / sB2 by sweep parity, sA in incremental mode)
    hipEvent_t ev_a = nullptr;        // marker on stream A: work that reads the state and must precede the next resolver
    hipStream_t sC = nullptr;         // copy stream of the speculative loop's snapshots (device -> pinned host), off the sweeps' critical path
    hipEvent_t ev_k = nullptr;        // the snapshot's kernels are done (stream A) — what stream C waits for
    hipEvent_t ev_blocks_busy = nullptr;   // the last asynchronous copy out of c->blocks on stream C (nullptr: none): the next k_blocksums waits for it
    bool sA_dirty = false;
    long long t_next = 0;     // internal index of the next sweep (0 after rc_set_state)
    long long bulk_enq = -1;  // highest sweep index whose k_bulk has been enqueued
    bool prefetch = true;     // enqueue k_bulk(t+1) together with k_resolve(t)
    bool incremental = false; // RC_MODE_INCREMENTAL: S is only maintained by exact corrections, never recomputed
    bool want_incremental = false;   // the mode the caller asked for (rc_set_mode): a wide context runs incrementally whatever was asked, and goes back to this when it narrows
    int inc_gen = 0;          // the S generation that incremental mode keeps current
    hipEvent_t ev_bulk[4] = {nullptr, nullptr, nullptr, nullptr}, ev_res[4] = {nullptr, nullptr, nullptr, nullptr};
    // timing of k_bulk
    bool timing = false;
    int timing_every = 1;     // time every N-th k_bulk launch (HIP timing events cost a few us each)
    size_t bulk_lds = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_free;
    double bulk_ms = 0.0;
    long long bulk_launches = 0;
    double ev_overhead_ms = 0.0;  // subtracted from every timed launch: 0 since the events ride in the dispatch (rc_event_overhead reports it)
    DevScalars last{};
    int dbg = 0;
    long long chain_rollbacks = 0, chain_split_evals = 0, chain_workers = 0, chain_grows = 0;   // rc_chain_stats: the last rc_run_chain
    // slot capacity (number of clusters the tables hold).  It grows on demand — rc_set_state with more clusters, a sweep or a
    // split–merge proposal that needs one more slot — up to kcap_max = min(n, RC_MAX_KCAP); the reference's clustsizes has
    // length n (types.jl:131-137, mcmc.jl:198-199).  kcap = 0 at rc_create: sized from the first state (kcap_auto).
    bool wide = false;            // kcap > RC_MAX_KCAP: slot tables in global memory, one S generation, k_sweep_wide (see there)
    unsigned *wide_scratch = nullptr;
    size_t blocks_cap = 0;        // slots c->blocks is sized for (k_blocksums output, blocks_cap² x 32 B; wide contexts size it on demand)
    bool kcap_auto = false;
    bool kcap_fixed = false;      // RC_KCAP_FIXED=1: never grow (the old behaviour: RC_ERR_CAPACITY), for tests of the error path
    // run-time options: defaults from the environment when the context is created, changed afterwards with rc_set_option — nothing
    // reads the environment per sweep or per chain
    int opt_cu_cache = 1;         // "lds_point_cache": 0 = k_resolve reads pi[] / slot_of[] from global memory in every pass (the path blocks with more than RC_CPB_LDS chunks take; tests)
    int opt_prune = -1;           // "prune": -1 automatic (on behind a sweep that changed at most 2 labels), 0 never, 1 always   (RC_NO_PRUNE / RC_PRUNE_ALWAYS)
    int opt_chain_workers = 0;    // "chain_workers": worker threads of rc_run_chain, 0 = automatic   (RC_CHAIN_WORKERS)
    int opt_chain_depth = 0;      // "chain_depth": iterations in flight, 0 = automatic   (RC_CHAIN_DEPTH)
    int opt_chain_pipeline = 1;   // "chain_pipeline": 0 = the synchronous form of the loop   (RC_CHAIN_PIPELINE)
    bool bulk_rows_forced = false;// (diagnostic builds: RC_BULK_ROWS pins the split length of k_bulk)
    bool sm_profile = false;      // (diagnostic builds: RC_SM_PROFILE)
    bool broken = false;          // a capacity growth ran out of device memory half-way: every later call returns RC_ERR_STATE
    int kcap_max = 0;
    int n_grows = 0;              // capacity growths so far (rc_capacity_info)
    struct SweepRec { double r, p; uint64_t seed, sweep_index; long long t; };
    std::deque<SweepRec> inflight;   // sweeps enqueued since the last successful synchronisation: replayed after a capacity growth
    bool recovering = false;
    // split–merge support (host-side proposal logic on borrowed host matrices)
    const double *hostD = nullptr, *hostL = nullptr;
    std::vector<double> ownL;            // host logD computed by the library when the caller passes none
    std::vector<int64_t> checkpoint;     // labels saved by rc_state_checkpoint
    int *d_moves = nullptr;
    size_t d_moves_cap = 0;
    long long state_version = 0;         // bumped whenever labels may have changed (sweeps, moves, rc_set_state)
    long long ll_version = -1;           // state_version the cached log-likelihood belongs to
    double ll_cached = 0.0;
    std::vector<long long> B_cur;        // block sums, slot sizes and labels of the state ll_cached belongs to
    std::vector<int> B_ssize, B_slabel;
    int B_hi = 0;
    long long B_version = -2;
    // lgammal(alpha + delta1*pairs) - lgammal(alpha) and the zeta analogue, memoised by the integer pair count: between
    // consecutive rc_loglik calls cluster sizes rarely change, so nearly all of the K + K(K-1)/2 evaluations hit
    struct LLTerm { long long e[4] = {0, 0, 0, 0}; int sk = -1, st = -1; long double term = 0; };
    struct LLCache {                     // memo of loglik_host: one per thread that evaluates log-likelihoods
        std::unordered_map<long long, long double> lg_memo1, lg_memo2;
        std::vector<LLTerm> ll_cache;    // loglik terms per slot pair ([t][k], ll_dim × ll_dim), see loglik_host
        int ll_dim = 0;
    };
    LLCache llc;
    long long params_version = 0;        // bumped by rc_set_params (worker-thread caches of the chain loop follow it)
    long long *pinB[RC_REC_SLOTS] = {};  // pinned staging: block sums / label snapshot / completion event per sample slot
    unsigned short *pinLab[RC_REC_SLOTS] = {};
    hipEvent_t pinEv[RC_REC_SLOTS] = {};
    int pin_hi = 0;
    char err[512] = {0};
};

static thread_local char g_err[512] = {0};

static int32_t fail(rc_ctx *c, int32_t code, const char *fmt, ...)
{
    char *dst = c ? c->err : g_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    if (c) snprintf(g_err, sizeof(g_err), "%s", c->err);
    return code;
}

#define HIPCHK(c, call)                                                                               \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail(c, (e_ == hipErrorOutOfMemory) ? RC_ERR_OOM : RC_ERR_HIP, "%s failed: %s (%s:%d)", #call, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                   \
    } while (0)

static View make_view(const rc_ctx *c)
{
    View V{};
    V.n = c->n; V.ld = c->ld; V.kcap = c->kcap; V.maxb = c->maxb; V.used_scratch = c->used_scratch; V.wide_scratch = c->wide_scratch;
    V.cu_cache = c->opt_cu_cache; V.wc = c->wc; V.wc_always = c->wc_always; V.ldw = (c->n + RC_PTS - 1) / RC_PTS * RC_PTS;
    V.Dq = c->Dq; V.Lq = c->Lq; V.Dq48 = c->Dq48; V.bits = c->bits; V.diagq = c->diagq; V.pi = c->pi;
    V.derived = c->derived ? 1 : 0; V.qsD = std::ldexp(1.0, -c->eD); V.qsL = std::ldexp(1.0, c->eL); V.ltab = c->ltab; V.flt = c->flt; V.qeD = c->eD;
    for (int g = 0; g < 3; ++g) { V.SD[g] = c->SD[g]; V.SL[g] = c->SL[g]; }
    for (int g = 0; g < 2; ++g) { V.perm[g] = c->perm[g]; V.pslot[g] = c->pslot[g]; V.keys[g] = c->keys[g]; V.arrive[g] = c->arrive[g]; V.snap[g] = c->lsnap[g]; V.work[g] = c->work[g]; V.cword[g] = c->cword[g]; }
    V.rec = c->rec;
    V.slot_of = c->slot_of; V.slot_size = c->slot_size; V.slot_label = c->slot_label;
    V.slot_pos = c->slot_pos; V.slot_act = c->slot_act;
    V.A = c->A; V.sc = c->sc; V.hsum = c->hsum_dev;
    V.ufast = c->ufast; V.uslow = c->uslow; V.nfast = c->nfast; V.nslow = c->nslow; V.wfast = c->wfast; V.wslow = c->wslow;
    V.scD = std::ldexp(1.0, -c->eD); V.scL = std::ldexp(1.0, -c->eL);
    V.alpha = c->P.alpha; V.beta = c->P.beta; V.zeta = c->P.zeta; V.gamma = c->P.gamma;
    V.delta1 = c->P.delta1; V.delta2 = c->P.delta2;
    V.repulsion = c->P.repulsion ? 1 : 0;
    V.cL = (c->P.delta1 - 1.0) - (c->P.repulsion ? (c->P.delta2 - 1.0) : 0.0);
    V.maxK = c->P.maxK;
    return V;
}

static int ceil_log2_ll(long long n)
{
    int b = 0;
    while (((long long)1 << b) < n) ++b;
    return b;
}

// 64-bit storage: any sum of n entries fits int64.  32-bit storage: every entry fits int32 (sums are 64-bit).
static int quant_exponent(long long n, double maxabs, int bits)
{
    if (maxabs == 0.0) return 0;
    int ex;
    std::frexp(maxabs, &ex);
    return bits == 64 ? 62 - ex - ceil_log2_ll(n) : 30 - ex;
}

extern "C" const char *rc_last_error(const rc_ctx *ctx) { return ctx ? ctx->err : g_err; }

static void free_all(rc_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->dev);
    void *ptrs[] = {c->Dq48, c->ltab, c->flt, c->Dq, c->Lq, c->Dq_src, c->Lq_src, c->diag_src, c->pi, c->ipi, c->diagq, c->SD[0], c->SD[1], c->SD[2], c->SL[0], c->SL[1], c->SL[2], c->slot_of,
                    c->slot_size, c->slot_label, c->slot_pos, c->slot_act, c->perm[0], c->perm[1], c->pslot[0],
                    c->pslot[1], c->lsnap[0], c->lsnap[1], c->work[0], c->work[1], c->cword[0], c->cword[1], c->rec, c->A, c->keys[0], c->keys[1], c->arrive[0], c->arrive[1], c->sc, c->blocks,
                    c->counts, c->cc_out, c->snap, c->d_moves, c->used_scratch, c->wide_scratch, c->wc, c->ufast, c->uslow, c->wfast, c->wslow, c->s2alt.ufast, c->s2alt.uslow, c->s2alt.wfast, c->s2alt.wslow};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (c->hsum) (void)hipHostFree(c->hsum);
    for (int q = 0; q < RC_REC_SLOTS; ++q) {
        if (c->pinB[q]) (void)hipHostFree(c->pinB[q]);
        if (c->pinLab[q]) (void)hipHostFree(c->pinLab[q]);
        if (c->pinEv[q]) (void)hipEventDestroy(c->pinEv[q]);
    }
    for (auto &e : c->ev_pending) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto &e : c->ev_free) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (c->ev_a) (void)hipEventDestroy(c->ev_a);
    if (c->ev_k) (void)hipEventDestroy(c->ev_k);
    if (c->sC) (void)hipStreamDestroy(c->sC);
    for (int q = 0; q < 4; ++q) {
        if (c->ev_bulk[q]) (void)hipEventDestroy(c->ev_bulk[q]);
        if (c->ev_res[q]) (void)hipEventDestroy(c->ev_res[q]);
    }
    if (c->sB && c->sB != c->sA) (void)hipStreamDestroy(c->sB);
    if (c->sB2 && c->sB2 != c->sA && c->sB2 != c->sB) (void)hipStreamDestroy(c->sB2);
    if (c->sA) (void)hipStreamDestroy(c->sA);
    delete c;
}

// RC_SM_PROFILE=1: wall time per phase of rc_splitmerge, printed by rc_destroy
struct SmProfile {
    double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long calls = 0;
    bool on = rc_env_diag("RC_SM_PROFILE") && atoi(rc_env_diag("RC_SM_PROFILE"));
    std::chrono::steady_clock::time_point last;
    void start() { if (on) { last = std::chrono::steady_clock::now(); ++calls; } }
    void lap(int k) { if (on) { auto now = std::chrono::steady_clock::now(); t[k] += std::chrono::duration<double>(now - last).count(); last = now; } }
    void report() const
    {
        if (!on || !calls) return;
        static const char *names[8] = {"pull_labels", "launch_state", "scans", "loglik_current", "apply", "loglik_proposed", "revert", "-"};
        fprintf(stderr, "[rc_splitmerge profile] %lld proposals:", calls);
        for (int k = 0; k < 7; ++k) fprintf(stderr, " %s %.1f us", names[k], t[k] / (double)calls * 1e6);
        fprintf(stderr, "\n");
    }
};
static SmProfile g_smprof;
static void smprof_report() { g_smprof.report(); }

// Resolver launches of different contexts on one device must not overlap (launch_resolve).  A device that only ever holds
// one context at a time — the normal case — skips the chain and its two barrier packets per sweep; the first time a second
// context appears the device is drained once and the chain is used until the device is back to one context.  All of this is
// PER DEVICE (rc_run_chains drives eight devices from eight host threads of one process: their launches share nothing).
struct ResDevice {
    std::mutex m;             // held across the resolver launch + event records of this device only
    hipEvent_t ev = nullptr;  // completion of the last chained resolver on this device
    int contexts = 0;         // live contexts
    bool multi = false;       // more than one context has been live at a time: resolvers are chained through `ev`
};
static ResDevice g_res[64];

static void res_register(rc_ctx *c)
{
    bool drain = false;
    {
        ResDevice &rd = g_res[c->dev & 63];
        std::lock_guard<std::mutex> lock(rd.m);
        if (++rd.contexts > 1 && !rd.multi) { rd.multi = true; drain = true; }
    }
    if (drain) (void)hipDeviceSynchronize();   // launches made without the chain are complete before the newcomer's first one
}

// called by rc_destroy AFTER the context's streams have been drained: once a single context is left on the device nothing of
// the leaving one is in flight, so the survivor's resolvers (ordered among themselves by ev_res) need no chain any more
static void res_unregister(rc_ctx *c)
{
    ResDevice &rd = g_res[c->dev & 63];
    std::lock_guard<std::mutex> lock(rd.m);
    if (rd.contexts > 0) --rd.contexts;
    if (rd.contexts <= 1) rd.multi = false;
}

extern "C" int32_t rc_destroy(rc_ctx *ctx)
{
    smprof_report();
    if (!ctx) return RC_OK;
    (void)hipSetDevice(ctx->dev);
    if (ctx->sA) (void)hipStreamSynchronize(ctx->sA);
    if (ctx->sB) (void)hipStreamSynchronize(ctx->sB);
    if (ctx->sB2) (void)hipStreamSynchronize(ctx->sB2);
    if (ctx->registered) { res_unregister(ctx); ctx->registered = false; }
    free_all(ctx);
    return RC_OK;
}

// The buffers sized by the slot capacity c->kcap: the three generations of the row-sum table, the slot tables, the score
// cache, the block sums of loglik.  Frees what is there first (capacity growth).  Contents: zeroed where an invariant needs it
// (rows of free slots are zero; free slots have size 0 / label 0) — rc_set_state fills the rest.
static int32_t alloc_slot_buffers(rc_ctx *c)
{
    void **ptrs[] = {(void **)&c->SD[0], (void **)&c->SD[1], (void **)&c->SD[2], (void **)&c->SL[0], (void **)&c->SL[1], (void **)&c->SL[2],
                     (void **)&c->slot_size, (void **)&c->slot_label, (void **)&c->slot_pos, (void **)&c->slot_act, (void **)&c->wc, (void **)&c->blocks};
    if (c->sC) { HIPCHK(c, hipStreamSynchronize(c->sC)); c->ev_blocks_busy = nullptr; }   // (a snapshot copy out of c->blocks may be in flight)
    for (void **pp : ptrs)
        if (*pp) { (void)hipFree(*pp); *pp = nullptr; }
    const size_t ld = (size_t)c->ld, k = (size_t)c->kcap;
    c->wide = c->kcap > RC_MAX_KCAP;
    if (c->wide_scratch) { (void)hipFree(c->wide_scratch); c->wide_scratch = nullptr; }
    c->blocks_cap = 0;
    // a wide context keeps ONE generation of the row-sum table, corrected in place (as the incremental mode does), no score cache,
    // and sizes the block-sum buffer of the log-likelihood by the clusters in use when it is asked for (kcap² x 32 B is 34 GB at 32767)
    for (int g = 0; g < (c->wide ? 1 : 3); ++g) {
        HIPCHK(c, hipMalloc(&c->SD[g], k * ld * sizeof(long long)));
        HIPCHK(c, hipMalloc(&c->SL[g], k * ld * sizeof(long long)));
    }
    HIPCHK(c, hipMalloc(&c->slot_size, k * sizeof(int)));
    HIPCHK(c, hipMalloc(&c->slot_label, k * sizeof(int)));
    HIPCHK(c, hipMalloc(&c->slot_pos, k * sizeof(short)));
    HIPCHK(c, hipMalloc(&c->slot_act, k * sizeof(short)));
    if (c->wide) {
        HIPCHK(c, hipMalloc((void **)&c->wide_scratch, ((size_t)(c->n + 31) / 32 + 2 * (k + 1)) * sizeof(unsigned)));
    } else {
        if (!rc_env("RC_SCORE_CACHE") || atoi(rc_env("RC_SCORE_CACHE")) != 0)
            HIPCHK(c, hipMalloc((void **)&c->wc, k * (size_t)((c->n + RC_PTS - 1) / RC_PTS * RC_PTS) * sizeof(double)));
        HIPCHK(c, hipMalloc(&c->blocks, k * k * 4 * sizeof(long long)));
        c->blocks_cap = k;
    }
    HIPCHK(c, hipMemsetAsync(c->slot_size, 0, k * sizeof(int), c->sA));
    HIPCHK(c, hipMemsetAsync(c->slot_label, 0, k * sizeof(int), c->sA));
    HIPCHK(c, hipStreamSynchronize(c->sA));
    return RC_OK;
}

static int32_t create_impl(rc_ctx *c, int64_t n, const double *D, const double *logD, const double *points = nullptr,
                           int64_t dim = 0)
{
    HIPCHK(c, hipSetDevice(c->dev));
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, c->dev));
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    int pr_least = 0, pr_greatest = 0;
    HIPCHK(c, hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest));
    HIPCHK(c, hipStreamCreateWithPriority(&c->sA, hipStreamNonBlocking, pr_greatest));
    if (rc_env_diag("RC_ONE_STREAM") && atoi(rc_env_diag("RC_ONE_STREAM"))) {
        c->sB = c->sA;  // experiment: everything in order on one stream, no cross-stream events
        c->sB2 = c->sA;
        c->prefetch = false;
    } else {
        HIPCHK(c, hipStreamCreateWithPriority(&c->sB, hipStreamNonBlocking, pr_least));
        if (rc_env_diag("RC_ONE_BULK_STREAM") && atoi(rc_env_diag("RC_ONE_BULK_STREAM"))) c->sB2 = c->sB;
        else HIPCHK(c, hipStreamCreateWithPriority(&c->sB2, hipStreamNonBlocking, pr_least));
    }
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_k, hipEventDisableTiming));
    HIPCHK(c, hipStreamCreateWithFlags(&c->sC, hipStreamNonBlocking));
    for (int q = 0; q < 4; ++q) {
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_bulk[q], hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_res[q], hipEventDisableTiming));
    }
    const size_t nn = (size_t)n * n;
    const size_t ld = (size_t)c->ld;
    double *tmpD = nullptr, *tmpL = nullptr;
    unsigned *flags = nullptr;
    u64 *mx = nullptr;
    auto cleanup = [&]() {
        if (tmpD) (void)hipFree(tmpD);
        if (tmpL) (void)hipFree(tmpL);
        if (flags) (void)hipFree(flags);
        if (mx) (void)hipFree(mx);
    };
#define HIPCHK2(call)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            cleanup();                                                                                    \
            return fail(c, (e_ == hipErrorOutOfMemory) ? RC_ERR_OOM : RC_ERR_HIP, "%s failed: %s (%s:%d)", #call, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                       \
        }                                                                                                 \
    } while (0)
    hipStream_t s = c->sA;
    HIPCHK2(hipMalloc(&tmpD, nn * sizeof(double)));
    HIPCHK2(hipMalloc(&tmpL, std::max(nn, (size_t)n * (size_t)std::max<int64_t>(dim, 1)) * sizeof(double)));
    HIPCHK2(hipMalloc(&flags, 2 * sizeof(unsigned)));
    HIPCHK2(hipMalloc(&mx, 2 * sizeof(u64)));
    const size_t esz = (size_t)c->bits / 8;
    // logD not given: derive it from Dq on the fly instead of storing it (rc_qlog; halves the bytes of every sweep)
    bool derived = !logD && c->bits == 64 && !(rc_env("RC_STORED_LOG") && atoi(rc_env("RC_STORED_LOG")));
    HIPCHK2(hipMalloc(&c->Dq, (size_t)n * ld * esz));
    HIPCHK2(hipMalloc(&c->Dq_src, (size_t)n * ld * esz));
    HIPCHK2(hipMalloc(&c->diagq, (size_t)n * sizeof(long long)));
    HIPCHK2(hipMalloc(&c->diag_src, (size_t)n * sizeof(long long)));
    HIPCHK2(hipMalloc(&c->pi, (size_t)n * sizeof(int)));
    HIPCHK2(hipMalloc(&c->ipi, (size_t)n * sizeof(int)));
    HIPCHK2(hipMalloc(&c->ltab, 128 * sizeof(double2)));
    {   // table of rc_flog (see there): interval i of z's offset from 0.6875 in units of 2^-7 of the mantissa; long double on the host
        double2 ft[128];
        for (int i = 0; i < 128; ++i) {
            const uint64_t b0 = ((uint64_t)(0x3fe60000u + ((uint32_t)i << 13))) << 32, b1 = ((uint64_t)(0x3fe60000u + ((uint32_t)(i + 1) << 13))) << 32;
            double z0, z1;
            memcpy(&z0, &b0, 8); memcpy(&z1, &b1, 8);
            if (i == 79 || i == 80) { ft[i].x = 1.0; ft[i].y = 0.0; continue; }   // the intervals that touch 1.0: r = z - 1, exactly
            const long double cc = 0.5L * ((long double)z0 + (long double)z1);
            const double invc = (double)(1.0L / cc);
            ft[i].x = invc; ft[i].y = (double)(-logl((long double)invc));
        }
        HIPCHK2(hipMalloc(&c->flt, sizeof(ft)));
        HIPCHK2(hipMemcpy(c->flt, ft, sizeof(ft), hipMemcpyHostToDevice));
    }
    for (int g = 0; g < 2; ++g) {
        HIPCHK2(hipMalloc(&c->perm[g], (size_t)n * sizeof(int)));
        HIPCHK2(hipMalloc(&c->pslot[g], (size_t)n * sizeof(int)));
        HIPCHK2(hipMalloc(&c->lsnap[g], (size_t)n * sizeof(int)));
        HIPCHK2(hipMalloc(&c->work[g], RC_WORK_BYTES));
        HIPCHK2(hipMemsetAsync(c->work[g], 0, RC_WORK_BYTES, s));
        HIPCHK2(hipMalloc(&c->cword[g], 2 * ((size_t)(n + RC_PTS - 1) / RC_PTS + 1) * sizeof(u64)));
        HIPCHK2(hipMemsetAsync(c->cword[g], 0, 2 * ((size_t)(n + RC_PTS - 1) / RC_PTS + 1) * sizeof(u64), s));
        HIPCHK2(hipMalloc(&c->keys[g], (size_t)(2 * n + 8) * sizeof(u64)));
        HIPCHK2(hipMalloc(&c->arrive[g], RC_BAR_WORDS * sizeof(unsigned)));
    }
    HIPCHK2(hipMalloc(&c->slot_of, (size_t)n * sizeof(int)));
    HIPCHK2(hipMalloc(&c->rec, 2 * (size_t)n * sizeof(unsigned)));
    c->wc_always = rc_env("RC_SCORE_CACHE") && atoi(rc_env("RC_SCORE_CACHE")) == 1;
    {
        const int32_t rcs = alloc_slot_buffers(c);   // everything sized by the slot capacity (re-allocated when it grows)
        if (rcs != RC_OK) { cleanup(); return rcs; }
    }
    if (c->n > RC_USED_LDS_MAX_N) HIPCHK2(hipMalloc((void **)&c->used_scratch, (size_t)std::max(c->num_cus, 256) * (size_t)((c->n + 31) / 32) * sizeof(unsigned)));
    HIPCHK2(hipMalloc(&c->A, (size_t)(n + 1) * sizeof(double)));
    HIPCHK2(hipMalloc(&c->sc, sizeof(DevScalars)));
    HIPCHK2(hipHostMalloc((void **)&c->hsum, hsum_bytes(c->kcap_max), hipHostMallocMapped));
    std::memset(c->hsum, 0, hsum_bytes(c->kcap_max));
    HIPCHK2(hipHostGetDevicePointer((void **)&c->hsum_dev, c->hsum, 0));
    HIPCHK2(hipMemsetAsync(c->Dq, 0, (size_t)n * ld * esz, s));
    HIPCHK2(hipMemsetAsync(c->Dq_src, 0, (size_t)n * ld * esz, s));
    HIPCHK2(hipMemsetAsync(c->sc, 0, sizeof(DevScalars), s));
    HIPCHK2(hipMemsetAsync(flags, 0, 2 * sizeof(unsigned), s));
    HIPCHK2(hipMemsetAsync(mx, 0, 2 * sizeof(u64), s));
    const int gb = std::min<size_t>((nn + 255) / 256, 4096);
    if (points) {
        // MCMCData(points): distances are computed on the device, no n×n host matrix is involved
        HIPCHK2(hipMemcpyAsync(tmpL, points, (size_t)n * (size_t)dim * sizeof(double), hipMemcpyHostToDevice, s));  // tmpL as staging
        const unsigned nt = (unsigned)((n + 63) / 64);
        k_pairwise<<<dim3(nt, nt), 256, 0, s>>>(tmpL, (int)n, (int)dim, tmpD);
    } else {
        HIPCHK2(hipMemcpyAsync(tmpD, D, nn * sizeof(double), hipMemcpyHostToDevice, s));
    }
    k_check<<<gb, 256, 0, s>>>(tmpD, (int)n, flags, mx);
    unsigned hflags[2] = {0, 0};
    u64 hmx[2] = {0, 0};
    auto stage_log = [&]() -> hipError_t {   // tmpL = logD (given or computed), its max and finiteness flag
        if (logD) {
            hipError_t e = hipMemcpyAsync(tmpL, logD, nn * sizeof(double), hipMemcpyHostToDevice, s);
            if (e != hipSuccess) return e;
        } else {
            k_make_log<<<gb, 256, 0, s>>>(tmpD, tmpL, (int)n);
        }
        k_maxabs<<<gb, 256, 0, s>>>(tmpL, nn, flags + 1, mx + 1, logD ? (int)n : 0);   // a caller's logD must be symmetric too
        return hipSuccess;
    };
    const bool log_staged = !derived;
    if (!derived) HIPCHK2(stage_log());
    HIPCHK2(hipMemcpyAsync(hflags, flags, sizeof(hflags), hipMemcpyDeviceToHost, s));
    HIPCHK2(hipMemcpyAsync(hmx, mx, sizeof(hmx), hipMemcpyDeviceToHost, s));
    HIPCHK2(hipStreamSynchronize(s));
    HIPCHK2(hipGetLastError());
    if (hflags[0] & 1u) { cleanup(); return fail(c, RC_ERR_DOMAIN, "D must be symmetric."); }
    if (hflags[0] & 2u) { cleanup(); return fail(c, RC_ERR_DOMAIN, "D must be finite."); }
    if (!logD && (hflags[0] & 4u)) {
        cleanup();
        return fail(c, RC_ERR_DOMAIN, "off-diagonal entries of D must be positive: log D = -Inf / NaN otherwise (duplicate observations? remove them or add a small jitter).");
    }
    double maxD, maxL = 0;
    std::memcpy(&maxD, &hmx[0], 8);
    c->eD = quant_exponent(n, maxD, c->bits);
    if (derived && maxD > 0.0) {
        // every Dq < 2^47: the streaming row reduction reads a 48-bit packed copy of D (k_pack48: 6 instead of 8 bytes per entry;
        // quantum 2^-47 of the largest entry — binding for n < 32768), and rc_qlog converts the entries to double through the
        // mantissa of 2^52
        int ex;
        std::frexp(maxD, &ex);
        c->eD = std::min(c->eD, (rc_env_diag("RC_NO_PACK48") && atoi(rc_env_diag("RC_NO_PACK48"))) ? 51 - ex : 47 - ex);
    }
    if (c->bits == 64) k_quantize<long long><<<gb, 256, 0, s>>>(tmpD, (int)n, c->ld, c->eD, (long long *)c->Dq_src, c->diag_src);
    else k_quantize<int><<<gb, 256, 0, s>>>(tmpD, (int)n, c->ld, c->eD, (int *)c->Dq_src, c->diag_src);
    if (derived) {
        long long *mn = nullptr;
        HIPCHK2(hipMalloc(&mn, sizeof(long long)));
        const long long big = 0x7fffffffffffffffll;
        HIPCHK2(hipMemcpyAsync(mn, &big, sizeof(big), hipMemcpyHostToDevice, s));
        k_derived_scan<<<gb, 256, 0, s>>>((const long long *)c->Dq_src, (int)n, c->ld, c->eD, mn, mx + 1);
        long long hmn = 0;
        hipError_t e1 = hipMemcpyAsync(&hmn, mn, sizeof(hmn), hipMemcpyDeviceToHost, s);
        hipError_t e2 = hipMemcpyAsync(&hmx[1], mx + 1, sizeof(u64), hipMemcpyDeviceToHost, s);
        hipError_t e3 = hipStreamSynchronize(s);
        (void)hipFree(mn);
        HIPCHK2(e1); HIPCHK2(e2); HIPCHK2(e3);
        // log(Dq·2^-eD) differs from log(D) by the relative rounding of the entry, 1/(2·Dq): derive only when every
        // off-diagonal entry is at least 2^32 quanta (error ≤ 1.2e-10 per entry, ≤ 1e-12 for entries within 2^-10 of the
        // largest); a matrix with (near-)zero distances keeps logD of the exact doubles
        if (n > 1 && hmn < (1ll << 32)) {
            // stored logD after all: the 47-bit cap on eD served only the packed copy and rc_qlog — quantise D again with every
            // fraction bit the 64-bit sums allow
            derived = false;
            const int e_full = quant_exponent(n, maxD, c->bits);
            if (e_full != c->eD) {
                c->eD = e_full;
                k_quantize<long long><<<gb, 256, 0, s>>>(tmpD, (int)n, c->ld, c->eD, (long long *)c->Dq_src, c->diag_src);
            }
        }
    }
    if (!derived && !log_staged) {
        // (stored mode reached through the fallback above: stage logD now)
        HIPCHK2(hipMemsetAsync(mx + 1, 0, sizeof(u64), s));
        HIPCHK2(stage_log());
        HIPCHK2(hipMemcpyAsync(hflags, flags, sizeof(hflags), hipMemcpyDeviceToHost, s));
        HIPCHK2(hipMemcpyAsync(hmx, mx, sizeof(hmx), hipMemcpyDeviceToHost, s));
        HIPCHK2(hipStreamSynchronize(s));
    }
    if (hflags[1] & 2u) { cleanup(); return fail(c, RC_ERR_DOMAIN, "logD must be finite."); }
    // the symmetric row-reduction kernels read the upper triangle only, the full-read kernel both: an asymmetric logD would
    // make the two disagree (the reference derives logD from the symmetric D, types.jl:155, so it is symmetric there)
    if (hflags[1] & 1u) { cleanup(); return fail(c, RC_ERR_DOMAIN, "logD must be symmetric."); }
    std::memcpy(&maxL, &hmx[1], 8);
    c->eL = quant_exponent(n, maxL, c->bits);
    if (derived && maxL > 0.0) {
        // rc_qlog rounds with the magic-number trick, which needs |logD·2^eL| < 2^51 (binding only for n < 2048)
        int ex;
        std::frexp(maxL, &ex);
        c->eL = std::min(c->eL, 50 - ex);
    }
    c->derived = derived;
    {
        // table of rc_qlog: (1/c_j, rint(log c_j · 2^eL) + 1.5·2^52), c_j = 1 + (j+½)/128 — in long double, once eL is known
        double tab[256];
        for (int j = 0; j < 128; ++j) {
            const long double cj = 1.0L + ((long double)j + 0.5L) / 128.0L;
            tab[2 * j] = 2.0 * (double)(1.0L / cj);                                  // (2/c_j: rc_qlog_prep hands over m/2)
            tab[2 * j + 1] = (double)(rintl(logl(cj) * ldexpl(1.0L, c->eL)) + 0x1.8p52L);
        }
        HIPCHK2(hipMemcpy(c->ltab, tab, sizeof(tab), hipMemcpyHostToDevice));
    }
    if (derived && !(rc_env_diag("RC_NO_LQ_COPY") && atoi(rc_env_diag("RC_NO_LQ_COPY")))) {
        // random-access copy of the derived values for the resolver (see k_derived_fill); the row reduction ignores it
        HIPCHK2(hipMalloc(&c->Lq, (size_t)n * ld * esz));
        HIPCHK2(hipMalloc(&c->Lq_src, (size_t)n * ld * esz));
        HIPCHK2(hipMemsetAsync(c->Lq_src, 0, (size_t)n * ld * esz, s));
        k_derived_fill<<<gb, 256, 0, s>>>((const long long *)c->Dq_src, (int)n, c->ld, c->eD, std::ldexp(1.0, c->eL), c->ltab, (long long *)c->Lq_src);
    }
    if (!derived) {
        HIPCHK2(hipMalloc(&c->Lq, (size_t)n * ld * esz));
        HIPCHK2(hipMalloc(&c->Lq_src, (size_t)n * ld * esz));
        HIPCHK2(hipMemsetAsync(c->Lq, 0, (size_t)n * ld * esz, s));
        HIPCHK2(hipMemsetAsync(c->Lq_src, 0, (size_t)n * ld * esz, s));
        if (c->bits == 64) k_quantize<long long><<<gb, 256, 0, s>>>(tmpL, (int)n, c->ld, c->eL, (long long *)c->Lq_src, nullptr);
        else k_quantize<int><<<gb, 256, 0, s>>>(tmpL, (int)n, c->ld, c->eL, (int *)c->Lq_src, nullptr);
    }
    // until rc_set_state chooses a cluster-contiguous layout the internal order is the caller's
    c->h_pi.resize((size_t)n); c->h_ipi.resize((size_t)n);
    for (int64_t q = 0; q < n; ++q) { c->h_pi[(size_t)q] = (int)q; c->h_ipi[(size_t)q] = (int)q; }
    HIPCHK2(hipMemcpyAsync(c->pi, c->h_pi.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK2(hipMemcpyAsync(c->ipi, c->h_ipi.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK2(hipMemcpyAsync(c->Dq, c->Dq_src, (size_t)n * ld * esz, hipMemcpyDeviceToDevice, s));
    if (c->Lq) HIPCHK2(hipMemcpyAsync(c->Lq, c->Lq_src, (size_t)n * ld * esz, hipMemcpyDeviceToDevice, s));
    HIPCHK2(hipMemcpyAsync(c->diagq, c->diag_src, (size_t)n * sizeof(long long), hipMemcpyDeviceToDevice, s));
    if (c->derived && c->bits == 64 && !(rc_env_diag("RC_NO_PACK48") && atoi(rc_env_diag("RC_NO_PACK48")))) {
        HIPCHK2(hipMalloc(&c->Dq48, (size_t)n * ld * 6));
        k_pack48<<<4096, 256, 0, s>>>((const long long *)c->Dq, (size_t)n * ld / 2, (unsigned *)c->Dq48);
    }
    HIPCHK2(hipStreamSynchronize(s));
    HIPCHK2(hipGetLastError());
    cleanup();
#undef HIPCHK2
    return RC_OK;
}

static int32_t alloc_ctx(int64_t n, int32_t storage_bits, int32_t device_id, int64_t kcap, rc_ctx **out)
{
    if (n < 1 || n > (1 << 20)) return fail(nullptr, RC_ERR_ARG, "rc_create: n must be in 1..2^20 (got %lld)", (long long)n);
    if (storage_bits != 64 && storage_bits != 32) return fail(nullptr, RC_ERR_ARG, "rc_create: storage_bits must be 64 or 32");
    // kcap is the INITIAL slot capacity; it grows on demand up to min(n, RC_MAX_KCAP).  0 = automatic: 128 slots to begin with (the
    // resolver's tables then sit beside four row-reduction blocks per CU at their smallest), re-sized from the first state's cluster
    // count by rc_set_state (twice its K, at least 128).
    const bool kcap_auto = (kcap == 0);
    if (kcap_auto) kcap = std::min<int64_t>(n, 128);
    if (kcap < 1 || kcap > RC_WIDE_MAX_KCAP) return fail(nullptr, RC_ERR_ARG, "rc_create: kcap must be in 0..%d (0 = automatic)", RC_WIDE_MAX_KCAP);
    if (kcap > n) kcap = n;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) return fail(nullptr, RC_ERR_HIP, "no HIP device available (%s)", hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, RC_ERR_ARG, "rc_create: device_id %d out of range (0..%d)", device_id, ndev - 1);
    rc_ctx *c = new (std::nothrow) rc_ctx();
    if (!c) return fail(nullptr, RC_ERR_OOM, "rc_create: host allocation failed");
    c->dev = device_id;
    c->n = (int)n;
    c->ld = (int)(((n + 1023) / 1024) * 1024);
    c->bits = storage_bits;
    c->kcap = (int)kcap;
    c->kcap_auto = kcap_auto;
    c->kcap_max = (int)std::min<int64_t>(n, RC_WIDE_MAX_KCAP);   // (beyond RC_MAX_KCAP the context is wide: k_sweep_wide)
    c->kcap_fixed = rc_env("RC_KCAP_FIXED") && atoi(rc_env("RC_KCAP_FIXED"));
    c->dbg = rc_env_diag("RC_DEBUG_FLAGS") ? atoi(rc_env_diag("RC_DEBUG_FLAGS")) : 0;
    if (rc_env("RC_NO_PRUNE") && atoi(rc_env("RC_NO_PRUNE"))) c->opt_prune = 0;
    else if (rc_env("RC_PRUNE_ALWAYS") && atoi(rc_env("RC_PRUNE_ALWAYS"))) c->opt_prune = 1;
    if (rc_env("RC_CHAIN_WORKERS")) c->opt_chain_workers = std::max(0, atoi(rc_env("RC_CHAIN_WORKERS")));
    if (rc_env("RC_CHAIN_DEPTH")) c->opt_chain_depth = std::max(0, atoi(rc_env("RC_CHAIN_DEPTH")));
    if (rc_env("RC_CHAIN_PIPELINE")) c->opt_chain_pipeline = atoi(rc_env("RC_CHAIN_PIPELINE")) != 0;
    c->bulk_rows_forced = rc_env_diag("RC_BULK_ROWS") != nullptr;
    c->sm_profile = rc_env_diag("RC_SM_PROFILE") != nullptr;
    c->prefetch = !(rc_env_diag("RC_NO_PREFETCH") && atoi(rc_env_diag("RC_NO_PREFETCH")));
    c->relayout = !(rc_env("RC_NO_RELAYOUT") && atoi(rc_env("RC_NO_RELAYOUT")));
    c->res_one_stream = rc_env("RC_RES_ONE_STREAM") ? atoi(rc_env("RC_RES_ONE_STREAM")) != 0 : (n <= RC_RES_ONE_STREAM_MAX_N);
    if (rc_env("RC_SYM_VARIANT")) c->sym_variant = atoi(rc_env("RC_SYM_VARIANT"));
    if (rc_env_diag("RC_SYMW_PER_CU")) c->symw_per_cu = std::max(1, atoi(rc_env_diag("RC_SYMW_PER_CU")));
    if (rc_env_diag("RC_SYML_PAD")) c->syml_pad = (size_t)std::max(0, atoi(rc_env_diag("RC_SYML_PAD")));
    if (rc_env_diag("RC_SW_COARSE")) c->sw_coarse = std::min(128, std::max(8, atoi(rc_env_diag("RC_SW_COARSE")) & ~3));
    if (rc_env("RC_BULK_KERNEL")) c->bulk_kernel = !strcmp(rc_env("RC_BULK_KERNEL"), "sym") ? 1 : (!strcmp(rc_env("RC_BULK_KERNEL"), "perm") ? 0 : -1);
    if (rc_env_diag("RC_RES_THREADS")) { const int rt_ = atoi(rc_env_diag("RC_RES_THREADS")); c->res_threads = (rt_ == 64 || rt_ == 128 || rt_ == 256 || rt_ == 768 || rt_ == 1024) ? rt_ : 512; }
    if (rc_env_diag("RC_SYM32_TR")) c->sym32_tr = atoi(rc_env_diag("RC_SYM32_TR")) == 32 ? 32 : 16;
    c->sym32_bpc = c->sym32_tr == 16 ? 3 : 2;
    if (rc_env_diag("RC_SYM32_BPC")) c->sym32_bpc = std::max(1, std::min(4, atoi(rc_env_diag("RC_SYM32_BPC"))));
    if (rc_env_diag("RC_SYM_ITEM_TILES")) c->sym_item_tiles = std::max(1, atoi(rc_env_diag("RC_SYM_ITEM_TILES")));
    *out = c;
    return RC_OK;
}

// Symmetric-kernel variant of a context: RC_SYM_VARIANT, else the wave-autonomous kernel when logD is derived and the block-tiled
// ones when it is stored.
static int sym_variant_of(const rc_ctx *c) { return c->sym_variant >= 0 ? c->sym_variant : (c->derived ? 3 : 0); }
// the wave-autonomous kernels (three 40-42 KiB blocks per CU) are the symmetric kernel of this context: 3 = k_bulk_syml2 (round 3),
// 2 = k_bulk_syml (round 2)
static bool uses_syml(const rc_ctx *c) { return sym_variant_of(c) >= 2; }

// Unit lists of k_bulk_syml2 for `cap_blocks` resident 4-wave blocks.  Column block J (128 columns) holds the rows 0 .. 128 J + 127
// of the upper triangle.  Rows [0, 128 J) lie entirely above the diagonal block: FAST units of g rows (a multiple of 8; the rest of a
// block is a shorter unit) — no mask of any kind in their tiles.  The diagonal block itself (and every unit of a ragged last column
// block) is SLOW: the round-2 unit code with its triangle masks, ~3x the cost per 4-row tile, so it is cut fine (RC_S2_SLOW_ROWS
// rows per unit).  The launch lasts as long as its slowest wave, so the units are dealt to the waves by cost — longest first, each
// to the least loaded wave — and a wave reads its share as a contiguous range of each list.  g is chosen for the shortest makespan.
#ifndef RC_S2_SLOW_ROWS
#define RC_S2_SLOW_ROWS 16
#endif
static int32_t build_syml2_lists(rc_ctx *c, int per_cu = 0)
{
    const int n = c->n, ncb = (n + RC_SW_COLS - 1) / RC_SW_COLS, cap_blocks = (per_cu > 0 ? per_cu : c->symw_per_cu) * c->num_cus, nwaves = 4 * cap_blocks;
    struct Unit { int4 u; int cost; bool fast; };
    const int slow_factor = rc_env_diag("RC_S2_SLOW_COST") ? std::max(1, atoi(rc_env_diag("RC_S2_SLOW_COST"))) : 3;
    auto make_units = [&](int g, std::vector<Unit> &out) {
        out.clear();
        for (int J = ncb - 1; J >= 0; --J) {
            const int c0 = J * RC_SW_COLS, rows = std::min(c0 + RC_SW_COLS, n);
            const int fast_rows = rows & ~7;                             // (diagonal block and padding columns included: the fast path masks them)
            for (int a0 = 0; a0 < fast_rows; a0 += g) {
                const int a1 = std::min(a0 + g, fast_rows);
                out.push_back({make_int4(c0, a0, a1, 0), (a1 - a0) / 4 + 2, true});
            }
            for (int a0 = fast_rows; a0 < rows; a0 += RC_S2_SLOW_ROWS) {
                const int a1 = std::min(a0 + RC_S2_SLOW_ROWS, rows);
                out.push_back({make_int4(c0, a0, a1, 0), slow_factor * ((a1 - a0 + 3) / 4) + 3, false});
            }
        }
    };
    auto schedule = [&](std::vector<Unit> &units, std::vector<int> &owner) -> long long {   // longest first, to the least loaded wave
        std::vector<int> order(units.size());
        for (size_t q = 0; q < order.size(); ++q) order[q] = (int)q;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return units[(size_t)x].cost > units[(size_t)y].cost; });
        std::vector<std::pair<long long, int>> heap;                   // (load, wave), min-heap
        heap.reserve((size_t)nwaves);
        for (int w = 0; w < nwaves; ++w) heap.push_back({0, w});
        auto cmp = [](const std::pair<long long, int> &x, const std::pair<long long, int> &y) { return x > y; };
        std::make_heap(heap.begin(), heap.end(), cmp);
        owner.assign(units.size(), 0);
        long long makespan = 0;
        for (int q : order) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            auto &top = heap.back();
            owner[(size_t)q] = top.second;
            top.first += units[(size_t)q].cost;
            makespan = std::max(makespan, top.first);
            std::push_heap(heap.begin(), heap.end(), cmp);
        }
        return makespan;
    };
    std::vector<Unit> units, best_units;
    std::vector<int> owner, best_owner;
    long long best = -1;
    int best_g = 64;
    const int g_lo = c->sw_coarse > 0 ? (c->sw_coarse + 7) / 8 * 8 : 16, g_hi = c->sw_coarse > 0 ? g_lo : 256;
    for (int g = g_lo; g <= g_hi; g += 8) {
        make_units(g, units);
        const long long mk = schedule(units, owner);
        if (best < 0 || mk < best || (mk == best && g > best_g)) { best = mk; best_g = g; best_units = units; best_owner = owner; }
    }
    // group by wave (stable: heavy column blocks first inside a wave), fast and slow lists apart
    std::vector<int4> fast, slow;
    std::vector<int> wfast((size_t)nwaves + 1, 0), wslow((size_t)nwaves + 1, 0);
    for (size_t q = 0; q < best_units.size(); ++q) (best_units[q].fast ? wfast : wslow)[(size_t)best_owner[q] + 1]++;
    for (int w = 0; w < nwaves; ++w) { wfast[(size_t)w + 1] += wfast[(size_t)w]; wslow[(size_t)w + 1] += wslow[(size_t)w]; }
    fast.resize((size_t)wfast[(size_t)nwaves]); slow.resize((size_t)wslow[(size_t)nwaves]);
    {
        std::vector<int> pf(wfast.begin(), wfast.end() - 1), ps(wslow.begin(), wslow.end() - 1);
        for (size_t q = 0; q < best_units.size(); ++q) {
            const int w = best_owner[q];
            if (best_units[q].fast) fast[(size_t)pf[(size_t)w]++] = best_units[q].u; else slow[(size_t)ps[(size_t)w]++] = best_units[q].u;
        }
    }
    // RC_S2_STAGGER=1 (experiment, off): each fast unit cut at a row that differs from wave to wave and taken lower part last, so that
    // the flushes of a launch spread over its duration instead of coming at the same moments in every wave.  Measured: no gain
    // (74.9 against 73.5 us) — the ~18 us the atomics cost a launch (55 us without them) are not a drain at its end: 18 MB of 64-bit
    // atomics at the ~1 TB/s the memory side executes them is a per-CU occupancy of the vector memory path, spread or not.
    if (rc_env_diag("RC_S2_STAGGER") && atoi(rc_env_diag("RC_S2_STAGGER"))) {
        std::vector<int4> fast2;
        std::vector<int> wfast2((size_t)nwaves + 1, 0);
        fast2.reserve(2 * fast.size());
        for (int w = 0; w < nwaves; ++w) {
            wfast2[(size_t)w] = (int)fast2.size();
            for (int q = wfast[(size_t)w]; q < wfast[(size_t)w + 1]; ++q) {
                const int4 u = fast[(size_t)q];
                const int groups = (u.z - u.y) / 8;
                if (groups < 4) { fast2.push_back(u); continue; }
                const unsigned h = (unsigned)w * 2654435761u + (unsigned)q * 40503u;
                const int cut = u.y + 8 * (1 + (int)((h >> 8) % (unsigned)(groups - 1)));
                fast2.push_back(make_int4(u.x, cut, u.z, 0));
                fast2.push_back(make_int4(u.x, u.y, cut, 0));
            }
        }
        wfast2[(size_t)nwaves] = (int)fast2.size();
        fast.swap(fast2); wfast.swap(wfast2);
    }
    for (void *pfree : {(void *)c->ufast, (void *)c->uslow, (void *)c->wfast, (void *)c->wslow})
        if (pfree) (void)hipFree(pfree);
    c->ufast = c->uslow = nullptr; c->wfast = c->wslow = nullptr;
    HIPCHK(c, hipMalloc(&c->ufast, std::max<size_t>(1, fast.size()) * sizeof(int4)));
    HIPCHK(c, hipMalloc(&c->uslow, std::max<size_t>(1, slow.size()) * sizeof(int4)));
    HIPCHK(c, hipMalloc(&c->wfast, wfast.size() * sizeof(int)));
    HIPCHK(c, hipMalloc(&c->wslow, wslow.size() * sizeof(int)));
    if (!fast.empty()) HIPCHK(c, hipMemcpy(c->ufast, fast.data(), fast.size() * sizeof(int4), hipMemcpyHostToDevice));
    if (!slow.empty()) HIPCHK(c, hipMemcpy(c->uslow, slow.data(), slow.size() * sizeof(int4), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->wfast, wfast.data(), wfast.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->wslow, wslow.data(), wslow.size() * sizeof(int), hipMemcpyHostToDevice));
    c->nfast = (int)fast.size(); c->nslow = (int)slow.size(); c->syml2_g = best_g;
    c->syml2_blocks = cap_blocks;      // every resident wave has its (possibly empty) share
    if (rc_env_diag("RC_SM_PROFILE"))
        fprintf(stderr, "[rc_create] k_bulk_syml2: %d fast units of %d rows, %d slow units of %d rows over %d waves, makespan %lld tile costs\n",
                c->nfast, best_g, c->nslow, RC_S2_SLOW_ROWS, nwaves, best);
    return RC_OK;
}

// The lists for two blocks per CU (moving regime, enqueue_bulk): built by the same routine into the context's primary fields,
// which are put back afterwards.
static int32_t build_syml2_alt(rc_ctx *c)
{
    if (c->symw_per_cu <= 2 || (rc_env_diag("RC_S2_NO_ALT") && atoi(rc_env_diag("RC_S2_NO_ALT")))) return RC_OK;
    int4 *uf = c->ufast, *us = c->uslow;
    int *wf = c->wfast, *ws = c->wslow;
    const int nf = c->nfast, ns = c->nslow, nb = c->syml2_blocks, g = c->syml2_g;
    c->ufast = c->uslow = nullptr; c->wfast = c->wslow = nullptr;
    const int32_t rc = build_syml2_lists(c, rc_env_diag("RC_S2_ALT_PER_CU") ? std::max(1, atoi(rc_env_diag("RC_S2_ALT_PER_CU"))) : 2);
    c->s2alt.ufast = c->ufast; c->s2alt.uslow = c->uslow; c->s2alt.wfast = c->wfast; c->s2alt.wslow = c->wslow;
    c->s2alt.nfast = c->nfast; c->s2alt.nslow = c->nslow; c->s2alt.blocks = c->syml2_blocks;
    c->ufast = uf; c->uslow = us; c->wfast = wf; c->wslow = ws; c->nfast = nf; c->nslow = ns; c->syml2_blocks = nb; c->syml2_g = g;
    return rc;
}

// launch geometry and LDS attributes that depend on (n, kcap, bits)
static int32_t finish_create(rc_ctx *c)
{
    if (c->bits == 64 && !c->ufast) {
        int32_t rcl = build_syml2_lists(c);
        if (rcl == RC_OK) rcl = build_syml2_alt(c);
        if (rcl != RC_OK) return rcl;
    }
    if (!c->registered) { res_register(c); c->registered = true; }
    if (c->wide) {
        // no resolver, no LDS tables: only the geometry of the full-read row reduction (it fills the table once per rc_set_state)
        const int nchunks_w = (c->n + RC_PTS - 1) / RC_PTS;
        c->G = std::max(1, std::min(nchunks_w, c->num_cus));
        const int col_chunks = c->ld / (c->bits == 64 ? 512 : 1024);
        c->rows_per_split = std::max(16, std::min(512, (c->n + std::max(1, 512 / col_chunks) - 1) / std::max(1, 512 / col_chunks)));
        c->bulk_lds = 0;
        (void)hipFuncSetAttribute((const void *)k_blocksums, hipFuncAttributeMaxDynamicSharedMemorySize, RC_BS_TILE * 4 * (int)sizeof(u64));
        return RC_OK;
    }
    {
        // Resolver batch capacity.  The resolver of sweep t has to be resident beside the row-reduction blocks of sweep t+1
        // (its grid barrier needs every block at once; behind persistent reduction blocks it would wait for the whole
        // reduction — config 5: 1.34 ms per resolver instead of 0.3).  LDS beside the reduction: three 40 KiB blocks of the
        // wave-autonomous kernel (64-bit, logD derived), two 65 KiB blocks of the block-tiled kernels otherwise; the
        // full-read kernel of small problems sizes itself around the resolver.  Largest capacity whose tables fit.
        const bool syml = uses_syml(c);
        const size_t beside = syml ? (size_t)c->symw_per_cu * 40960 : (c->bits != 64 && c->sym32_tr == 16) ? (size_t)c->sym32_bpc * 36864 : (size_t)2 * 69632;   // (k_bulk_sym32: 67,864 B per block, k_bulk_sym: 67,288 B; measured: beside two of them 24.6 KB of tables become resident, 26.5 KB do not — 4 KiB allocation granules)
        const size_t avail = 160 * 1024 > beside + 1024 ? 160 * 1024 - beside - 1024 : 0;
        c->maxb = RC_MAXB;
        if (rc_env("RC_RES_MAXB")) c->maxb = std::max(16, std::min(RC_MAXB, atoi(rc_env("RC_RES_MAXB"))));
        else if (c->n > 4096 && tab_bytes(c->kcap, c->n, RC_RES_THREADS / 64, RC_MAXB) > avail)
            for (int mb : {384, 256, 192, 128, 96, 64})
                if (tab_bytes(c->kcap, c->n, RC_RES_THREADS / 64, mb) <= avail) { c->maxb = mb; break; }
        // a slot capacity whose tables cannot sit beside the reduction at any batch capacity (kcap >= 1024) must at least fit the CU:
        // kcap = 4096 needs 160 KiB at 512 entries per batch for n >= 8192, 149 KiB at 128
        // (RC_RES_MAXB is an upper bound: a forced batch capacity that does not fit the CU with this slot capacity is lowered like the default)
        if (tab_bytes(c->kcap, c->n, RC_RES_THREADS / 64, c->maxb) > 160 * 1024)
            for (int mb : {384, 256, 192, 128, 96, 64, 32, 16})
                if (mb < c->maxb && tab_bytes(c->kcap, c->n, RC_RES_THREADS / 64, mb) <= 160 * 1024) { c->maxb = mb; break; }
        if (rc_env_diag("RC_SM_PROFILE"))
            fprintf(stderr, "[rc_create] resolver batch capacity %d: tables %zu B (512: %zu, 256: %zu, 128: %zu, 64: %zu), %zu B free beside the row reduction\n", c->maxb,
                    tab_bytes(c->kcap, c->n, RC_RES_THREADS / 64, c->maxb), tab_bytes(c->kcap, c->n, RC_RES_THREADS / 64, 512), tab_bytes(c->kcap, c->n, RC_RES_THREADS / 64, 256),
                    tab_bytes(c->kcap, c->n, RC_RES_THREADS / 64, 128), tab_bytes(c->kcap, c->n, RC_RES_THREADS / 64, 64), avail);
    }
    const int nchunks = (c->n + RC_PTS - 1) / RC_PTS;
    c->G = std::max(1, std::min(nchunks, c->num_cus));
    {
        // k_bulk split length: aim for >= 512 workgroups, but keep splits long (each split boundary costs a
        // round of 64-bit atomic flushes): 256 rows at n = 8192 (tools/bulk_tune.hip)
        const int col_chunks = c->ld / (c->bits == 64 ? 512 : 1024);
        const int splits_target = std::max(1, 512 / col_chunks);
        c->rows_per_split = std::max(16, std::min(512, (c->n + splits_target - 1) / splits_target));
        if (rc_env_diag("RC_BULK_ROWS")) c->rows_per_split = std::max(1, atoi(rc_env_diag("RC_BULK_ROWS")));
        const int per_cu = rc_env_diag("RC_BULK_PER_CU") ? atoi(rc_env_diag("RC_BULK_PER_CU")) : 2;
        // ... and the LDS the resolver's tables need must stay free beside them (160 KiB per CU): with the tables at 25-35 KiB
        // a fixed 150 KiB for the reduction left the resolver waiting for reduction blocks to retire
        const size_t lds_res = std::max(tab_bytes(c->kcap, c->n, RC_RES_THREADS / 64, c->maxb), 2 * sizeof(int) * (size_t)c->kcap);
        const size_t avail = lds_res + 4096 < 150 * 1024 ? std::min<size_t>(150 * 1024, 160 * 1024 - lds_res - 2048) : 16 * 1024;
        c->bulk_lds = per_cu > 0 ? (size_t)((avail / per_cu) & ~(size_t)1023) : 0;
        if (c->bulk_lds > 64 * 1024) {
            (void)hipFuncSetAttribute(c->bits == 64 ? (const void *)k_bulk<long long> : (const void *)k_bulk<int>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->bulk_lds);
            (void)hipFuncSetAttribute((const void *)k_bulk<long long, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->bulk_lds);
        }
    }
    // kernels whose dynamic LDS can exceed the 64 KiB default (large kcap)
    size_t lds_r = std::max(tab_bytes(c->kcap, c->n, RC_RES_THREADS / 64, c->maxb), 2 * sizeof(int) * (size_t)c->kcap);
    {   // (the wider block of the incremental mode needs a little more reduction scratch: used only while it fits the CU)
        const size_t wide = tab_bytes(c->kcap, c->n, RC_RES_THREADS_INC / 64, c->maxb);
        if (wide <= 160 * 1024) lds_r = std::max(lds_r, wide);
    }
    const size_t lds_d = std::max(tab_bytes(c->kcap, c->n, 1), 2 * sizeof(int) * (size_t)c->kcap);
    const size_t lds_b = (size_t)std::min(c->kcap, RC_BS_TILE) * 4 * sizeof(u64);
    // (the attribute belongs to the function, not to the context: a second context with smaller tables must not lower it under a first one's)
    hipError_t e1 = hipFuncSetAttribute((const void *)k_resolve, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipError_t e2 = hipFuncSetAttribute((const void *)k_blocksums, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
    hipError_t e3 = hipFuncSetAttribute((const void *)k_derive, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_d);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || lds_r > 160 * 1024 || lds_b > 160 * 1024)
        return fail(c, RC_ERR_ARG, "rc_create: kcap=%d needs more LDS than a CU has", c->kcap);
    return RC_OK;
}

extern "C" int32_t rc_create(int64_t n, const double *D, const double *logD_or_null, int32_t storage_bits,
                             int32_t device_id, int64_t kcap, rc_ctx **out)
{
    if (!out) return fail(nullptr, RC_ERR_ARG, "rc_create: out is NULL");
    *out = nullptr;
    if (!D) return fail(nullptr, RC_ERR_ARG, "rc_create: D is NULL");
    rc_ctx *c = nullptr;
    {
        int32_t rc0 = alloc_ctx(n, storage_bits, device_id, kcap, &c);
        if (rc0 != RC_OK) return rc0;
    }
    int32_t rc = create_impl(c, n, D, logD_or_null);
    if (rc == RC_OK) rc = finish_create(c);
    if (rc != RC_OK) {
        snprintf(g_err, sizeof(g_err), "%s", c->err);
        if (c->registered) { res_unregister(c); c->registered = false; }
        free_all(c);
        return rc;
    }
    *out = c;
    return RC_OK;
}

// MCMCData(points) constructor, src/types.jl:159-162: D = pairwise(Euclidean(), makematrix(pnts), dims=2), computed
// on the device (k_pairwise) — the n×n matrix never exists on the host.  points: n×dim row-major.
extern "C" int32_t rc_create_from_points(int64_t n, int64_t dim, const double *points, int32_t storage_bits,
                                         int32_t device_id, int64_t kcap, rc_ctx **out)
{
    if (!out) return fail(nullptr, RC_ERR_ARG, "rc_create_from_points: out is NULL");
    *out = nullptr;
    if (!points) return fail(nullptr, RC_ERR_ARG, "rc_create_from_points: points is NULL");
    if (dim < 1 || dim > (1 << 20)) return fail(nullptr, RC_ERR_ARG, "rc_create_from_points: dim must be in 1..2^20");
    rc_ctx *c = nullptr;
    int32_t rc = alloc_ctx(n, storage_bits, device_id, kcap, &c);
    if (rc != RC_OK) return rc;
    rc = create_impl(c, n, nullptr, nullptr, points, dim);
    if (rc == RC_OK) rc = finish_create(c);
    if (rc != RC_OK) {
        snprintf(g_err, sizeof(g_err), "%s", c->err);
        if (c->registered) { res_unregister(c); c->registered = false; }
        free_all(c);
        return rc;
    }
    *out = c;
    return RC_OK;
}

// The fixed-point matrix held on the device, as doubles (which = 0: D, 1: logD); value = q·2^-e exactly.
extern "C" int32_t rc_get_matrix(rc_ctx *c, int32_t which, double *out_n_by_n)
{
    if (!c || !out_n_by_n) return fail(c, RC_ERR_ARG, "rc_get_matrix: NULL argument");
    if (which != 0 && which != 1) return fail(c, RC_ERR_ARG, "rc_get_matrix: which must be 0 (D) or 1 (logD)");
    HIPCHK(c, hipSetDevice(c->dev));
    const size_t nn = (size_t)c->n * c->n;
    double *tmp = nullptr;
    HIPCHK(c, hipMalloc(&tmp, nn * sizeof(double)));
    const int gb = (int)std::min<size_t>((nn + 255) / 256, 8192);
    const void *Q = which ? c->Lq_src : c->Dq_src;  // the caller's point order
    const double scale = std::ldexp(1.0, -(which ? c->eL : c->eD));
    if (which && c->derived)
        k_derived_matrix<<<gb, 256, 0, c->sA>>>((const long long *)c->Dq_src, c->n, c->ld, c->eD, std::ldexp(1.0, c->eL), scale, c->ltab, tmp);
    else if (c->bits == 64) k_dequantize<long long><<<gb, 256, 0, c->sA>>>((const long long *)Q, c->n, c->ld, scale, tmp);
    else k_dequantize<int><<<gb, 256, 0, c->sA>>>((const int *)Q, c->n, c->ld, scale, tmp);
    hipError_t e = hipMemcpyAsync(out_n_by_n, tmp, nn * sizeof(double), hipMemcpyDeviceToHost, c->sA);
    if (e == hipSuccess) e = hipStreamSynchronize(c->sA);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(c, RC_ERR_HIP, "rc_get_matrix: %s", hipGetErrorString(e));
    return RC_OK;
}

// Selected rows of the same matrices (rc_get_matrix needs n² doubles on the host: 8 GiB at n = 32768).
extern "C" int32_t rc_get_matrix_rows(rc_ctx *c, int32_t which, const int64_t *rows, int64_t nrows, double *out)
{
    if (!c || !rows || !out) return fail(c, RC_ERR_ARG, "rc_get_matrix_rows: NULL argument");
    if (which != 0 && which != 1) return fail(c, RC_ERR_ARG, "rc_get_matrix_rows: which must be 0 (D) or 1 (logD)");
    if (nrows < 0 || nrows > 65535) return fail(c, RC_ERR_ARG, "rc_get_matrix_rows: nrows must be in 0..65535");
    if (nrows == 0) return RC_OK;
    std::vector<int> r32((size_t)nrows);
    for (int64_t q = 0; q < nrows; ++q) {
        if (rows[q] < 0 || rows[q] >= c->n) return fail(c, RC_ERR_ARG, "rc_get_matrix_rows: row %lld outside 0..n-1", (long long)rows[q]);
        r32[(size_t)q] = (int)rows[q];
    }
    HIPCHK(c, hipSetDevice(c->dev));
    int *drows = nullptr;
    double *tmp = nullptr;
    HIPCHK(c, hipMalloc(&drows, (size_t)nrows * sizeof(int)));
    hipError_t e = hipMalloc(&tmp, (size_t)nrows * c->n * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(drows, r32.data(), (size_t)nrows * sizeof(int), hipMemcpyHostToDevice, c->sA);
    if (e == hipSuccess) {
        const bool derive = which && c->derived;
        const void *Q = (which && !derive) ? c->Lq_src : c->Dq_src;
        const double scale = std::ldexp(1.0, -(which ? c->eL : c->eD));
        dim3 g((unsigned)std::min(64, (c->n + 255) / 256), (unsigned)nrows);
        if (c->bits == 64) k_get_rows<long long><<<g, 256, 0, c->sA>>>((const long long *)Q, drows, c->n, c->ld, scale, derive ? 1 : 0, c->eD, std::ldexp(1.0, c->eL), c->ltab, tmp);
        else k_get_rows<int><<<g, 256, 0, c->sA>>>((const int *)Q, drows, c->n, c->ld, scale, 0, c->eD, 0.0, c->ltab, tmp);
        e = hipMemcpyAsync(out, tmp, (size_t)nrows * c->n * sizeof(double), hipMemcpyDeviceToHost, c->sA);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->sA);
    (void)hipFree(drows);
    if (tmp) (void)hipFree(tmp);
    if (e != hipSuccess) return fail(c, RC_ERR_HIP, "rc_get_matrix_rows: %s", hipGetErrorString(e));
    return RC_OK;
}

// Fixed-point row totals Σ_j D[i,j] and Σ_j logD[i,j] (value = q·2^-e) of the matrices as the caller gave them — k_rowtotals.
// the score logarithms of the resolver on caller-supplied arguments (tests: against libm, tests/test_gpu_logs.py)
__global__ void k_flog_eval(const double *__restrict__ x, long long m, int which, const double2 *__restrict__ tab, double *__restrict__ out)
{
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < m) out[q] = which == 0 ? rc_flog(x[q], tab) : which == 1 ? rc_flog1p(x[q], tab) : rc_gumbel(x[q], tab);
}
extern "C" int32_t rc_debug_flog(rc_ctx *c, int32_t which, const double *x, int64_t m, double *out)
{
    if (!c || !x || !out || m < 0 || which < 0 || which > 2) return fail(c, RC_ERR_ARG, "rc_debug_flog: bad argument (which: 0 log, 1 log1p, 2 -log(-log))");
    if (m == 0) return RC_OK;
    HIPCHK(c, hipSetDevice(c->dev));
    double *d = nullptr;
    HIPCHK(c, hipMalloc(&d, 2 * (size_t)m * sizeof(double)));
    hipError_t e = hipMemcpyAsync(d, x, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->sA);
    if (e == hipSuccess) {
        k_flog_eval<<<(unsigned)((m + 255) / 256), 256, 0, c->sA>>>(d, (long long)m, which, c->flt, d + m);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d + m, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, c->sA);
    if (e == hipSuccess) e = hipStreamSynchronize(c->sA);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, RC_ERR_HIP, "rc_debug_flog: %s", hipGetErrorString(e));
    return RC_OK;
}

extern "C" int32_t rc_debug_rowtotals(rc_ctx *c, int64_t *totD_q, int64_t *totL_q)
{
    if (!c || !totD_q || !totL_q) return fail(c, RC_ERR_ARG, "rc_debug_rowtotals: NULL argument");
    HIPCHK(c, hipSetDevice(c->dev));
    long long *d = nullptr;
    HIPCHK(c, hipMalloc(&d, 2 * (size_t)c->n * sizeof(long long)));
    const bool derive = c->derived && !c->Lq_src;
    if (c->bits == 64) k_rowtotals<long long><<<c->n, 256, 0, c->sA>>>((const long long *)c->Dq_src, (const long long *)c->Lq_src, c->n, c->ld, derive ? 1 : 0, c->eD, std::ldexp(1.0, c->eL), c->ltab, d, d + c->n);
    else k_rowtotals<int><<<c->n, 256, 0, c->sA>>>((const int *)c->Dq_src, (const int *)c->Lq_src, c->n, c->ld, 0, c->eD, 0.0, c->ltab, d, d + c->n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(totD_q, d, (size_t)c->n * sizeof(long long), hipMemcpyDeviceToHost, c->sA);
    if (e == hipSuccess) e = hipMemcpyAsync(totL_q, d + c->n, (size_t)c->n * sizeof(long long), hipMemcpyDeviceToHost, c->sA);
    if (e == hipSuccess) e = hipStreamSynchronize(c->sA);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, RC_ERR_HIP, "rc_debug_rowtotals: %s", hipGetErrorString(e));
    return RC_OK;
}

extern "C" int32_t rc_set_params(rc_ctx *c, const rc_params *P)
{
    if (!c || !P) return fail(c, RC_ERR_ARG, "rc_set_params: NULL argument");
    if (!(P->delta1 > 0 && P->delta2 > 0 && P->alpha > 0 && P->beta > 0 && P->zeta > 0 && P->gamma > 0))
        return fail(c, RC_ERR_ARG, "rc_set_params: likelihood hyperparameters must be positive");
    if (P->maxK < 0) return fail(c, RC_ERR_ARG, "rc_set_params: maxK must be >= 0");
    HIPCHK(c, hipSetDevice(c->dev));
    HIPCHK(c, hipStreamSynchronize(c->sA));
    HIPCHK(c, hipStreamSynchronize(c->sB));
    HIPCHK(c, hipStreamSynchronize(c->sB2));
    c->P = *P;
    c->llc = rc_ctx::LLCache();
    c->params_version++;
    c->ll_version = -1;   // cached log-likelihood / block sums belong to the old parameters
    c->B_version = -2;
    // size table (see DESIGN.md "Score arithmetic"): long double on the host, once per parameter set
    std::vector<double> A((size_t)c->n + 1);
    const long double d1 = P->delta1, d2 = P->delta2, al = P->alpha, be = P->beta, ze = P->zeta, ga = P->gamma;
    const long double lga = lgammal(al), lgz = lgammal(ze), lgd1 = lgammal(d1), lgd2 = lgammal(d2);
    const long double lb = logl(be), lg = logl(ga);
    A[0] = 0.0;
    for (int s = 1; s <= c->n; ++s) {
        const long double S = (long double)s;
        const long double t1 = lgammal(al + d1 * S) - lga - d1 * S * lb - S * lgd1;
        const long double t2 = lgammal(ze + d2 * S) - lgz - d2 * S * lg - S * lgd2;
        A[(size_t)s] = (double)(t1 - (P->repulsion ? t2 : 0.0L) + logl((S + 1) / S));
    }
    HIPCHK(c, hipMemcpyAsync(c->A, A.data(), A.size() * sizeof(double), hipMemcpyHostToDevice, c->sA));
    HIPCHK(c, hipStreamSynchronize(c->sA));
    c->have_params = true;
    return RC_OK;
}

static int32_t drain_events(rc_ctx *c)
{
    for (auto &e : c->ev_pending) {
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, e.first, e.second));
        c->bulk_ms += std::max(0.0, (double)ms - c->ev_overhead_ms);
        c->bulk_launches += 1;
        c->ev_free.push_back(e);
    }
    c->ev_pending.clear();
    return RC_OK;
}

static int32_t recover_capacity(rc_ctx *c);

// Waits for the resolve/observable stream (and for the k_bulk stream too when `both`), then surfaces device errors.  A sweep
// that ran out of slots is not an error: the tables are grown, the sweep resumed and the sweeps behind it replayed here.
static int32_t sync_and_check(rc_ctx *c, bool both = false)
{
  for (;;) {
    if (c->s_res_last && c->s_res_last != c->sA) HIPCHK(c, hipStreamSynchronize(c->s_res_last));   // the last sweep's resolver
    HIPCHK(c, hipStreamSynchronize(c->sA));
    if (both) {
        HIPCHK(c, hipStreamSynchronize(c->sB));
        HIPCHK(c, hipStreamSynchronize(c->sB2));
        int32_t rc = drain_events(c);
        if (rc != RC_OK) return rc;
    }
    HIPCHK(c, hipGetLastError());
    c->last.K = c->hsum->K; c->last.n_changes = c->hsum->n_changes; c->last.n_rounds = c->hsum->n_rounds;
    c->last.err = c->hsum->err; c->last.slot_hi = c->hsum->slot_hi;
    if (c->last.err & RC_DERR_BARRIER) {
        // some blocks of the resolver gave up waiting for the others (another process holding part of the GPU?): the sweep is
        // half-committed, so the state is void until rc_set_state installs labels again
        c->have_state = false;
        return fail(c, RC_ERR_HIP, "grid barrier timed out inside the sweep kernel (is another process using this GPU? one chain per "
                                   "GPU); the label state is void: call rc_set_state before sweeping again");
    }
    if (c->last.err & RC_DERR_CAPACITY) {
        const int32_t rcr = recover_capacity(c);
        if (rcr != RC_OK) return rcr;
        continue;   // the resumed / replayed sweeps are in flight: wait for them (and recover again if they overflow again)
    }
    c->inflight.clear();   // every sweep enqueued so far is complete (resolvers are chained)
    return RC_OK;
  }
}

extern "C" int32_t rc_synchronize(rc_ctx *c)
{
    if (!c) return fail(c, RC_ERR_ARG, "rc_synchronize: NULL ctx");
    HIPCHK(c, hipSetDevice(c->dev));
    return sync_and_check(c, true);
}

// Slot capacity for `need` clusters: twice as many (room to move), at least 128, a power of two, at most kcap_max.
static int capacity_for(const rc_ctx *c, long long need)
{
    long long k = 128;
    while (k < 2 * need) k *= 2;
    if (need <= RC_MAX_KCAP) k = std::min<long long>(k, RC_MAX_KCAP);   // (as long as the clusters fit the fast path's tables the context stays on it)
    return (int)std::min<long long>(std::max<long long>(k, need), c->kcap_max);
}

// Re-sizes everything that depends on the slot capacity (device buffers, the resolver's LDS layout and batch capacity).  The
// streams must be drained; the label state is void afterwards — the caller installs labels with rc_set_state.
static int32_t resize_capacity(rc_ctx *c, int new_kcap)
{
    if (new_kcap == c->kcap) return RC_OK;
    const int old_kcap = c->kcap;
    c->have_state = false;                  // the old buffers go first: whatever happens below, the labels are gone from the device
    c->kcap = new_kcap;
    int32_t rc = alloc_slot_buffers(c);
    if (rc == RC_OK) rc = finish_create(c);
    if (rc != RC_OK) {
        // out of device memory (kcap = 4096 at n = 8192 is 2.4 GB) or of LDS: back to the old capacity, so that the context stays
        // usable after an rc_set_state that fits; if even that fails the context is dead and says so
        char msg[512];
        snprintf(msg, sizeof(msg), "%s", c->err);
        c->kcap = old_kcap;
        int32_t rc2 = alloc_slot_buffers(c);
        if (rc2 == RC_OK) rc2 = finish_create(c);
        if (rc2 != RC_OK) c->broken = true;
        snprintf(c->err, sizeof(c->err), "growing the slot capacity %d -> %d failed: %s%s", old_kcap, new_kcap, msg,
                 c->broken ? " (the context is unusable: destroy it)" : " (call rc_set_state again)");
        snprintf(g_err, sizeof(g_err), "%s", c->err);
        return rc;
    }
    c->n_grows++;
    c->B_version = -2; c->ll_version = -1;
    if (c->sm_profile) fprintf(stderr, "[redclust] slot capacity -> %d (batch capacity %d)\n", c->kcap, c->maxb);
    return RC_OK;
}

// derived tables after the slot tables changed (rc_set_state, apply_labels, rc_set_mode): in LDS on the fast path, in global memory for a wide context
static int32_t launch_derive(rc_ctx *c, const View &V, int rebuild_perm)
{
    if (c->wide) {
        k_derive_wide<<<1, 1024, 0, c->sA>>>(V, rebuild_perm);
    } else {
        const size_t lds = std::max(tab_bytes(c->kcap, c->n, 1), 2 * sizeof(int) * (size_t)c->kcap);
        k_derive<<<1, 1024, lds, c->sA>>>(V, rebuild_perm);
    }
    HIPCHK(c, hipGetLastError());
    return RC_OK;
}

extern "C" int32_t rc_set_state(rc_ctx *c, const int64_t *clusts)
{
    if (!c || !clusts) return fail(c, RC_ERR_ARG, "rc_set_state: NULL argument");
    if (c->broken) return fail(c, RC_ERR_STATE, "rc_set_state: the context lost its device buffers in a failed capacity growth; destroy it");
    HIPCHK(c, hipSetDevice(c->dev));
    HIPCHK(c, hipStreamSynchronize(c->sA));
    HIPCHK(c, hipStreamSynchronize(c->sB));
    HIPCHK(c, hipStreamSynchronize(c->sB2));
    if (c->sC) HIPCHK(c, hipStreamSynchronize(c->sC));
    if (!c->recovering) c->inflight.clear();
    const int n = c->n;
    {
        // the capacity follows the state: more clusters than slots -> grow; automatic capacity -> sized from the first state
        std::vector<unsigned char> seen((size_t)n + 1, 0);
        long long K0 = 0;
        for (int i = 0; i < n; ++i) {
            if (clusts[i] < 1 || clusts[i] > n) return fail(c, RC_ERR_ARG, "rc_set_state: label %lld of point %d outside 1..n", (long long)clusts[i], i + 1);
            if (!seen[(size_t)clusts[i]]) { seen[(size_t)clusts[i]] = 1; ++K0; }
        }
        if (K0 > c->kcap_max)
            return fail(c, RC_ERR_CAPACITY, "rc_set_state: %lld clusters, the library holds at most min(n, %d) = %d (slot ids are 16-bit)", K0, RC_WIDE_MAX_KCAP, c->kcap_max);
        if (!c->kcap_fixed && (K0 > c->kcap || (c->kcap_auto && !c->have_state && c->n_grows == 0 && capacity_for(c, K0) > c->kcap))) {
            int32_t rcg = resize_capacity(c, std::max(c->kcap, capacity_for(c, K0)));
            if (rcg != RC_OK) return rcg;
        } else if (!c->kcap_fixed && c->wide && K0 * 4 <= RC_MAX_KCAP) {
            // A wide context (more than 4096 slots: the sweep point by point on one workgroup, hundreds of milliseconds) whose state
            // has come down to a quarter of what the fast path holds goes back to it: the capacity follows the state downwards too
            // (a chain started from all singletons collapses to a few dozen clusters within a sweep or two).
            int32_t rcg = resize_capacity(c, capacity_for(c, K0));
            if (rcg != RC_OK) return rcg;
        }
    }
    // clustsizes = counts(clusts, 1:n), K = sum(clustsizes .> 0)  (types.jl:135-136); slots in label order
    std::vector<int> size_by_label((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) {
        if (clusts[i] < 1 || clusts[i] > n) return fail(c, RC_ERR_ARG, "rc_set_state: label %lld of point %d outside 1..n", (long long)clusts[i], i + 1);
        size_by_label[(size_t)clusts[i]]++;
    }
    std::vector<int> slot_of_label((size_t)n + 1, -1), ssize((size_t)c->kcap, 0), slabel((size_t)c->kcap, 0);
    int K = 0;
    for (int lab = 1; lab <= n; ++lab)
        if (size_by_label[(size_t)lab] > 0) {
            if (K >= c->kcap) return fail(c, RC_ERR_CAPACITY, "rc_set_state: more than kcap=%d clusters", c->kcap);
            slot_of_label[(size_t)lab] = K;
            ssize[(size_t)K] = size_by_label[(size_t)lab];
            slabel[(size_t)K] = lab;
            ++K;
        }
    // Internal point order: points of a cluster contiguous (stable sort by label), so that the symmetric row reduction
    // meets few label runs whatever order the caller's points are in.  The sweep itself still visits the points in the
    // caller's order (k_resolve maps i -> pi[i]).
    std::vector<int> ipi((size_t)n), pi((size_t)n);
    for (int i = 0; i < n; ++i) ipi[(size_t)i] = i;
    if (c->relayout) {
        // Small clusters FIRST (then by label).  A moving chain holds dozens to hundreds of singletons.  As COLUMNS of the upper
        // triangle every one of them is a cluster of its own — no pre-added direction-2 path, one atomic per element and row — so
        // they belong where the columns are shortest: the first column block or two (128 / 256 rows above the diagonal), not
        // wherever their labels fall (sorted by label the 154 singletons of the sigma = 0.2 chain sit at the END: two column
        // blocks of 8000 rows each, 2.4 M element-wise atomics per launch, 260-340 us for k_bulk_syml2 instead of 70).  As ROWS they
        // cost a coalesced flush of direction 1 per row, in the first unit of every column block.  RC_LAYOUT_SMALL=0: by label only.
        {
            static const int small_max = rc_env_diag("RC_LAYOUT_SMALL") ? atoi(rc_env_diag("RC_LAYOUT_SMALL")) : 7;
            std::stable_sort(ipi.begin(), ipi.end(), [&](int a, int b) {
                const bool la = size_by_label[(size_t)clusts[a]] > small_max, lb = size_by_label[(size_t)clusts[b]] > small_max;
                return la != lb ? lb : clusts[a] < clusts[b];
            });
        }
        // Every cluster's run gets an EVEN length and hence an even start: the last point of each odd-sized cluster goes to the
        // tail of the order.  The row reduction gives a lane two adjacent columns (2 l, 2 l + 1); a cluster boundary at an odd
        // position splits a lane between two clusters, and such a lane's elements cannot go through the pre-added direction-2
        // path (k_bulk_syml2: element-wise atomics for every row of every unit of that column block — a third of all units with
        // boundaries at arbitrary positions).  The tail — one point per odd-sized cluster, every singleton — is ragged, but it is
        // a column block or two.  Exactness does not depend on any of this.
        // (Kept as an experiment, RC_EVEN_RUNS=1: the ragged tail it creates costs more than it saves once the boundary lanes are
        // handled inside the kernel — k_bulk_syml2's donor lanes.)
        if (uses_syml(c) && rc_env_diag("RC_EVEN_RUNS") && atoi(rc_env_diag("RC_EVEN_RUNS"))) {
            std::vector<int> body, tail;
            body.reserve((size_t)n);
            for (int w = 0; w < n;) {
                int e = w;
                while (e < n && clusts[ipi[(size_t)e]] == clusts[ipi[(size_t)w]]) ++e;
                const int keep = (e - w) & ~1;
                for (int q = w; q < w + keep; ++q) body.push_back(ipi[(size_t)q]);
                if (keep < e - w) tail.push_back(ipi[(size_t)e - 1]);
                w = e;
            }
            body.insert(body.end(), tail.begin(), tail.end());
            ipi.swap(body);
        }
    }
    for (int w = 0; w < n; ++w) pi[(size_t)ipi[(size_t)w]] = w;
    if (pi != c->h_pi) {
        c->h_pi = pi; c->h_ipi = ipi;
        c->n_relayouts++;
        HIPCHK(c, hipMemcpy(c->pi, pi.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->ipi, ipi.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
        dim3 g((unsigned)std::min(64, (n + 255) / 256), (unsigned)n);
        if (c->bits == 64) {
            k_relayout<long long><<<g, 256, 0, c->sA>>>((const long long *)c->Dq_src, c->ipi, n, c->ld, (long long *)c->Dq);
            if (c->Lq) k_relayout<long long><<<g, 256, 0, c->sA>>>((const long long *)c->Lq_src, c->ipi, n, c->ld, (long long *)c->Lq);
        } else {
            k_relayout<int><<<g, 256, 0, c->sA>>>((const int *)c->Dq_src, c->ipi, n, c->ld, (int *)c->Dq);
            k_relayout<int><<<g, 256, 0, c->sA>>>((const int *)c->Lq_src, c->ipi, n, c->ld, (int *)c->Lq);
        }
        k_gather_ll<<<(n + 255) / 256, 256, 0, c->sA>>>(c->diag_src, c->ipi, n, c->diagq);
        if (c->Dq48) k_pack48<<<4096, 256, 0, c->sA>>>((const long long *)c->Dq, (size_t)n * c->ld / 2, (unsigned *)c->Dq48);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->sA));
    }
    std::vector<int> so((size_t)n);
    for (int w = 0; w < n; ++w) so[(size_t)w] = slot_of_label[(size_t)clusts[ipi[(size_t)w]]];
    DevScalars s{};
    s.K = K;
    s.slot_hi = K;
    s.last_change_sweep = -1;
    HIPCHK(c, hipMemcpy(c->slot_of, so.data(), so.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->slot_size, ssize.data(), ssize.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->slot_label, slabel.data(), slabel.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->sc, &s, sizeof(s), hipMemcpyHostToDevice));
    // every S generation may be stale: clear them all (keeps the "rows of free slots are zero" invariant)
    for (int g = 0; g < 3; ++g) {
        if (!c->SD[g]) continue;   // (a wide context has one generation)
        HIPCHK(c, hipMemsetAsync(c->SD[g], 0, (size_t)c->kcap * c->ld * sizeof(long long), c->sA));
        HIPCHK(c, hipMemsetAsync(c->SL[g], 0, (size_t)c->kcap * c->ld * sizeof(long long), c->sA));
    }
    for (int g = 0; g < 2; ++g) {
        HIPCHK(c, hipMemsetAsync(c->keys[g], 0xFF, (size_t)(2 * n + 8) * sizeof(u64), c->sA));
        HIPCHK(c, hipMemsetAsync(c->cword[g], 0, 2 * ((size_t)(n + RC_PTS - 1) / RC_PTS + 1) * sizeof(u64), c->sA));
        HIPCHK(c, hipMemsetAsync(c->arrive[g], 0, RC_BAR_WORDS * sizeof(unsigned), c->sA));
        HIPCHK(c, hipMemsetAsync(c->work[g], 0, 64, c->sA));
    }
    View V = make_view(c);
    {
        int32_t rcd = launch_derive(c, V, 1);
        if (rcd != RC_OK) return rcd;
    }
    HIPCHK(c, hipStreamSynchronize(c->sA));
    if (c->wide) { c->incremental = true; c->inc_gen = 0; }   // one generation, corrected in place by k_sweep_wide (RC_MODE_FULL again once the context narrows)
    else c->incremental = c->want_incremental;
    c->last = s;
    c->t_next = 0;
    c->bulk_enq = -1;
    c->s_res_last = nullptr;
    c->state_version++;
    c->have_state = true;
    return RC_OK;
}

// Kernel choice.  The symmetric kernels read half the bytes but want the points of a cluster to be contiguous in the
// point order (few label runs); every kernel is exact for any labelling, so a stale run count only costs speed.
// Small problems are launch- and latency-bound: with separate launches the full-read kernel (one pass, no per-unit
// prologue) wins up to n = 4096 even though it reads and — in the derived mode — computes twice as much (measured:
// n = 4096: 58 vs 75 µs per sweep, n = 6000: 102 vs 70).
static bool choose_sym(const rc_ctx *c)
{
    static const int runs_div = rc_env_diag("RC_SYM_RUNS_DIV") ? std::max(1, atoi(rc_env_diag("RC_SYM_RUNS_DIV"))) : 32;   // (diag builds: the threshold of the choice)
    return c->bulk_kernel == 1 || (c->bulk_kernel < 0 && (long long)c->hsum->runs * runs_div <= (long long)c->n &&
                                   !(c->derived && c->n <= (sym_variant_of(c) == 3 ? 2560 : 4096)));
}

// Static unit list of the wave-autonomous symmetric reduction for `cap_blocks` resident 4-wave blocks.  Every column block
// is cut into units of gc rows, gc chosen so that the list is a whole number of rounds over the resident waves with the
// shortest makespan: rounds · (gc + overhead), overhead ≈ 4 rows for a unit's set-up and flushes.  (n = 8192: 68-row units
// = one round over 4096 waves, 88-row units over 3072; the earlier scheme — 64-row units plus a tail of 8-row units —
// lost 12 % at four blocks per CU and 55 % at three, where a quarter of all rows ended up in 8-row units.)
static void syml_geometry(const rc_ctx *c, int cap_blocks, int *gc_out, int *nitems_out)
{
    const int ncb = (c->n + RC_SW_COLS - 1) / RC_SW_COLS, nwaves = 4 * cap_blocks;
    auto rows_of = [&](int J) { return std::min(RC_SW_COLS * J + RC_SW_COLS, c->n); };
    long long best_cost = -1;
    int best_g = RC_SW_ROWS, best_items = 0;
    for (int g = (c->sw_coarse > 0 ? c->sw_coarse : 8); g <= (c->sw_coarse > 0 ? c->sw_coarse : 2 * RC_SW_ROWS); g += 4) {
        long long items = 0;
        for (int J = 0; J < ncb; ++J) items += (rows_of(J) + g - 1) / g;
        const long long rounds = (items + nwaves - 1) / nwaves;
        const long long cost = rounds * (g + 4);
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && g > best_g)) { best_cost = cost; best_g = g; best_items = (int)items; }
    }
    *gc_out = best_g; *nitems_out = best_items;
}

// k_bulk of sweep t on stream B: fills S generation t%3 from perm generation t%2 (labels after sweep t-2),
// clears generation (t+1)%3.  Needs k_resolve(t-2) (perm, and the last reader of the generation being cleared).
static int32_t enqueue_bulk(rc_ctx *c, const View &V, long long t)
{
    const hipStream_t sb = c->res_one_stream ? c->sB2 : ((t & 1) ? c->sB2 : c->sB);
    // resolver(t-2) — the last reader of the generation being filled and the writer of the labels read here.  With the sweep's
    // parity streams it ran on this very stream: in order already, and every event wait is a barrier packet the command
    // processor works through on the critical path (12-19 µs between two kernels of one stream with five of them, ~4 without)
    if (t >= 2 && (c->res_one_stream || c->incremental)) HIPCHK(c, hipStreamWaitEvent(sb, c->ev_res[(t - 2) & 3], 0));
    // While labels move, the resolver of the previous sweep — several rounds, the critical path — runs beside this reduction and the
    // two compete for the CUs: the full-read kernel then takes half as many, longer splits (one block per CU at n = 8192 instead of
    // two), which leaves the resolver the issue slots it needs and still finishes inside its run (moving regime of bench.py:
    // 2,530 -> 3,230 sweeps/s; 1,024-row splits 2,890, 2,048-row splits 2,100: then the reduction is the longer of the two)
    const int rows_split = (c->hsum->n_changes > 32 && !c->bulk_rows_forced) ? std::max(c->rows_per_split, std::min(512, 2 * c->rows_per_split)) : c->rows_per_split;
    const int splits = (c->n + rows_split - 1) / rows_split;
    dim3 gb((unsigned)(c->ld / (c->bits == 64 ? 512 : 1024)), (unsigned)splits);
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    const bool timed = c->timing && (c->timing_every <= 1 || (t % c->timing_every) == 0);
    if (timed) {
        if (!c->ev_free.empty()) { ev = c->ev_free.back(); c->ev_free.pop_back(); }
        else { HIPCHK(c, hipEventCreate(&ev.first)); HIPCHK(c, hipEventCreate(&ev.second)); }
    }
    // A timed launch carries its two events in the dispatch itself (hipExtLaunchKernelGGL: start / stop times of this kernel on
    // this stream): no marker packets before and after the kernel — an hipEventRecord pair is two more packets per launch for the
    // command processor, on the critical path of the sweep (11.6 k instead of 13 k sweeps/s when every launch is timed).
#define RC_BULK_LAUNCH(kf, grid, block, lds, ...)                                                                              \
    do {                                                                                                                       \
        if (timed) hipExtLaunchKernelGGL(kf, dim3(grid), dim3(block), (uint32_t)(lds), sb, ev.first, ev.second, 0u, __VA_ARGS__); \
        else kf<<<grid, block, lds, sb>>>(__VA_ARGS__);                                                                        \
    } while (0)
    const bool use_sym = choose_sym(c);
    c->last_bulk_kernel = use_sym ? 1 : 0;
    const int sym_variant = sym_variant_of(c);
    if (use_sym && c->bits == 32 && sym_variant == 2) {
        int gc = 0, nitems = 0;
        syml_geometry(c, c->symw_per_cu * c->num_cus, &gc, &nitems);
        const int nblocks = std::max(1, std::min((nitems + 3) / 4, c->symw_per_cu * c->num_cus));
        auto kf_ = k_bulk_syml32;
        RC_BULK_LAUNCH(kf_, nblocks, 256, c->syml_pad, V, (int)(t % 3), (int)((t + 1) % 3), (int)(t & 1), (int)(t & 1), nitems, 0, 8, gc);
    } else if (use_sym && c->bits == 64 && sym_variant >= 1) {
        const int ncb = (c->n + RC_SW_COLS - 1) / RC_SW_COLS;
        const int cap_blocks = c->symw_per_cu * c->num_cus;          // resident 4-wave blocks
        auto rows_of = [&](int J) { return std::min(RC_SW_COLS * J + RC_SW_COLS, c->n); };
        if (sym_variant == 3) {
            // While labels move the resolver of the previous sweep runs its rounds with 512-thread blocks (two waves per SIMD at 128
            // registers), which do not fit on a CU beside three of this kernel's blocks: they would wait for the reduction to retire
            // and the two launches run one after the other.  Two blocks per CU (their own unit lists) leave the room: N = 8192, 40
            // label changes per sweep: 3.2 k -> 3.9 k sweeps/s.
            View V2 = V;
            int s2_blocks = c->syml2_blocks, s2_nslow = c->nslow;
            if (c->hsum->n_changes > 32 && c->s2alt.ufast) {
                V2.ufast = c->s2alt.ufast; V2.uslow = c->s2alt.uslow; V2.wfast = c->s2alt.wfast; V2.wslow = c->s2alt.wslow;
                V2.nfast = c->s2alt.nfast; V2.nslow = c->s2alt.nslow;
                s2_blocks = c->s2alt.blocks; s2_nslow = c->s2alt.nslow;
            }
            if (c->derived && c->Dq48) { auto kf_ = k_bulk_syml2<true, true>; RC_BULK_LAUNCH(kf_, s2_blocks, 256, c->syml_pad, V2, (int)(t % 3), (int)(t & 1), (int)(t & 1)); }
            else if (c->derived) { auto kf_ = k_bulk_syml2<true, false>; RC_BULK_LAUNCH(kf_, s2_blocks, 256, c->syml_pad, V2, (int)(t % 3), (int)(t & 1), (int)(t & 1)); }
            else { auto kf_ = k_bulk_syml2<false, false>; RC_BULK_LAUNCH(kf_, s2_blocks, 256, c->syml_pad, V2, (int)(t % 3), (int)(t & 1), (int)(t & 1)); }
            if (s2_nslow > 0) {   // ragged last column block: its units by the round-2 code, behind the main launch
                const int nb = std::min((s2_nslow + 3) / 4, cap_blocks);
                if (c->derived) k_bulk_syml_list<true><<<nb, 256, 0, sb>>>(V2, (int)(t % 3), (int)(t & 1));
                else k_bulk_syml_list<false><<<nb, 256, 0, sb>>>(V2, (int)(t % 3), (int)(t & 1));
            }
        } else if (sym_variant == 2) {
            int gc = 0, nitems = 0;
            const int jsplit = 0, gfine = 8;   // (every column block in gc-row units)
            syml_geometry(c, cap_blocks, &gc, &nitems);
            const int nblocks = std::max(1, std::min((nitems + 3) / 4, cap_blocks));
            if (c->derived) { auto kf_ = k_bulk_syml<true>; RC_BULK_LAUNCH(kf_, nblocks, 256, c->syml_pad, V, (int)(t % 3), (int)((t + 1) % 3), (int)(t & 1), (int)(t & 1), nitems, jsplit, gfine, gc); }
            else { auto kf_ = k_bulk_syml<false>; RC_BULK_LAUNCH(kf_, nblocks, 256, c->syml_pad, V, (int)(t % 3), (int)((t + 1) % 3), (int)(t & 1), (int)(t & 1), nitems, jsplit, gfine, gc); }
        } else {
            int nitems = 0;
            for (int J = 0; J < ncb; ++J) nitems += (rows_of(J) + RC_SW_ROWS - 1) / RC_SW_ROWS;
            const int nblocks = std::max(1, std::min((nitems + 3) / 4, cap_blocks));
            if (c->derived) { auto kf_ = k_bulk_symw<true>; RC_BULK_LAUNCH(kf_, nblocks, 256, 0, V, (int)(t % 3), (int)((t + 1) % 3), (int)(t & 1), (int)(t & 1), nitems); }
            else { auto kf_ = k_bulk_symw<false>; RC_BULK_LAUNCH(kf_, nblocks, 256, 0, V, (int)(t % 3), (int)((t + 1) % 3), (int)(t & 1), (int)(t & 1), nitems); }
        }
    } else if (use_sym) {
        const int TC = (c->bits == 64) ? RC_SYM_TC : RC_SYM32_TC;
        const int TR = (c->bits == 64) ? RC_SYM_TR : c->sym32_tr;                   // rows per tile
        const int item_tiles = c->sym_item_tiles * (RC_SYM_TR / TR);                 // (a work item is the same number of rows either way)
        const int ncb = (c->n + TC - 1) / TC;
        int nitems = 0;
        for (int J = 0; J < ncb; ++J) {
            const int ntile = (std::min(TC * J + TC, c->n) + TR - 1) / TR;
            nitems += (ntile + item_tiles - 1) / item_tiles;
        }
        const int nblocks = std::max(1, std::min(nitems, (c->bits == 64 ? 2 : c->sym32_bpc) * c->num_cus));
        if (c->bits == 64 && c->derived)
            { auto kf_ = k_bulk_sym<true>; RC_BULK_LAUNCH(kf_, nblocks, 256, 0, V, (int)(t % 3), (int)((t + 1) % 3), (int)(t & 1), (int)(t & 1), c->sym_item_tiles, nitems); }
        else if (c->bits == 64)
            { auto kf_ = k_bulk_sym<false>; RC_BULK_LAUNCH(kf_, nblocks, 256, 0, V, (int)(t % 3), (int)((t + 1) % 3), (int)(t & 1), (int)(t & 1), c->sym_item_tiles, nitems); }
        else
            {
                // (the second generation argument is unused by this kernel: diag builds pass their timing-ablation flags through it)
                if (c->sym32_tr == 16) { auto kf_ = k_bulk_sym32<16>; RC_BULK_LAUNCH(kf_, nblocks, 256, 0, V, (int)(t % 3), c->dbg, (int)(t & 1), (int)(t & 1), item_tiles, nitems); }
                else { auto kf_ = k_bulk_sym32<32>; RC_BULK_LAUNCH(kf_, nblocks, 256, 0, V, (int)(t % 3), c->dbg, (int)(t & 1), (int)(t & 1), item_tiles, nitems); }
            }
    } else {
        // bulk_lds: unused dynamic LDS that caps k_bulk at bulk_blocks_per_cu workgroups per CU, which (i) spreads the
        // grid evenly over the CUs and (ii) leaves registers/wave slots on every CU for the concurrent k_resolve
        if (c->bits == 64 && c->derived)
            { auto kf_ = k_bulk<long long, true>; RC_BULK_LAUNCH(kf_, gb, 256, c->bulk_lds, V, rows_split, (int)(t % 3), (int)((t + 1) % 3), (int)(t & 1)); }
        else if (c->bits == 64)
            { auto kf_ = k_bulk<long long>; RC_BULK_LAUNCH(kf_, gb, 256, c->bulk_lds, V, rows_split, (int)(t % 3), (int)((t + 1) % 3), (int)(t & 1)); }
        else
            { auto kf_ = k_bulk<int>; RC_BULK_LAUNCH(kf_, gb, 256, c->bulk_lds, V, rows_split, (int)(t % 3), (int)((t + 1) % 3), (int)(t & 1)); }
    }
#undef RC_BULK_LAUNCH
    if (timed) c->ev_pending.push_back(ev);
    if (t == 0 || c->res_one_stream) HIPCHK(c, hipEventRecord(c->ev_bulk[t & 3], sb));   // (read by ensure_S for sweep 0 and by the one-stream mode)
    c->bulk_enq = t;
    return RC_OK;
}

// Stream roles.  Sweep t lives on ONE stream — sB for even t, sB2 for odd t: its row reduction, then its resolver — so the
// chain that bounds the sweep rate, row reduction(t) -> resolver(t) -> row reduction(t+2), runs in stream order (a kernel
// boundary of ~2 µs) instead of through two cross-stream events of 10-17 µs each; the one cross-stream dependence left,
// resolver(t) after resolver(t-1), is normally satisfied long before it is needed.  Stream A carries everything else
// (log-likelihood block sums, recorded samples, state edits): work enqueued there that reads the state first waits for the
// last resolver, and the next resolver is made to wait for stream A in turn (sA_dirty -> ev_a).
static int32_t order_A_after_sweeps(rc_ctx *c)
{
    if (c->s_res_last && c->s_res_last != c->sA && c->t_next > 0) HIPCHK(c, hipStreamWaitEvent(c->sA, c->ev_res[(c->t_next - 1) & 3], 0));
    c->sA_dirty = true;
    return RC_OK;
}

// Makes the S generation of the CURRENT labels available to work enqueued on stream A; returns its index.
static int32_t ensure_S(rc_ctx *c, int *gen)
{
    {
        int32_t rc0 = order_A_after_sweeps(c);
        if (rc0 != RC_OK) return rc0;
    }
    if (c->incremental && c->bulk_enq >= 0 && c->t_next > 0) {
        *gen = c->inc_gen;
        return RC_OK;
    }
    if (c->t_next == 0) {
        if (c->bulk_enq < 0) {
            View V = make_view(c);
            int32_t rc = enqueue_bulk(c, V, 0);
            if (rc != RC_OK) return rc;
        }
        HIPCHK(c, hipStreamWaitEvent(c->sA, c->ev_bulk[0], 0));
        *gen = 0;
    } else {
        *gen = (int)((c->t_next - 1) % 3);  // corrected in place by k_resolve(t_next-1), already ordered on stream A
    }
    return RC_OK;
}

// k_resolve synchronises its blocks with a grid barrier, so all of them must be resident at once.  One launch always
// fits (G <= number of CUs), but two launches from different contexts of this process could each hold part of the
// chip and wait for the rest for ever (until the bounded spin reports RC_DERR_BARRIER).  Resolver launches on one
// device are therefore chained: each waits for the completion of the previous one, whichever context it came from.

static int32_t launch_resolve(rc_ctx *c, const View &V, const SweepArgs &sa, int res_threads, size_t lds, hipStream_t sx)
{
    if (sx != c->sA && c->sA_dirty) {   // recorded samples, block sums ... enqueued on stream A read the state this sweep changes
        HIPCHK(c, hipEventRecord(c->ev_a, c->sA));
        HIPCHK(c, hipStreamWaitEvent(sx, c->ev_a, 0));
    }
    c->sA_dirty = false;
    if (sa.t >= 1 && c->s_res_last && c->s_res_last != sx) HIPCHK(c, hipStreamWaitEvent(sx, c->ev_res[(sa.t - 1) & 3], 0));
    ResDevice &rd = g_res[c->dev & 63];
    std::lock_guard<std::mutex> lock(rd.m);   // per device: contexts on other GPUs (rc_run_chains' threads) never meet here
    const bool chain = rd.multi;   // several contexts on this device: their resolvers must not overlap (see above)
    if (chain) {
        if (rd.ev) HIPCHK(c, hipStreamWaitEvent(sx, rd.ev, 0));
        else HIPCHK(c, hipEventCreateWithFlags(&rd.ev, hipEventDisableTiming));
    }
    if (c->wide) k_sweep_wide<<<1, 1024, 0, sx>>>(V, sa, c->inc_gen);
    else k_resolve<<<c->G, res_threads, lds, sx>>>(V, sa, c->G);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, RC_ERR_HIP, "k_resolve launch failed: %s", hipGetErrorString(e));
    if (chain) HIPCHK(c, hipEventRecord(rd.ev, sx));
    HIPCHK(c, hipEventRecord(c->ev_res[sa.t & 3], sx));
    c->s_res_last = sx;
    return RC_OK;
}

static int32_t pull_labels(rc_ctx *c, std::vector<int64_t> &labels, std::vector<int64_t> &sizes, int64_t &K);

static int32_t sweep_enqueue(rc_ctx *c, double r, double p, uint64_t seed, uint64_t sweep_index, int after0, int changes0, int rounds0);

extern "C" int32_t rc_gibbs_sweep_async(rc_ctx *c, double r, double p, uint64_t seed, uint64_t sweep_index)
{
    return sweep_enqueue(c, r, p, seed, sweep_index, -1, 0, 0);
}

// after0 / changes0 / rounds0: -1, 0, 0 for a fresh sweep; a sweep resumed after a capacity growth continues behind point
// after0 (recover_capacity)
static int32_t sweep_enqueue(rc_ctx *c, double r, double p, uint64_t seed, uint64_t sweep_index, int after0, int changes0, int rounds0)
{
    if (!c) return fail(c, RC_ERR_ARG, "rc_gibbs_sweep: NULL ctx");
    if (!c->have_params || !c->have_state) return fail(c, RC_ERR_STATE, "rc_gibbs_sweep: rc_set_params and rc_set_state must be called first");
    if (!(r > 0.0) || !(p > 0.0 && p < 1.0)) return fail(c, RC_ERR_ARG, "rc_gibbs_sweep: need r > 0 and 0 < p < 1 (got r=%g p=%g)", r, p);
    HIPCHK(c, hipSetDevice(c->dev));
    int32_t rc;
    // Label movement fragments the internal layout (every move can add two label runs).  Once the symmetric row
    // reduction would be given up for that reason, and a fresh layout would bring it back, re-lay the points out:
    // drain the pipeline and set the same labels again.  Slot numbers change, the partition and its labels do not, and
    // all sums are exact integers, so the chain is bit-identical with or without this step.
    // A wide context whose chain has come down to few clusters narrows again (rc_set_state of the same labels sizes the tables by the
    // state): drain the pipeline, re-install.  The chain is bit-identical with or without this step, as with the re-layout below.
    if (!c->recovering && c->wide && !c->kcap_fixed && c->t_next >= 2 && (long long)c->hsum->K * 4 <= RC_MAX_KCAP) {
        std::vector<int64_t> labels, sizes;
        int64_t K = 0;
        rc = pull_labels(c, labels, sizes, K);
        if (rc != RC_OK) return rc;
        if (K * 4 <= RC_MAX_KCAP) {
            rc = rc_set_state(c, labels.data());
            if (rc != RC_OK) return rc;
        }
    }
    // (up to K = n/64 clusters a fresh layout is far below the n/32 runs the symmetric kernels accept: every 32 sweeps if need be.  Up to
    // K = n/36 it still is — K plus a few runs — but a few dozen moves can undo it, and a re-layout costs what 40 sweeps do (n = 8192:
    // 10 ms): there the interval starts at 128 sweeps and doubles whenever a layout did not last sixteen intervals.  Moving regime of
    // bench.py, K = 212 at n = 8192: 379 runs and the full-read kernel, 4.0-4.2 k sweeps/s, against 220 runs and k_bulk_syml2, 4.4-4.6 k —
    // which of the two a run ended in used to depend on when the last re-layout had happened while K was still below n/64)
    const long long K_now = c->hsum->K;
    const bool relay_near = c->t_next >= 32 && K_now * 64 <= (long long)c->n;
    const bool relay_far = c->t_next >= c->relayout_gap && K_now * 36 <= (long long)c->n;
    if (!c->recovering && !c->incremental && c->relayout && c->bulk_kernel < 0 && (long long)c->hsum->runs * 32 > (long long)c->n && (relay_near || relay_far)) {
        if (!relay_near) c->relayout_gap = c->t_next < 16 * c->relayout_gap ? std::min<long long>(c->relayout_gap * 2, 1ll << 24) : 128;
        std::vector<int64_t> labels, sizes;
        int64_t K = 0;
        rc = pull_labels(c, labels, sizes, K);
        if (rc != RC_OK) return rc;
        rc = rc_set_state(c, labels.data());
        if (rc != RC_OK) return rc;
    }
    if (!c->recovering && c->inflight.size() >= 65536) {   // (a caller that never synchronises: bound the replay log — every entry before a
        // completed sweep is dead.  BEFORE this sweep's view and record are made: the synchronisation may grow the capacity, which
        // re-allocates the slot buffers and replays the log)
        int32_t rcq = sync_and_check(c);
        if (rcq != RC_OK) return rcq;
    }
    View V = make_view(c);
    const long long t = c->t_next;
    if (t >= 0x7ffffff0ll) return fail(c, RC_ERR_STATE, "rc_gibbs_sweep: internal sweep counter exhausted; call rc_set_state");
    SweepArgs sa;
    sa.r = r;
    sa.logp = std::log(p);
    sa.log1mp = std::log(1 - p);
    sa.k0 = (unsigned)seed; sa.k1 = (unsigned)(seed >> 32);
    sa.sw_lo = (unsigned)sweep_index; sa.sw_hi = (unsigned)(sweep_index >> 32);
    sa.t = (int)t;
    sa.dbg = c->dbg;
    sa.after0 = after0; sa.changes0 = changes0; sa.rounds0 = rounds0;
    // Pruning pays where most candidates are far from the point's own cluster — the stationary regime (N = 8192, K = 50: 15.7 k -> 16.4 k
    // sweeps/s) — and costs a few per cent where a chain moves among many small clusters (every stream evaluates the own cluster first)
    sa.prune = (c->opt_prune != 0 && (c->opt_prune > 0 || c->hsum->n_changes <= 2)) ? 1 : 0;   // (rc_set_option "prune"; sigma = 0.18, 9 changes per sweep among 117 clusters: 9.8 k -> 9.2 k with it on)
    c->inflight.push_back(rc_ctx::SweepRec{r, p, seed, sweep_index, t});
    // Resolver block size.  With 256 threads (one wave per SIMD, 112 VGPRs) a k_resolve block fits on a CU beside two
    // k_bulk_sym blocks, so the resolver of sweep t really overlaps the row reduction of sweep t+1 (config 5:
    // 1.36 -> 0.99 ms per sweep).  With many label changes per sweep the rounds dominate and 512 threads are faster.
    int res_threads = c->res_threads;
    if (res_threads == 0)
        // (wide blocks pay with many candidates per point — 32 streams of K = 200 candidates; with a few dozen clusters the extra waves only wait)
        res_threads = c->incremental ? ((c->hsum->K > RC_RES_WIDE_MIN_K && tab_bytes(c->kcap, c->n, RC_RES_THREADS_INC / 64, c->maxb) <= 160 * 1024) ? RC_RES_THREADS_INC : RC_RES_THREADS)
                                     : ((c->prefetch && c->last_bulk_kernel == 1 && c->hsum->n_changes <= 32) ? 256 : RC_RES_THREADS);
    // (the tables' layout depends on the waves per block — the reduction scratch of eval_chunk — and k_resolve carves it by its blockDim)
#ifdef RC_TRACE_RESOLVE
    const size_t lds = std::max(tab_bytes(c->kcap, c->n, res_threads / 64, c->maxb), 2 * sizeof(int) * (size_t)c->kcap) + 4096;
#else
    const size_t lds = c->wide ? 0 : std::max(tab_bytes(c->kcap, c->n, res_threads / 64, c->maxb), 2 * sizeof(int) * (size_t)c->kcap);
#endif
    if (c->incremental) {
        // exact incremental mode: the row-sum table of the current labels already exists (one k_bulk after
        // rc_set_state) and every label change corrects it in place — no matrix traffic at all in this sweep
        int gen = 0;
        if (t == 0 || c->bulk_enq < 0) {
            rc = ensure_S(c, &gen);
            if (rc != RC_OK) return rc;
            c->inc_gen = gen;
        }
        sa.own_gen = c->inc_gen;
        sa.next_gen = -1;
        sa.zero_gen = -1;
        rc = launch_resolve(c, V, sa, res_threads, lds, c->sA);
        if (rc != RC_OK) return rc;
        c->t_next = t + 1;
        c->state_version++;
        return RC_OK;
    }
    sa.own_gen = (int)(t % 3);
    sa.next_gen = (int)((t + 1) % 3);
    sa.zero_gen = (int)((t + 2) % 3);
    if (c->bulk_enq < t) {
        rc = enqueue_bulk(c, V, t);
        if (rc != RC_OK) return rc;
    }
    // the stream of sweep t: its row reduction is already on it — except for small problems, where the chain that bounds the
    // sweep rate is resolver(t) -> resolver(t+1) (a cross-stream event costs more than the resolver itself): all resolvers
    // then run in order on stream B and the (short) row reductions on B2, each resolver waiting for its reduction's event
    const hipStream_t sx = c->res_one_stream ? c->sB : ((t & 1) ? c->sB2 : c->sB);
    if (c->res_one_stream) HIPCHK(c, hipStreamWaitEvent(sx, c->ev_bulk[t & 3], 0));
    rc = launch_resolve(c, V, sa, res_threads, lds, sx);
    if (rc != RC_OK) return rc;
    c->t_next = t + 1;
    c->state_version++;
    if (c->prefetch) {
        // software pipeline: the row reduction of the next sweep starts now, under the labels known before
        // this sweep; k_resolve adds this sweep's label changes to it (exact integer atomics)
        rc = enqueue_bulk(c, V, t + 1);
        if (rc != RC_OK) return rc;
    }
    return RC_OK;
}

extern "C" int32_t rc_gibbs_sweep(rc_ctx *c, double r, double p, uint64_t seed, uint64_t sweep_index)
{
    int32_t rc = rc_gibbs_sweep_async(c, r, p, seed, sweep_index);
    if (rc != RC_OK) return rc;
    return sync_and_check(c);
}

// A sweep stopped at the first point that needed a new cluster when every slot was taken (k_resolve: RC_DERR_CAPACITY).  The
// device state is consistent — every point up to hsum->resume_after is final, the others still carry their old labels — and the
// sweeps enqueued behind it returned at once without touching anything.  The reference's state has room for n clusters
// (clustsizes of length n, types.jl:131-137; a new cluster is offered whenever maxK allows, mcmc.jl:198-199), so the sweep must
// simply go on: double the slot capacity, install the current labels again (rc_set_state: a fresh layout and a fresh row-sum
// table), resume the sweep behind that point and replay the later ones.  Every draw is a pure function of (state, sweep index,
// point, label), so the chain is exactly the one a larger initial capacity would have produced.
static int32_t recover_capacity(rc_ctx *c)
{
    HIPCHK(c, hipStreamSynchronize(c->sA));
    HIPCHK(c, hipStreamSynchronize(c->sB));
    HIPCHK(c, hipStreamSynchronize(c->sB2));
    {
        int32_t rc = drain_events(c);
        if (rc != RC_OK) return rc;
    }
    const int fail_t = c->hsum->fail_t, resume_after = c->hsum->resume_after, ch0 = c->hsum->fail_changes, rd0 = c->hsum->fail_rounds;
    std::deque<rc_ctx::SweepRec> log;
    log.swap(c->inflight);
    while (!log.empty() && log.front().t < fail_t) log.pop_front();
    if (c->kcap_fixed || c->kcap >= c->kcap_max || log.empty() || log.front().t != fail_t) {
        c->have_state = c->have_state && !log.empty();
        return fail(c, RC_ERR_CAPACITY, c->kcap >= c->kcap_max && !c->kcap_fixed
                        ? "number of clusters exceeded %d = min(n, %d), the most the library holds (slot ids are 16-bit)"
                        : "number of clusters exceeded the slot capacity kcap=%d (fixed: RC_KCAP_FIXED)", c->kcap, RC_WIDE_MAX_KCAP);
    }
    // the labels as they stand (straight from the device: the summary's tables are those of the failed launch as well, but the
    // sweep's error bit would turn pull_labels -> sync_and_check back into this function)
    std::vector<int> so((size_t)c->n), slabel((size_t)c->kcap);
    HIPCHK(c, hipMemcpy(so.data(), c->slot_of, so.size() * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(slabel.data(), c->slot_label, slabel.size() * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<int64_t> labels((size_t)c->n);
    for (int i = 0; i < c->n; ++i) labels[(size_t)i] = slabel[(size_t)so[(size_t)c->h_pi[(size_t)i]]];
    c->recovering = true;
    struct Guard { rc_ctx *c; ~Guard() { c->recovering = false; } } guard{c};
    int32_t rc = resize_capacity(c, (int)std::min<long long>(c->kcap_max, std::max<long long>(2ll * c->kcap, 128)));
    if (rc != RC_OK) return rc;
    rc = rc_set_state(c, labels.data());      // clears the device error word (DevScalars is rewritten) and the summary's
    if (rc != RC_OK) return rc;
    c->last.err = 0;
    bool first = true;
    for (const rc_ctx::SweepRec &q : log) {
        rc = first ? sweep_enqueue(c, q.r, q.p, q.seed, q.sweep_index, resume_after, ch0, rd0)
                   : sweep_enqueue(c, q.r, q.p, q.seed, q.sweep_index, -1, 0, 0);
        if (rc != RC_OK) return rc;
        first = false;
    }
    return RC_OK;
}

extern "C" int32_t rc_capacity_info(rc_ctx *c, int64_t *kcap, int64_t *kcap_max, int64_t *n_grows, int64_t *batch_capacity)
{
    if (!c) return fail(c, RC_ERR_ARG, "rc_capacity_info: NULL ctx");
    if (kcap) *kcap = c->kcap;
    if (kcap_max) *kcap_max = c->kcap_max;
    if (n_grows) *n_grows = c->n_grows;
    if (batch_capacity) *batch_capacity = c->maxb;
    return RC_OK;
}

extern "C" int32_t rc_last_sweep_stats(rc_ctx *c, rc_sweep_stats *out)
{
    if (!c || !out) return fail(c, RC_ERR_ARG, "rc_last_sweep_stats: NULL argument");
    HIPCHK(c, hipSetDevice(c->dev));
    int32_t rc = sync_and_check(c);
    if (rc != RC_OK) return rc;
    out->n_changes = c->last.n_changes;
    out->n_rounds = c->last.n_rounds;
    out->K = c->last.K;
    return RC_OK;
}

// labels / sizes of the current device state
static int32_t pull_state(rc_ctx *c, std::vector<int> &so, std::vector<int> &ssize, std::vector<int> &slabel,
                          bool want_points = true)
{
    int32_t rc = sync_and_check(c);
    if (rc != RC_OK) return rc;
    ssize.resize((size_t)c->kcap); slabel.resize((size_t)c->kcap);
    for (int k = 0; k < c->kcap; ++k) {  // host-mapped summary written by the last kernel: no device copy
        ssize[(size_t)k] = c->hsum->size_label[2 * k];
        slabel[(size_t)k] = c->hsum->size_label[2 * k + 1];
    }
    if (want_points) {
        so.resize((size_t)c->n);
        HIPCHK(c, hipMemcpyAsync(so.data(), c->slot_of, so.size() * sizeof(int), hipMemcpyDeviceToHost, c->sA));
        HIPCHK(c, hipStreamSynchronize(c->sA));
    }
    return RC_OK;
}
extern "C" int32_t rc_get_state(rc_ctx *c, int64_t *clusts, int64_t *clustsizes, int64_t *K)
{
    if (!c) return fail(c, RC_ERR_ARG, "rc_get_state: NULL ctx");
    if (!c->have_state) return fail(c, RC_ERR_STATE, "rc_get_state: no state set");
    HIPCHK(c, hipSetDevice(c->dev));
    std::vector<int> so, ssize, slabel;
    int32_t rc = pull_state(c, so, ssize, slabel, clusts != nullptr);
    if (rc != RC_OK) return rc;
    if (clusts)
        for (int i = 0; i < c->n; ++i) clusts[i] = slabel[(size_t)so[(size_t)c->h_pi[(size_t)i]]];
    if (clustsizes) {
        std::memset(clustsizes, 0, (size_t)c->n * sizeof(int64_t));
        for (int k = 0; k < c->kcap; ++k)
            if (slabel[(size_t)k] > 0) clustsizes[slabel[(size_t)k] - 1] = ssize[(size_t)k];
    }
    if (K) *K = c->last.K;
    return RC_OK;
}

// Scalar part of loglik (mcmc.jl:26-54) in long double, regrouped as in the oracle's "stable" mode, from the exact
// fixed-point block sums B[t][k] (hi × hi × 4: D hi/lo, logD hi/lo) and the slot sizes.  A term depends only on its
// two cluster sizes and its block sums, and most of them do not change from one recorded sample to the next, so the
// terms are cached per slot pair and re-evaluated only when their (integer) inputs differ: the value — and the order
// of the summation — is exactly that of evaluating every term afresh.
// The terms are summed in ascending LABEL order (slabel: the label of every slot), the order of the reference's loops over
// findall(clustsizes .> 0) (mcmc.jl:13-53) — not in slot order: slot numbers depend on the history of births and deaths (and
// differ between the speculative and the synchronous chain loop after a rollback), labels do not, so the value is a function of
// the partition alone, bit for bit.
static double loglik_host_c(const rc_ctx *c, rc_ctx::LLCache &cache, int hi, const int *ssize, const long long *B, const int *slabel)
{
    const rc_params &P = c->P;
    const long double d1 = P.delta1, d2 = P.delta2, al = P.alpha, be = P.beta, ze = P.zeta, ga = P.gamma;
    const long double lga = lgammal(al), lgz = lgammal(ze), lgd1 = lgammal(d1), lgd2 = lgammal(d2);
    const long double lb = logl(be), lg = logl(ga);
    const long double scD = ldexpl(1.0L, -c->eD), scL = ldexpl(1.0L, -c->eL);
    // (the per-pair memo is 64 B per slot pair: kept up to the fast path's capacity; a wide context evaluates every term afresh)
    const bool memo = hi <= RC_MAX_KCAP;
    if (memo && hi > cache.ll_dim) {
        cache.ll_dim = std::max(hi, std::min(std::min(c->kcap, RC_MAX_KCAP), 2 * hi));
        cache.ll_cache.assign((size_t)cache.ll_dim * cache.ll_dim, rc_ctx::LLTerm{});
    }
    rc_ctx::LLTerm scratch_term;
    auto blk = [&](const long long *e, int which) -> long double {
        return ((long double)e[which ? 2 : 0] * (long double)(1ll << RC_LO_BITS) + (long double)e[which ? 3 : 1]) * (which ? scL : scD);
    };
    std::vector<int> act;
    for (int k = 0; k < hi; ++k)
        if (ssize[k] > 0) act.push_back(k);
    std::sort(act.begin(), act.end(), [&](int x, int y) { return slabel[x] < slabel[y]; });
    long double L1 = 0, L2 = 0;
    for (int k : act) {
        const long long *e = &B[((size_t)k * hi + k) * 4];
        if (!memo) scratch_term = rc_ctx::LLTerm{};
        rc_ctx::LLTerm &T = memo ? cache.ll_cache[(size_t)k * cache.ll_dim + k] : scratch_term;
        if (!(T.sk == ssize[k] && T.e[0] == e[0] && T.e[1] == e[1] && T.e[2] == e[2] && T.e[3] == e[3])) {
            const long double sz = ssize[k];
            const long double pairs = sz * (sz - 1) / 2;  // binomial(sz_k, 2)
            const long double a = al + d1 * pairs;
            const long double bd = blk(e, 0) / 2, bl = blk(e, 1) / 2;
            const long long pk = (long long)pairs;
            auto it = cache.lg_memo1.find(pk);
            if (it == cache.lg_memo1.end()) it = cache.lg_memo1.emplace(pk, lgammal(a) - lga).first;
            T.term = (d1 - 1) * bl - pairs * lgd1 + it->second - d1 * pairs * lb - a * log1pl(bd / be);
            T.sk = ssize[k]; T.st = ssize[k];
            std::memcpy(T.e, e, sizeof(T.e));
        }
        L1 += T.term;
    }
    if (cache.lg_memo1.size() > (1u << 20)) cache.lg_memo1.clear();
    if (cache.lg_memo2.size() > (1u << 20)) cache.lg_memo2.clear();
    if (P.repulsion)
        for (size_t x = 0; x < act.size(); ++x)
            for (size_t y = x + 1; y < act.size(); ++y) {
                const int k = act[x], t = act[y];
                const long long *e = &B[((size_t)t * hi + k) * 4];
                if (!memo) scratch_term = rc_ctx::LLTerm{};
                rc_ctx::LLTerm &T = memo ? cache.ll_cache[(size_t)t * cache.ll_dim + k] : scratch_term;
                if (!(T.sk == ssize[k] && T.st == ssize[t] && T.e[0] == e[0] && T.e[1] == e[1] && T.e[2] == e[2] && T.e[3] == e[3])) {
                    const long double pairs = (long double)ssize[k] * (long double)ssize[t];
                    const long double z = ze + d2 * pairs;
                    const long double bd = blk(e, 0), bl = blk(e, 1);
                    const long long pk = (long long)pairs;
                    auto it = cache.lg_memo2.find(pk);
                    if (it == cache.lg_memo2.end()) it = cache.lg_memo2.emplace(pk, lgammal(z) - lgz).first;
                    T.term = (d2 - 1) * bl - pairs * lgd2 + it->second - d2 * pairs * lg - z * log1pl(bd / ga);
                    T.sk = ssize[k]; T.st = ssize[t];
                    std::memcpy(T.e, e, sizeof(T.e));
                }
                L2 += T.term;
            }
    return (double)(L1 + L2);
}

static double loglik_host(rc_ctx *c, int hi, const int *ssize, const long long *B, const int *slabel) { return loglik_host_c(c, c->llc, hi, ssize, B, slabel); }

// enqueues the block sums of the current state on stream A and their copy into the pinned buffer `dst`
// (hi·hi·4 int64); the caller synchronises (stream or event) before reading
// copy_stream != nullptr: the copy to the host goes to that stream behind the kernel (the caller records its own completion event
// there and sets c->ev_blocks_busy to it); stream A — the sweeps' stream in incremental mode — carries only the kernel.
static int32_t loglik_enqueue(rc_ctx *c, int hi, long long *dst, hipStream_t copy_stream = nullptr)
{
    int gen = 0;
    int32_t rc = ensure_S(c, &gen);
    if (rc != RC_OK) return rc;
    View V = make_view(c);
    if (c->ev_blocks_busy) {   // c->blocks is one buffer: the last copy out of it must be complete — it nearly always is by now, and a
        // wait packet on stream A is a cross-stream hop on the path between two sweeps (10-20 µs) even when the event has fired
        if (hipEventQuery(c->ev_blocks_busy) != hipSuccess) HIPCHK(c, hipStreamWaitEvent(c->sA, c->ev_blocks_busy, 0));
        (void)hipGetLastError();   // (hipErrorNotReady is not an error)
        c->ev_blocks_busy = nullptr;
    }
    if ((size_t)hi > c->blocks_cap) {   // (wide contexts: the buffer follows the clusters in use)
        HIPCHK(c, hipStreamSynchronize(c->sA));
        if (c->blocks) (void)hipFree(c->blocks);
        c->blocks = nullptr; c->blocks_cap = 0;
        const size_t cap = std::min<size_t>((size_t)c->kcap, (size_t)hi + (size_t)hi / 4 + 64);
        HIPCHK(c, hipMalloc(&c->blocks, cap * cap * 4 * sizeof(long long)));
        c->blocks_cap = cap;
    }
    k_blocksums<<<dim3((unsigned)hi, (unsigned)((hi + RC_BS_TILE - 1) / RC_BS_TILE)), 256, (size_t)std::min(hi, RC_BS_TILE) * 4 * sizeof(u64), c->sA>>>(V, gen, hi, c->blocks);
    HIPCHK(c, hipGetLastError());
    if (copy_stream) {
        HIPCHK(c, hipEventRecord(c->ev_k, c->sA));
        HIPCHK(c, hipStreamWaitEvent(copy_stream, c->ev_k, 0));
    }
    HIPCHK(c, hipMemcpyAsync(dst, c->blocks, (size_t)hi * hi * 4 * sizeof(long long), hipMemcpyDeviceToHost, copy_stream ? copy_stream : c->sA));
    return RC_OK;
}

// pinned staging for block sums / label snapshots of sample slot q (0..RC_REC_SLOTS-1), sized for `hi` slots.
// Grows by reallocation: call only while no asynchronous copy into the buffers is outstanding.
static int32_t ensure_pinned(rc_ctx *c, int hi)
{
    if (c->pinB[0] && hi <= c->pin_hi) return RC_OK;
    // (beyond the fast path's capacities the staging follows the clusters in use closely: 2·hi slots would be 4x the bytes, and they are GBs there)
    const int cap = hi > RC_MAX_KCAP / 2 ? std::min(c->kcap, hi + hi / 8 + 64) : std::min(c->kcap, std::max(64, 2 * hi));
    HIPCHK(c, hipStreamSynchronize(c->sA));
    for (int q = 0; q < RC_REC_SLOTS; ++q) {
        if (c->pinB[q]) (void)hipHostFree(c->pinB[q]);
        c->pinB[q] = nullptr;
        HIPCHK(c, hipHostMalloc((void **)&c->pinB[q], (size_t)cap * cap * 4 * sizeof(long long), hipHostMallocDefault));
        if (!c->pinLab[q]) {
            HIPCHK(c, hipHostMalloc((void **)&c->pinLab[q], (size_t)(c->n + 8) * sizeof(unsigned short), hipHostMallocDefault));
            HIPCHK(c, hipEventCreateWithFlags(&c->pinEv[q], hipEventDisableTiming));
        }
    }
    c->pin_hi = cap;
    return RC_OK;
}

extern "C" int32_t rc_loglik(rc_ctx *c, double *out)
{
    if (!c || !out) return fail(c, RC_ERR_ARG, "rc_loglik: NULL argument");
    if (!c->have_params || !c->have_state) return fail(c, RC_ERR_STATE, "rc_loglik: params and state must be set");
    HIPCHK(c, hipSetDevice(c->dev));
    std::vector<int> so, ssize, slabel;
    int32_t rc = pull_state(c, so, ssize, slabel, false);  // also refreshes c->last (slot_hi)
    if (rc != RC_OK) return rc;
    const int hi = std::max(1, std::min(c->kcap, c->last.slot_hi));
    rc = ensure_pinned(c, hi);
    if (rc != RC_OK) return rc;
    rc = loglik_enqueue(c, hi, c->pinB[0]);
    if (rc != RC_OK) return rc;
    HIPCHK(c, hipStreamSynchronize(c->sA));
    *out = loglik_host(c, hi, ssize.data(), c->pinB[0], slabel.data());
    // keep the block sums: a merge proposal's log-likelihood follows from them without touching the device
    c->B_cur.assign(c->pinB[0], c->pinB[0] + (size_t)hi * hi * 4);
    c->B_ssize = ssize; c->B_slabel = slabel;
    c->B_hi = hi;
    c->B_version = c->state_version;
    return RC_OK;
}

// The within- / between-cluster split of the upper triangle that fitprior feeds to its Gamma fits
// (src/prior.jl:73-75: A = distances of pairs in the same cluster, B = the others; :96-110 use |A|, ΣA, Σlog A and the
// same for B) for the CURRENT labels, from the block sums: Σ_A = ½ Σ_k B(k,k), Σ_B = Σ_{k<t} B(k,t).  Exact integer
// accumulation; the only rounding is the fixed-point quantisation of the entries.
extern "C" int32_t rc_within_between(rc_ctx *c, rc_wb_stats *out)
{
    if (!c || !out) return fail(c, RC_ERR_ARG, "rc_within_between: NULL argument");
    if (!c->have_state) return fail(c, RC_ERR_STATE, "rc_within_between: no state set");
    HIPCHK(c, hipSetDevice(c->dev));
    std::vector<int> so, ssize, slabel;
    int32_t rc = pull_state(c, so, ssize, slabel, false);
    if (rc != RC_OK) return rc;
    const int hi = std::max(1, std::min(c->kcap, c->last.slot_hi));
    rc = ensure_pinned(c, hi);
    if (rc != RC_OK) return rc;
    rc = loglik_enqueue(c, hi, c->pinB[0]);
    if (rc != RC_OK) return rc;
    HIPCHK(c, hipStreamSynchronize(c->sA));
    const long long *B = c->pinB[0];
    __int128 wD = 0, wL = 0, bD = 0, bL = 0;
    long long cntA = 0, cntB = 0;
    auto val = [&](int t, int k, int which) -> __int128 {
        const long long *e = &B[((size_t)t * hi + k) * 4 + (which ? 2 : 0)];
        return (__int128)e[0] * ((__int128)1 << RC_LO_BITS) + (__int128)e[1];
    };
    for (int k = 0; k < hi; ++k) {
        if (ssize[(size_t)k] <= 0) continue;
        const long long sk = ssize[(size_t)k];
        cntA += sk * (sk - 1) / 2;
        wD += val(k, k, 0); wL += val(k, k, 1);          // every within pair twice, the (zero) diagonal once
        for (int t = k + 1; t < hi; ++t) {
            if (ssize[(size_t)t] <= 0) continue;
            cntB += sk * (long long)ssize[(size_t)t];
            bD += val(t, k, 0); bL += val(t, k, 1);
        }
    }
    // the diagonal of D is kept as stored (types.jl:155 zeroes only logD's): remove it from the within sum
    std::vector<long long> dg((size_t)c->n);
    HIPCHK(c, hipMemcpy(dg.data(), c->diagq, (size_t)c->n * sizeof(long long), hipMemcpyDeviceToHost));
    __int128 dsum = 0;
    for (long long v : dg) dsum += v;
    const long double scD = ldexpl(1.0L, -c->eD), scL = ldexpl(1.0L, -c->eL);
    out->count_within = cntA; out->count_between = cntB;
    out->sum_within = (double)((long double)(wD - dsum) * scD / 2);
    out->sumlog_within = (double)((long double)wL * scL / 2);
    out->sum_between = (double)((long double)bD * scD);
    out->sumlog_between = (double)((long double)bL * scL);
    return RC_OK;
}

// logprior (mcmc.jl:58-78) from slot sizes / labels
// (nslots: length of ssize / slabel — the capacity at the time the tables were taken, which the context's may have outgrown since)
static double logprior_host(const rc_ctx *c, const int *ssize, const int *slabel, double r, double p, int nslots = -1)
{
    const rc_params &P = c->P;
    const double n = c->n;
    if (nslots < 0) nslots = c->kcap;
    double K = 0;
    for (int k = 0; k < nslots; ++k) K += ssize[k] > 0;
    // logpdf(Gamma(η, 1/σ), r) + logpdf(Beta(u, v), p)   (mcmc.jl:73)
    const double lgam = P.eta * std::log(P.sigma) - std::lgamma(P.eta) + (P.eta - 1) * std::log(r) - P.sigma * r;
    const double lbet = std::lgamma(P.u + P.v) - std::lgamma(P.u) - std::lgamma(P.v) + (P.u - 1) * std::log(p) + (P.v - 1) * std::log(1 - p);
    double L = std::lgamma(K + 1) + (n - K) * std::log(p) + (r * K) * std::log(1 - p) - K * std::lgamma(r) + lgam + lbet;
    // Σ_j log n_j + lgΓ(n_j + r − 1) over non-empty clusters in ascending label order (mcmc.jl:74-76)
    std::vector<std::pair<int, int>> bylabel;
    for (int k = 0; k < nslots; ++k)
        if (ssize[k] > 0) bylabel.push_back({slabel[k], ssize[k]});
    std::sort(bylabel.begin(), bylabel.end());
    for (auto &e : bylabel) L += std::log((double)e.second) + std::lgamma((double)e.second + r - 1);
    return L;
}

extern "C" int32_t rc_logprior(rc_ctx *c, double r, double p, double *out)
{
    if (!c || !out) return fail(c, RC_ERR_ARG, "rc_logprior: NULL argument");
    if (!c->have_params || !c->have_state) return fail(c, RC_ERR_STATE, "rc_logprior: params and state must be set");
    if (!(r > 0.0) || !(p > 0.0 && p < 1.0)) return fail(c, RC_ERR_ARG, "rc_logprior: need r > 0 and 0 < p < 1");
    HIPCHK(c, hipSetDevice(c->dev));
    std::vector<int> so, ssize, slabel;
    int32_t rc = pull_state(c, so, ssize, slabel, false);
    if (rc != RC_OK) return rc;
    *out = logprior_host(c, ssize.data(), slabel.data(), r, p);
    return RC_OK;
}

static int32_t ensure_counts(rc_ctx *c)
{
    if (c->counts) return RC_OK;
    c->ldc = ((c->n + 3) / 4) * 4;
    HIPCHK(c, hipMalloc(&c->counts, (size_t)c->n * c->ldc * sizeof(unsigned)));
    HIPCHK(c, hipMalloc(&c->snap, (size_t)RC_CC_BATCH * c->ldc * sizeof(unsigned short)));
    HIPCHK(c, hipMemsetAsync(c->counts, 0, (size_t)c->n * c->ldc * sizeof(unsigned), c->sA));
    c->snap_cnt = 0;
    return RC_OK;
}

// adds the queued label snapshots to the count matrix (one pass over the matrix for up to RC_CC_BATCH samples)
static int32_t flush_counts(rc_ctx *c)
{
    if (!c->counts || c->snap_cnt == 0) return RC_OK;
    dim3 g((unsigned)((c->ldc + 1023) / 1024), (unsigned)((c->n + RC_CC_ROWS - 1) / RC_CC_ROWS));
    k_cocluster_batch<<<g, 256, 0, c->sA>>>(c->snap, c->snap_cnt, c->n, c->ldc, c->ldc, c->counts);
    HIPCHK(c, hipGetLastError());
    c->snap_cnt = 0;
    return RC_OK;
}

extern "C" int32_t rc_record_sample(rc_ctx *c, int64_t *canonical_out)
{
    if (!c) return fail(c, RC_ERR_ARG, "rc_record_sample: NULL ctx");
    if (!c->have_state) return fail(c, RC_ERR_STATE, "rc_record_sample: no state set");
    HIPCHK(c, hipSetDevice(c->dev));
    int32_t rc = ensure_counts(c);
    if (rc != RC_OK) return rc;
    // Sweeps still in flight may have run out of slots: the capacity is grown and the sweeps resumed / replayed by sync_and_check,
    // and a snapshot enqueued before that would hold the half-swept state (and its slot ids would be mapped through the layout of
    // the re-installed state).  So the reader waits first.  (rc_run_chain synchronises before every snapshot of its own.)
    if (!c->inflight.empty()) {
        rc = sync_and_check(c);
        if (rc != RC_OK) return rc;
    }
    rc = order_A_after_sweeps(c);
    if (rc != RC_OK) return rc;
    k_snapshot<<<(c->ldc + 255) / 256, 256, 0, c->sA>>>(c->slot_of, c->pi, c->n, c->ldc, c->snap + (size_t)c->snap_cnt * c->ldc);
    HIPCHK(c, hipGetLastError());
    if (++c->snap_cnt == RC_CC_BATCH) {
        rc = flush_counts(c);
        if (rc != RC_OK) return rc;
    }
    if (canonical_out) {
        std::vector<int> so((size_t)c->n);
        HIPCHK(c, hipMemcpyAsync(so.data(), c->slot_of, so.size() * sizeof(int), hipMemcpyDeviceToHost, c->sA));
        rc = sync_and_check(c);
        if (rc != RC_OK) return rc;
        // sortlabels (utils.jl:69-74): relabel by order of first appearance
        std::vector<int> map((size_t)c->kcap, 0);
        int next = 0;
        for (int i = 0; i < c->n; ++i) {
            int &m = map[(size_t)so[(size_t)c->h_pi[(size_t)i]]];
            if (m == 0) m = ++next;
            canonical_out[i] = m;
        }
    }
    return RC_OK;
}

extern "C" int32_t rc_cocluster_reset(rc_ctx *c)
{
    if (!c) return fail(c, RC_ERR_ARG, "rc_cocluster_reset: NULL ctx");
    HIPCHK(c, hipSetDevice(c->dev));
    int32_t rc = ensure_counts(c);
    if (rc != RC_OK) return rc;
    c->snap_cnt = 0;
    HIPCHK(c, hipMemsetAsync(c->counts, 0, (size_t)c->n * c->ldc * sizeof(unsigned), c->sA));
    return RC_OK;
}

extern "C" int32_t rc_cocluster_counts(rc_ctx *c, uint32_t *out)
{
    if (!c || !out) return fail(c, RC_ERR_ARG, "rc_cocluster_counts: NULL argument");
    HIPCHK(c, hipSetDevice(c->dev));
    int32_t rc = ensure_counts(c);
    if (rc != RC_OK) return rc;
    rc = flush_counts(c);
    if (rc != RC_OK) return rc;
    rc = sync_and_check(c);
    if (rc != RC_OK) return rc;
    HIPCHK(c, hipMemcpy2D(out, (size_t)c->n * sizeof(unsigned), c->counts, (size_t)c->ldc * sizeof(unsigned),
                          (size_t)c->n * sizeof(unsigned), (size_t)c->n, hipMemcpyDeviceToHost));
    return RC_OK;
}

extern "C" int32_t rc_cocluster_device_buffer(rc_ctx *c, void **dev_ptr, int64_t *ld)
{
    if (!c || !dev_ptr || !ld) return fail(c, RC_ERR_ARG, "rc_cocluster_device_buffer: NULL argument");
    HIPCHK(c, hipSetDevice(c->dev));
    int32_t rc = ensure_counts(c);
    if (rc != RC_OK) return rc;
    rc = flush_counts(c);
    if (rc != RC_OK) return rc;
    rc = sync_and_check(c);
    if (rc != RC_OK) return rc;
    *dev_ptr = c->counts;
    *ld = c->ldc;
    return RC_OK;
}

extern "C" int32_t rc_cocluster(rc_ctx *c, double *out, int64_t numsamples)
{
    if (!c || !out) return fail(c, RC_ERR_ARG, "rc_cocluster: NULL argument");
    if (numsamples < 1) return fail(c, RC_ERR_ARG, "rc_cocluster: numsamples must be >= 1");
    HIPCHK(c, hipSetDevice(c->dev));
    int32_t rc = ensure_counts(c);
    if (rc != RC_OK) return rc;
    rc = flush_counts(c);
    if (rc != RC_OK) return rc;
    const size_t nn = (size_t)c->n * c->n;
    if (!c->cc_out) HIPCHK(c, hipMalloc(&c->cc_out, nn * sizeof(double)));
    const int gb = (int)std::min<size_t>((nn + 255) / 256, 8192);
    k_cocluster_final<<<gb, 256, 0, c->sA>>>(c->counts, c->n, c->ldc, (double)numsamples, c->cc_out);
    rc = sync_and_check(c);
    if (rc != RC_OK) return rc;
    HIPCHK(c, hipMemcpy(out, c->cc_out, nn * sizeof(double), hipMemcpyDeviceToHost));
    return RC_OK;
}

extern "C" int32_t rc_debug_rowsums(rc_ctx *c, int64_t label, int64_t *sumD_q, int64_t *sumL_q, int32_t *eD, int32_t *eL)
{
    if (!c) return fail(c, RC_ERR_ARG, "rc_debug_rowsums: NULL ctx");
    if (!c->have_state) return fail(c, RC_ERR_STATE, "rc_debug_rowsums: no state set");
    HIPCHK(c, hipSetDevice(c->dev));
    int gen = 0;
    int32_t rc = sync_and_check(c);        // (a sweep in flight may grow the capacity: buffers and generations change under it)
    if (rc != RC_OK) return rc;
    rc = ensure_S(c, &gen);
    if (rc != RC_OK) return rc;
    std::vector<int> so, ssize, slabel;
    rc = pull_state(c, so, ssize, slabel);
    if (rc != RC_OK) return rc;
    int slot = -1;
    for (int k = 0; k < c->kcap; ++k)
        if (slabel[(size_t)k] == (int)label) slot = k;
    if (eD) *eD = c->eD;
    if (eL) *eL = c->eL;
    if (slot < 0) {
        if (sumD_q) std::memset(sumD_q, 0, (size_t)c->n * 8);
        if (sumL_q) std::memset(sumL_q, 0, (size_t)c->n * 8);
        return RC_OK;
    }
    std::vector<int64_t> tmp((size_t)c->n);
    if (sumD_q) {
        HIPCHK(c, hipMemcpy(tmp.data(), c->SD[gen] + (size_t)slot * c->ld, (size_t)c->n * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < c->n; ++i) sumD_q[i] = tmp[(size_t)c->h_pi[(size_t)i]];  // back to the caller's point order
    }
    if (sumL_q) {
        HIPCHK(c, hipMemcpy(tmp.data(), c->SL[gen] + (size_t)slot * c->ld, (size_t)c->n * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < c->n; ++i) sumL_q[i] = tmp[(size_t)c->h_pi[(size_t)i]];
    }
    return RC_OK;
}

extern "C" int32_t rc_kernel_timing(rc_ctx *c, int32_t enable, double *bulk_ms_total, int64_t *bulk_launches)
{
    if (!c) return fail(c, RC_ERR_ARG, "rc_kernel_timing: NULL ctx");
    HIPCHK(c, hipSetDevice(c->dev));
    int32_t rc = sync_and_check(c, true);
    if (rc != RC_OK) return rc;
    if (bulk_ms_total) *bulk_ms_total = c->bulk_ms;
    if (bulk_launches) *bulk_launches = c->bulk_launches;
    // (the two events of a timed launch ride in its dispatch — hipExtLaunchKernelGGL — and report the kernel's own start and stop:
    // nothing to calibrate away; round 1 recorded marker pairs around the launch and subtracted what such a pair reports around an
    // empty kernel, ≈6 µs.  ev_overhead_ms stays 0.)
    if (enable >= 0) {
        c->timing = enable != 0;
        c->timing_every = enable > 1 ? enable : 1;
        c->bulk_ms = 0.0;
        c->bulk_launches = 0;
    }
    return RC_OK;
}

// ===================================================================================================
// Split–merge step (src/mcmc.jl:356-479) — SURVEY.md §8f-1.
// The proposal is a short sequential scalar computation over the |S| members of two clusters (restricted Gibbs
// scans, src/mcmc.jl:259-354): it runs on the host, as it does in the reference, on host matrices borrowed from
// the caller (MCMCData.D / .logD), in the reference's literal arithmetic including its quirks Q1–Q3 (SURVEY.md
// §3.2).  The device supplies what is data-parallel: both log-likelihoods of the acceptance ratio (mcmc.jl:462-464)
// from the exact S table — the proposed state is applied with k_apply_moves and, unless accepted, reverted
// bit-exactly (integer corrections).
// ===================================================================================================
static double rc_uniform_mh(uint64_t seed, uint64_t iter, uint64_t mh, uint64_t draw)
{
    uint32_t c[4] = {(uint32_t)draw, (uint32_t)mh, (uint32_t)iter, (uint32_t)(iter >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ 0x4D485F52u;
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const uint64_t bits = (((uint64_t)c[0] << 32) | c[1]) >> 12;
    return ((double)bits + 0.5) * 0x1p-52;
}

// out = log.(D - Diagonal(D) + I) (types.jl:155) with libm's log of the caller's doubles — what the split–merge scans read (they are
// pinned in the reference's literal arithmetic on log(D), golden_mh.npz).  Rows dealt to the host's cores: a serial loop is 0.5 s at
// n = 8192 and 10 s at n = 32768.
static void host_log_matrix(const double *D, int64_t n, double *out)
{
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), (int64_t)64, n / 64 + 1}));
    auto rows = [&](int t) {
        for (int64_t i = t; i < n; i += nt)
            for (int64_t j = 0; j < n; ++j) out[(size_t)(i * n + j)] = (i == j) ? 0.0 : std::log(D[(size_t)(i * n + j)]);
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) {
        try { th.emplace_back(rows, t); }
        catch (const std::system_error &) { for (int u = t; u < nt; ++u) rows(u); break; }   // (no more threads: the rest here)
    }
    rows(0);
    for (auto &t : th) t.join();
}

extern "C" int32_t rc_attach_host_matrices(rc_ctx *c, const double *D, const double *logD_or_null)
{
    if (!c || !D) return fail(c, RC_ERR_ARG, "rc_attach_host_matrices: NULL argument");
    c->hostD = D;
    if (logD_or_null) {
        c->hostL = logD_or_null;
        c->ownL.clear();
        c->ownL.shrink_to_fit();
    } else {
        const size_t n = (size_t)c->n;
        try { c->ownL.resize(n * n); }
        catch (const std::bad_alloc &) { c->hostD = nullptr; c->hostL = nullptr; return fail(c, RC_ERR_OOM, "rc_attach_host_matrices: no host memory for the %zu x %zu logD (pass logD)", n, n); }
        host_log_matrix(D, (int64_t)n, c->ownL.data());
        c->hostL = c->ownL.data();
    }
    return RC_OK;
}

// labels (1-based, by point), sizes by label (length n), K of the current device state
static int32_t pull_labels(rc_ctx *c, std::vector<int64_t> &labels, std::vector<int64_t> &sizes, int64_t &K)
{
    std::vector<int> so, ssize, slabel;
    int32_t rc = pull_state(c, so, ssize, slabel, true);
    if (rc != RC_OK) return rc;
    labels.resize((size_t)c->n);
    sizes.assign((size_t)c->n, 0);
    for (int i = 0; i < c->n; ++i) labels[(size_t)i] = slabel[(size_t)so[(size_t)c->h_pi[(size_t)i]]];
    K = 0;
    for (int k = 0; k < c->kcap; ++k)
        if (slabel[(size_t)k] > 0) { sizes[(size_t)slabel[(size_t)k] - 1] = ssize[(size_t)k]; ++K; }
    return RC_OK;
}

// Moves the device state from labelling `cur` (which it currently holds) to `next`: slot tables, slot_of, exact S
// corrections in every S generation that is valid, derived tables and both perm generations.
static int32_t apply_labels(rc_ctx *c, const std::vector<int64_t> &cur, const std::vector<int64_t> &next)
{
    // perm / pslot hold slot ids, and a revert may re-create a dead cluster in a different slot: both perm
    // generations are always rebuilt (k_derive) and the sweep pipeline's change marker is reset accordingly
    const bool persist = true;
    const int n = c->n;
    HIPCHK(c, hipStreamSynchronize(c->sA));
    HIPCHK(c, hipStreamSynchronize(c->sB));
    HIPCHK(c, hipStreamSynchronize(c->sB2));
    int32_t rc = drain_events(c);
    if (rc != RC_OK) return rc;
    std::vector<int> ssize((size_t)c->kcap), slabel((size_t)c->kcap);
    for (int k = 0; k < c->kcap; ++k) { ssize[(size_t)k] = c->hsum->size_label[2 * k]; slabel[(size_t)k] = c->hsum->size_label[2 * k + 1]; }
    std::vector<int> slot_of_label((size_t)n + 1, -1);
    for (int k = 0; k < c->kcap; ++k)
        if (slabel[(size_t)k] > 0) slot_of_label[(size_t)slabel[(size_t)k]] = k;
    // new sizes; births take the lowest slot that is free now, before any slot freed by this batch
    std::vector<int> moved;
    for (int i = 0; i < n; ++i)
        if (cur[(size_t)i] != next[(size_t)i]) {
            if (next[(size_t)i] < 1 || next[(size_t)i] > n) return fail(c, RC_ERR_ARG, "apply_labels: label outside 1..n");
            moved.push_back(i);
        }
    if (moved.empty()) return RC_OK;
    c->state_version++;
    int free_scan = 0;
    for (int i : moved) {
        const int lab = (int)next[(size_t)i];
        if (slot_of_label[(size_t)lab] < 0) {
            while (free_scan < c->kcap && slabel[(size_t)free_scan] != 0) ++free_scan;
            if (free_scan >= c->kcap) {
                // one more cluster than slots: grow the tables (as a sweep does, recover_capacity), install the labelling the
                // device holds again and start over
                if (c->kcap_fixed || c->kcap >= c->kcap_max)
                    return fail(c, RC_ERR_CAPACITY, "split-merge: more than %d clusters (slot capacity%s)", c->kcap, c->kcap_fixed ? " fixed by RC_KCAP_FIXED" : " at the library's maximum");
                int32_t rg = resize_capacity(c, (int)std::min<long long>(c->kcap_max, 2ll * c->kcap));
                if (rg != RC_OK) return rg;
                rg = rc_set_state(c, cur.data());
                if (rg != RC_OK) return rg;
                return apply_labels(c, cur, next);
            }
            slot_of_label[(size_t)lab] = free_scan;
            slabel[(size_t)free_scan] = lab;
            ssize[(size_t)free_scan] = 0;
        }
    }
    std::vector<int> mv;
    mv.reserve(moved.size() * 3);
    for (int i : moved) {
        const int a = slot_of_label[(size_t)cur[(size_t)i]], b = slot_of_label[(size_t)next[(size_t)i]];
        ssize[(size_t)a] -= 1;
        ssize[(size_t)b] += 1;
        mv.push_back(c->h_pi[(size_t)i]); mv.push_back(a); mv.push_back(b);  // device arrays are in internal order
    }
    int K = 0, hi = c->last.slot_hi;
    for (int k = 0; k < c->kcap; ++k) {
        if (slabel[(size_t)k] > 0 && ssize[(size_t)k] == 0) slabel[(size_t)k] = 0;  // death: its S row is now exactly zero
        if (slabel[(size_t)k] > 0) { ++K; hi = std::max(hi, k + 1); }
    }
    // S generations that currently hold valid sums
    MoveList ML{};
    if (c->incremental && c->bulk_enq >= 0 && c->t_next > 0) {
        ML.gens[ML.ngens++] = c->inc_gen;
    } else if (c->bulk_enq >= 0) {
        if (c->t_next == 0) ML.gens[ML.ngens++] = 0;
        else {
            ML.gens[ML.ngens++] = (int)((c->t_next - 1) % 3);
            if (c->bulk_enq >= c->t_next) ML.gens[ML.ngens++] = (int)(c->t_next % 3);
        }
    }
    View V = make_view(c);
    if (mv.size() > c->d_moves_cap) {
        if (c->d_moves) (void)hipFree(c->d_moves);
        c->d_moves = nullptr;
        c->d_moves_cap = std::max(mv.size(), (size_t)3 * 4096);
        HIPCHK(c, hipMalloc(&c->d_moves, c->d_moves_cap * sizeof(int)));
    }
    HIPCHK(c, hipMemcpyAsync(c->d_moves, mv.data(), mv.size() * sizeof(int), hipMemcpyHostToDevice, c->sA));
    if (ML.ngens > 0) {
        ML.mv = c->d_moves;
        ML.count = (int)moved.size();
        k_apply_moves<<<c->ld / 512, 256, 0, c->sA>>>(V, ML);
        HIPCHK(c, hipGetLastError());
    }
    // tables
    std::vector<int> so((size_t)n);
    HIPCHK(c, hipMemcpyAsync(so.data(), c->slot_of, so.size() * sizeof(int), hipMemcpyDeviceToHost, c->sA));
    HIPCHK(c, hipStreamSynchronize(c->sA));
    for (size_t q = 0; q < moved.size(); ++q) so[(size_t)mv[3 * q]] = mv[3 * q + 2];
    // only K / slot_hi (and, when the change persists and both perm generations are rebuilt, last_change_sweep) are
    // touched: the other device scalars belong to the sweep pipeline
    const int kv[2] = {K, hi}, minus1 = -1;
    HIPCHK(c, hipMemcpyAsync(&c->sc->K, &kv[0], sizeof(int), hipMemcpyHostToDevice, c->sA));
    HIPCHK(c, hipMemcpyAsync(&c->sc->slot_hi, &kv[1], sizeof(int), hipMemcpyHostToDevice, c->sA));
    if (persist) HIPCHK(c, hipMemcpyAsync(&c->sc->last_change_sweep, &minus1, sizeof(int), hipMemcpyHostToDevice, c->sA));
    HIPCHK(c, hipMemcpyAsync(c->slot_of, so.data(), so.size() * sizeof(int), hipMemcpyHostToDevice, c->sA));
    HIPCHK(c, hipMemcpyAsync(c->slot_size, ssize.data(), ssize.size() * sizeof(int), hipMemcpyHostToDevice, c->sA));
    HIPCHK(c, hipMemcpyAsync(c->slot_label, slabel.data(), slabel.size() * sizeof(int), hipMemcpyHostToDevice, c->sA));
    {
        int32_t rcd = launch_derive(c, V, persist ? 1 : 0);
        if (rcd != RC_OK) return rcd;
    }
    HIPCHK(c, hipStreamSynchronize(c->sA));  // also keeps the stack buffers above alive until the copies are done
    c->last.K = K; c->last.slot_hi = hi;
    return RC_OK;
}

extern "C" int32_t rc_state_checkpoint(rc_ctx *c)
{
    if (!c) return fail(c, RC_ERR_ARG, "rc_state_checkpoint: NULL ctx");
    if (!c->have_state) return fail(c, RC_ERR_STATE, "rc_state_checkpoint: no state set");
    HIPCHK(c, hipSetDevice(c->dev));
    std::vector<int64_t> sizes;
    int64_t K;
    return pull_labels(c, c->checkpoint, sizes, K);
}

extern "C" int32_t rc_state_restore(rc_ctx *c)
{
    if (!c) return fail(c, RC_ERR_ARG, "rc_state_restore: NULL ctx");
    if (c->checkpoint.size() != (size_t)c->n) return fail(c, RC_ERR_STATE, "rc_state_restore: no checkpoint");
    HIPCHK(c, hipSetDevice(c->dev));
    std::vector<int64_t> cur, sizes;
    int64_t K;
    int32_t rc = pull_labels(c, cur, sizes, K);
    if (rc != RC_OK) return rc;
    return apply_labels(c, cur, c->checkpoint);
}

namespace {
// literal arithmetic of sample_labels_Gibbs_restricted! on the host matrices
struct Restricted {
    const rc_ctx *c;
    const double *D, *L;
    int64_t n;
    double r, p;
    uint64_t seed, iter, mh;
    double abratio, zgratio, lg_d1, lg_d2, logp;
    const std::vector<int64_t> *U;  // sorted indices of i, j and S: every member of the two candidate clusters
    std::vector<std::vector<int64_t>> fixed_members;  // members of C[1] / C[2] when they are not candidates
    int64_t fixed_label[2] = {-1, -1};

    // Row sums of item x over the members of the two candidate clusters, in ONE pass over U (ascending index, so
    // each cluster's sum is accumulated in ascending member order exactly as findall + matsum do, mcmc.jl:308-311,
    // utils.jl:9-17).  out[0..1] = D sums, out[2..3] = logD sums.
    // The members are kept as two sorted index lists (rebuilt from U at the start of a scan, updated as items move), so
    // the four sums are four independent add chains without a label test per element.
    std::vector<int64_t> mem[2];
    void cand_sums(int64_t x, double out[4]) const
    {
        double d0 = 0, d1 = 0, l0 = 0, l1 = 0;
        const double *Dx = D + (size_t)x * (size_t)n, *Lx = L + (size_t)x * (size_t)n;
        const std::vector<int64_t> &A = mem[0], &B = mem[1];
        const size_t na = A.size(), nb = B.size(), both = na < nb ? na : nb;
        size_t q = 0;
        for (; q < both; ++q) {
            const int64_t ya = A[q], yb = B[q];
            d0 += Dx[ya]; l0 += Lx[ya];
            d1 += Dx[yb]; l1 += Lx[yb];
        }
        for (size_t t = q; t < na; ++t) { d0 += Dx[A[t]]; l0 += Lx[A[t]]; }
        for (size_t t = q; t < nb; ++t) { d1 += Dx[B[t]]; l1 += Lx[B[t]]; }
        out[0] = d0; out[1] = d1; out[2] = l0; out[3] = l1;
    }
    static double list_sum(const double *M, int64_t x, int64_t n, const std::vector<int64_t> &mem)
    {
        double s = 0;
        for (int64_t y : mem) s += M[(size_t)x * (size_t)n + (size_t)y];
        return s;
    }
    // sums over the (static) members of C[1] / C[2] when they are not candidates: computed once per item and proposal
    std::vector<double> fixedD[2], fixedL[2];
    std::vector<char> fixed_have[2];
    // lgamma(α + δ1·sz) and lgamma(ζ + δ2·sz) depend on the integer size only: memoised (same values, fewer calls)
    std::vector<double> lgA, lgZ;
    double lg_alpha_i(double sz) { const size_t k = (size_t)sz; if (lgA.size() <= k) lgA.resize(k + 64, NAN); if (lgA[k] != lgA[k]) lgA[k] = std::lgamma(c->P.alpha + c->P.delta1 * sz); return lgA[k]; }
    double lg_zeta_i(double sz) { const size_t k = (size_t)sz; if (lgZ.size() <= k) lgZ.resize(k + 64, NAN); if (lgZ[k] != lgZ[k]) lgZ[k] = std::lgamma(c->P.zeta + c->P.delta2 * sz); return lgZ[k]; }
    // the prior-ratio term log(sz+1) + log p + log(sz−1+r) − log(sz) (mcmc.jl:312-313) also depends on the size only
    std::vector<double> lprM;
    double lpr_i(double sz)
    {
        const size_t k = (size_t)sz;
        if (lprM.size() <= k) lprM.resize(k + 64, NAN);
        if (lprM[k] != lprM[k] && sz > 0) lprM[k] = std::log(sz + 1) + logp + std::log(sz - 1 + r) - std::log(sz);
        return sz > 0 ? lprM[k] : std::log(sz + 1) + logp + std::log(sz - 1 + r) - std::log(sz);
    }
    // the L2 term of a non-candidate first cluster is the same in every scan of the proposal
    std::vector<double> fixedTerm[2];

    // one scan (mcmc.jl:302-352); returns log_transition_prob
    double scan(std::vector<int64_t> &clusts, std::vector<int64_t> &sizes, const std::vector<int64_t> &items,
                const int64_t cand[2], const std::vector<int64_t> *final_clusts, int64_t scan_index)
    {
        const rc_params &P = c->P;
        const double d1 = P.delta1, d2 = P.delta2, al = P.alpha, be = P.beta, ze = P.zeta, ga = P.gamma;
        // C = findall(clustsizes .> 0), frozen at entry (mcmc.jl:273); only C[1], C[2] are ever read (Q3)
        int64_t firsts[2] = {0, 0};
        for (int64_t k = 0, f = 0; k < n && f < 2; ++k)
            if (sizes[(size_t)k] > 0) firsts[f++] = k + 1;
        if (fixed_label[0] != firsts[0] || fixed_label[1] != firsts[1] || fixed_members.size() != 2) {
            fixed_members.assign(2, {});
            for (int t = 0; t < 2; ++t) {
                fixed_label[t] = firsts[t];
                fixedD[t].assign((size_t)n, 0.0); fixedL[t].assign((size_t)n, 0.0); fixed_have[t].assign((size_t)n, 0);
                fixedTerm[t].assign((size_t)n, 0.0);
                if (firsts[t] != 0 && firsts[t] != cand[0] && firsts[t] != cand[1])
                    for (int64_t y = 0; y < n; ++y)
                        if (clusts[(size_t)y] == firsts[t]) fixed_members[(size_t)t].push_back(y);
            }
        }
        const int64_t m = (int64_t)items.size();
        double ltp = 0;
        mem[0].clear(); mem[1].clear();
        for (int64_t y : *U) {
            if (clusts[(size_t)y] == cand[0]) mem[0].push_back(y);
            else if (clusts[(size_t)y] == cand[1]) mem[1].push_back(y);
        }
        for (int64_t q = 0; q < m; ++q) {
            const int64_t x = items[(size_t)q];
            {
                std::vector<int64_t> &from = mem[clusts[(size_t)x] == cand[0] ? 0 : 1];
                from.erase(std::lower_bound(from.begin(), from.end(), x));
            }
            sizes[(size_t)clusts[(size_t)x] - 1] -= 1;                                   // mcmc.jl:303
            clusts[(size_t)x] = -1;                                                      // mcmc.jl:304
            double L1[2], lpr[2], L2p_c[2], logprobs[2], cs[4];
            cand_sums(x, cs);
            for (int k = 0; k < 2; ++k) {                                                // mcmc.jl:307-326
                const double sz = (double)sizes[(size_t)cand[k] - 1];
                const double sD = cs[k];
                const double sL = cs[2 + k];
                const double a_i = al + d1 * sz, b_i = be + sD, z_i = ze + d2 * sz, g_i = ga + sD;
                L1[k] = lg_alpha_i(sz) + abratio - a_i * std::log(b_i) + (d1 - 1) * sL - sz * lg_d1;
                lpr[k] = lpr_i(sz);
                L2p_c[k] = lg_zeta_i(sz) - z_i * std::log(g_i) + zgratio + (d2 - 1) * sL - sz * lg_d2;
            }
            double L2p_first[2];
            for (int t = 0; t < 2; ++t) {                                                // mcmc.jl:327-331
                if (firsts[t] == 0) { L2p_first[t] = 0; continue; }
                if (firsts[t] == cand[0]) { L2p_first[t] = L2p_c[0]; continue; }
                if (firsts[t] == cand[1]) { L2p_first[t] = L2p_c[1]; continue; }
                const double sz = (double)sizes[(size_t)firsts[t] - 1];
                if (!fixed_have[t][(size_t)x]) {  // members of a non-candidate cluster do not change during the proposal
                    fixedD[t][(size_t)x] = list_sum(D, x, n, fixed_members[(size_t)t]);
                    fixedL[t][(size_t)x] = list_sum(L, x, n, fixed_members[(size_t)t]);
                    fixed_have[t][(size_t)x] = 1;
                    const double sD = fixedD[t][(size_t)x], sL = fixedL[t][(size_t)x];
                    const double z_i = ze + d2 * sz, g_i = ga + sD;
                    fixedTerm[t][(size_t)x] = lg_zeta_i(sz) - z_i * std::log(g_i) + zgratio + (d2 - 1) * sL - sz * lg_d2;
                }
                L2p_first[t] = fixedTerm[t][(size_t)x];
            }
            const double L2_i = L2p_first[0] + L2p_first[1];                             // Q3
            for (int k = 0; k < 2; ++k)
                logprobs[k] = lpr[k] + (L1[k] + (P.repulsion ? (L2_i - L2p_c[k]) : 0.0)); // mcmc.jl:333-335
            int k;
            if (!final_clusts) {                                                         // mcmc.jl:336-338
                const double mn = logprobs[0] < logprobs[1] ? logprobs[0] : logprobs[1];
                logprobs[0] -= mn; logprobs[1] -= mn;                                    // utils.jl:3 mutates its argument
                const uint64_t base = 4 + (uint64_t)m + 2 * (uint64_t)m * (uint64_t)scan_index + 2 * (uint64_t)q;
                const double g0 = -std::log(-std::log(rc_uniform_mh(seed, iter, mh, base))) + logprobs[0];
                const double g1 = -std::log(-std::log(rc_uniform_mh(seed, iter, mh, base + 1))) + logprobs[1];
                k = (g1 > g0) ? 1 : 0;
            } else {
                k = ((*final_clusts)[(size_t)x] == cand[0]) ? 0 : 1;                     // mcmc.jl:340-341
            }
            clusts[(size_t)x] = cand[k];                                                 // mcmc.jl:344-345
            sizes[(size_t)cand[k] - 1] += 1;
            mem[k].insert(std::lower_bound(mem[k].begin(), mem[k].end(), x), x);
            double mn = logprobs[0] < logprobs[1] ? logprobs[0] : logprobs[1];           // mcmc.jl:348 (Q2)
            if (logprobs[0] != logprobs[0] || logprobs[1] != logprobs[1]) mn = NAN;      // minimum() propagates NaN
            const double p0 = std::exp(logprobs[0] + mn), p1 = std::exp(logprobs[1] + mn);
            ltp += std::log((k ? p1 : p0) / (p0 + p1));                                  // mcmc.jl:349-351
        }
        return ltp;
    }
};
}  // namespace

// The host part of one proposal of the MH loop of sample_labels! (src/mcmc.jl:374-473), as a pure function of a SNAPSHOT
// of the state: labels, sizes by label, K, and the exact block sums B[t][k] of that state with the slot tables they are
// indexed by.  Touches neither the device nor the context (only its parameters and the borrowed host matrices), so the
// chain loop can run it on worker threads for several iterations at once (chain.inc.hip).  A merge is decided here
// completely — block sums are additive: B'(m, l) = B(ci, l) + B(cj, l), B'(m, m) = B(ci,ci) + B(cj,cj) + 2 B(ci,cj), so the
// merged state's log-likelihood follows from the snapshot's integers, the same ones k_blocksums would produce after
// applying the merge.  A split needs the row sums of the two new clusters: needs_device is set and the caller applies
// cfinal on the device, takes its log-likelihood and finishes the decision (finish_decision).
struct ProposalSnapshot {
    const int64_t *clusts, *sizes;   // n labels, n sizes by label
    int64_t K;
    int hi;                          // block sums: hi × hi × 4 int64, slot sizes and labels [hi]
    const int *ssize, *slabel;
    const long long *B;
};
struct ProposalResult {
    bool accept = false, split = false, skipped = false, needs_device = false;
    double log_prior_ratio = 0, log_proposal_ratio = 0, ll_cur = 0, ll_fin = 0;
    std::vector<int64_t> cfinal;     // proposed labels (filled for splits and for accepted merges)
    int32_t err = RC_OK;
    const char *errmsg = nullptr;
};

static bool finish_decision(ProposalResult &R, uint64_t seed, uint64_t iter, uint64_t mh_counter)
{
    const double x = R.log_prior_ratio + (R.ll_fin - R.ll_cur) - R.log_proposal_ratio;
    const double lar = (x != x) ? NAN : (x < 0 ? x : 0.0);                                // minimum([0, x]) propagates NaN
    const double lu = std::log(rc_uniform_mh(seed, iter, mh_counter, 2));
    R.accept = lu < lar;                                                                 // mcmc.jl:469-472
    return R.accept;
}

static void proposal_core(const rc_ctx *c, rc_ctx::LLCache &cache, const ProposalSnapshot &S0, double r, double p, int64_t numGibbs,
                          uint64_t seed, uint64_t iter, uint64_t mh_counter, ProposalResult &out, bool profile)
{
    const int64_t n = c->n;
    const int64_t *clusts = S0.clusts, *sizes = S0.sizes;
    const int64_t K = S0.K;
    // chaperones (mcmc.jl:379)
    int64_t i = (int64_t)std::floor(rc_uniform_mh(seed, iter, mh_counter, 0) * (double)n);
    int64_t j = (int64_t)std::floor(rc_uniform_mh(seed, iter, mh_counter, 1) * (double)(n - 1));
    if (i >= n) i = n - 1;
    if (j >= n - 1) j = n - 2;
    if (j >= i) j += 1;
    const int64_t ci = clusts[(size_t)i], cj = clusts[(size_t)j];
    if (c->P.maxK > 0 && ci == cj && K >= c->P.maxK) { out.skipped = true; return; }      // mcmc.jl:384-386
    std::vector<int64_t> S, U;
    for (int64_t k = 0; k < n; ++k)
        if (clusts[(size_t)k] == ci || clusts[(size_t)k] == cj) {
            U.push_back(k);
            if (k != i && k != j) S.push_back(k);                                        // mcmc.jl:389-390
        }
    if ((uint64_t)S.size() * (uint64_t)(2 * numGibbs + 4) + 8 > 0xFFFFFFFFull) {
        out.err = RC_ERR_ARG; out.errmsg = "rc_splitmerge: uniform counter overflow (|S| * numGibbs too large)"; return;
    }
    std::vector<int64_t> claunch(clusts, clusts + n), szlaunch(sizes, sizes + n);
    if (ci == cj) {                                                                      // mcmc.jl:396-401
        int64_t e = 0;
        while (e < n && sizes[(size_t)e] != 0) ++e;
        if (e >= n) { out.err = RC_ERR_STATE; out.errmsg = "rc_splitmerge: no empty label for a split"; return; }
        claunch[(size_t)i] = e + 1;
        szlaunch[(size_t)ci - 1] -= 1;
        szlaunch[(size_t)e] += 1;
    }
    const int64_t cand[2] = {claunch[(size_t)i], claunch[(size_t)j]};                    // mcmc.jl:402
    for (size_t q = 0; q < S.size(); ++q) {                                              // mcmc.jl:403-407
        const int64_t k = S[q];
        claunch[(size_t)k] = cand[rc_uniform_mh(seed, iter, mh_counter, 4 + (uint64_t)q) < 0.5 ? 0 : 1];
        szlaunch[(size_t)clusts[(size_t)k] - 1] -= 1;
        szlaunch[(size_t)claunch[(size_t)k] - 1] += 1;
    }
    const rc_params &P = c->P;
    Restricted R;
    R.c = c; R.D = c->hostD; R.L = c->hostL; R.n = n; R.r = r; R.p = p; R.seed = seed; R.iter = iter; R.mh = mh_counter;
    R.abratio = P.alpha * std::log(P.beta) - std::lgamma(P.alpha);                       // mcmc.jl:293-297
    R.zgratio = P.zeta * std::log(P.gamma) - std::lgamma(P.zeta);
    R.lg_d1 = std::lgamma(P.delta1); R.lg_d2 = std::lgamma(P.delta2); R.logp = std::log(p);
    R.U = &U;
    if (profile) g_smprof.lap(1);
    for (int64_t s = 0; s < numGibbs; ++s) R.scan(claunch, szlaunch, S, cand, nullptr, s);  // mcmc.jl:411-414
    std::vector<int64_t> clusts_v;   // the merge's forced final scan wants the original labels as a vector
    if (ci == cj) {                                                                      // split, mcmc.jl:416-434
        out.split = true;
        const double ltp = R.scan(claunch, szlaunch, S, cand, nullptr, numGibbs);
        out.cfinal = claunch;
        const double sfi = (double)szlaunch[(size_t)claunch[(size_t)i] - 1], sfj = (double)szlaunch[(size_t)claunch[(size_t)j] - 1];
        const double sci = (double)sizes[(size_t)ci - 1];
        out.log_prior_ratio = std::log((double)(K + 1)) + r * std::log(1 - p) - std::log(p) - std::lgamma(r) +
                              std::lgamma(sfi - 1 + r) + std::lgamma(sfj - 1 + r) + std::log(sfi) + std::log(sfj) +
                              -(std::lgamma(sci - 1 + r) + std::log(sci));
        out.log_proposal_ratio = ltp;
    } else {                                                                             // merge, mcmc.jl:435-459
        std::vector<int64_t> szfinal = szlaunch;
        out.cfinal = claunch;
        int64_t sz_clust_i = 0;
        for (int64_t k : U)
            if (out.cfinal[(size_t)k] == ci) { out.cfinal[(size_t)k] = cj; ++sz_clust_i; }
        szfinal[(size_t)ci - 1] = 0;
        szfinal[(size_t)cj - 1] += sz_clust_i;
        const double sfj = (double)szfinal[(size_t)cj - 1], sci = (double)sizes[(size_t)ci - 1], scj = (double)sizes[(size_t)cj - 1];
        out.log_prior_ratio = -(std::log((double)K) + r * std::log(1 - p) - std::log(p) - std::lgamma(r)) +
                              std::lgamma(sfj - 1 + r) + std::log(sfj) +
                              -(std::lgamma(sci - 1 + r) + std::lgamma(scj - 1 + r) + std::log(sci) + std::log(scj));
        clusts_v.assign(clusts, clusts + n);
        const double ltp = R.scan(claunch, szlaunch, S, cand, &clusts_v, numGibbs);
        out.log_proposal_ratio = -ltp;
    }
    if (profile) g_smprof.lap(2);
    // likelihood ratio (mcmc.jl:462-464) from the exact block sums of the snapshot
    out.ll_cur = loglik_host_c(c, cache, S0.hi, S0.ssize, S0.B, S0.slabel);
    if (profile) g_smprof.lap(3);
    if (ci == cj) { out.needs_device = true; return; }
    const int hi = S0.hi;
    int si = -1, sj = -1;
    for (int k = 0; k < hi; ++k) {
        if (S0.ssize[k] > 0 && S0.slabel[k] == ci) si = k;
        if (S0.ssize[k] > 0 && S0.slabel[k] == cj) sj = k;
    }
    if (si < 0 || sj < 0) { out.err = RC_ERR_STATE; out.errmsg = "rc_splitmerge: internal: slots of the merged clusters not found"; return; }
    std::vector<long long> Bm(S0.B, S0.B + (size_t)hi * hi * 4);
    std::vector<int> szm(S0.ssize, S0.ssize + hi);
    for (int k = 0; k < hi; ++k)
        for (int w = 0; w < 4; ++w) Bm[((size_t)sj * hi + k) * 4 + w] += Bm[((size_t)si * hi + k) * 4 + w];
    for (int t = 0; t < hi; ++t)
        for (int w = 0; w < 4; ++w) Bm[((size_t)t * hi + sj) * 4 + w] += Bm[((size_t)t * hi + si) * 4 + w];
    szm[(size_t)sj] += szm[(size_t)si];
    szm[(size_t)si] = 0;
    out.ll_fin = loglik_host_c(c, cache, hi, szm.data(), Bm.data(), S0.slabel);   // the merged cluster keeps cj's slot and label
    if (profile) g_smprof.lap(5);
    finish_decision(out, seed, iter, mh_counter);
    if (!out.accept) out.cfinal.clear();
}

// One proposal of the MH loop of sample_labels! (src/mcmc.jl:374-473) on the current device state.  On acceptance
// the device state BECOMES the proposed state (what the reference's rebinding `state = finalstate`, mcmc.jl:470,
// means for its caller is the host loop's business: rc_state_checkpoint / rc_state_restore).
extern "C" int32_t rc_splitmerge(rc_ctx *c, double r, double p, int64_t numGibbs, uint64_t seed, uint64_t iter,
                                 uint64_t mh_counter, uint8_t *accept_out, uint8_t *split_out)
{
    if (!c || !accept_out || !split_out) return fail(c, RC_ERR_ARG, "rc_splitmerge: NULL argument");
    if (!c->have_params || !c->have_state) return fail(c, RC_ERR_STATE, "rc_splitmerge: params and state must be set");
    if (!c->hostD || !c->hostL) return fail(c, RC_ERR_STATE, "rc_splitmerge: call rc_attach_host_matrices first");
    if (!(r > 0.0) || !(p > 0.0 && p < 1.0) || numGibbs < 0) return fail(c, RC_ERR_ARG, "rc_splitmerge: need r > 0, 0 < p < 1, numGibbs >= 0");
    if (c->n < 2) return fail(c, RC_ERR_ARG, "rc_splitmerge: needs n >= 2 (sample(1:n, 2, replace=false))");
    HIPCHK(c, hipSetDevice(c->dev));
    *accept_out = 0; *split_out = 0;
    std::vector<int64_t> clusts, sizes;
    int64_t K;
    g_smprof.start();
    int32_t rc = pull_labels(c, clusts, sizes, K);
    if (rc != RC_OK) return rc;
    g_smprof.lap(0);
    // block sums of the current state (kept from the last evaluation when the labels have not changed since)
    if (!(c->ll_version == c->state_version && c->B_version == c->state_version)) {
        double ll_tmp = 0;
        rc = rc_loglik(c, &ll_tmp);
        if (rc != RC_OK) return rc;
    }
    ProposalSnapshot S0{clusts.data(), sizes.data(), K, c->B_hi, c->B_ssize.data(), c->B_slabel.data(), c->B_cur.data()};
    ProposalResult R;
    proposal_core(c, c->llc, S0, r, p, numGibbs, seed, iter, mh_counter, R, true);
    if (R.err != RC_OK) return fail(c, R.err, "%s", R.errmsg);
    if (R.skipped) return RC_OK;
    *split_out = R.split ? 1 : 0;
    if (!R.needs_device) {                                                               // merge: decided on the host
        if (R.accept) {
            *accept_out = 1;  // the proposed state goes to the device (tables and perm generations follow it)
            rc = apply_labels(c, clusts, R.cfinal);
            if (rc != RC_OK) return rc;
            g_smprof.lap(4);
            c->ll_cached = R.ll_fin; c->ll_version = c->state_version;
        }
        return RC_OK;
    }
    // split: the proposed state is applied on the device for its log-likelihood
    rc = apply_labels(c, clusts, R.cfinal);
    if (rc != RC_OK) return rc;
    g_smprof.lap(4);
    rc = rc_loglik(c, &R.ll_fin);
    if (rc != RC_OK) return rc;
    g_smprof.lap(5);
    if (finish_decision(R, seed, iter, mh_counter)) {
        *accept_out = 1;
        c->ll_cached = R.ll_fin; c->ll_version = c->state_version;
        return RC_OK;
    }
    rc = apply_labels(c, R.cfinal, clusts);                                              // rejected: revert, bit-exactly
    if (rc != RC_OK) return rc;
    g_smprof.lap(6);
    c->ll_cached = R.ll_cur; c->ll_version = c->state_version;                           // the reverted state is the evaluated one
    return RC_OK;
}


// RC_MODE_FULL (default): every sweep recomputes the row-sum table from the matrices, as the reference re-reads
// D and logD in every sweep (src/mcmc.jl:206-214) — the HBM-bound data flow the roofline metric is defined on.
// RC_MODE_INCREMENTAL: the table is computed once and then only corrected for label changes.  Both modes give
// bit-identical results (exact integer sums); the incremental one does no matrix traffic while labels are stable.
extern "C" int32_t rc_set_mode(rc_ctx *c, int32_t mode)
{
    if (!c) return fail(c, RC_ERR_ARG, "rc_set_mode: NULL ctx");
    if (mode != RC_MODE_FULL && mode != RC_MODE_INCREMENTAL) return fail(c, RC_ERR_ARG, "rc_set_mode: unknown mode %d", mode);
    HIPCHK(c, hipSetDevice(c->dev));
    const bool inc = (mode == RC_MODE_INCREMENTAL);
    c->want_incremental = inc;
    if (inc == c->incremental) return RC_OK;
    if (c->wide && !inc) return RC_OK;   // a wide context maintains its one table in place whatever the mode says (same results in both modes anyway)
    int32_t rc = sync_and_check(c, true);
    if (rc != RC_OK) return rc;
    if (inc) {
        if (c->have_state) {
            int gen = 0;
            rc = ensure_S(c, &gen);  // the generation holding the sums of the current labels
            if (rc != RC_OK) return rc;
            c->inc_gen = gen;
        }
        c->incremental = true;
        return RC_OK;
    }
    // back to full recomputation: restart the sweep pipeline from the current labels
    c->incremental = false;
    if (c->have_state) {
        for (int g = 0; g < 3; ++g) {
            HIPCHK(c, hipMemsetAsync(c->SD[g], 0, (size_t)c->kcap * c->ld * sizeof(long long), c->sA));
            HIPCHK(c, hipMemsetAsync(c->SL[g], 0, (size_t)c->kcap * c->ld * sizeof(long long), c->sA));
        }
        for (int g = 0; g < 2; ++g) {
            HIPCHK(c, hipMemsetAsync(c->keys[g], 0xFF, (size_t)(2 * c->n + 8) * sizeof(u64), c->sA));
            HIPCHK(c, hipMemsetAsync(c->cword[g], 0, 2 * ((size_t)(c->n + RC_PTS - 1) / RC_PTS + 1) * sizeof(u64), c->sA));
            HIPCHK(c, hipMemsetAsync(c->arrive[g], 0, RC_BAR_WORDS * sizeof(unsigned), c->sA));
            HIPCHK(c, hipMemsetAsync(c->work[g], 0, 64, c->sA));
        }
        const int minus1 = -1;
        HIPCHK(c, hipMemcpyAsync(&c->sc->last_change_sweep, &minus1, sizeof(int), hipMemcpyHostToDevice, c->sA));
        View V = make_view(c);
        rc = launch_derive(c, V, 1);
        if (rc != RC_OK) return rc;
        HIPCHK(c, hipStreamSynchronize(c->sA));
        c->t_next = 0;
        c->bulk_enq = -1;
        c->s_res_last = nullptr;
    }
    return RC_OK;
}


// Run-time options of one context (include/redclust_hip.h).  Their defaults are read from the environment ONCE, when the context is
// created; nothing reads the environment per sweep or per chain, so contexts driven from different host threads (rc_run_chains,
// one chain per GPU) can be configured independently.
extern "C" int32_t rc_set_option(rc_ctx *c, const char *name, int64_t value)
{
    if (!c || !name) return fail(c, RC_ERR_ARG, "rc_set_option: NULL argument");
    if (!strcmp(name, "prune")) {
        if (value < -1 || value > 1) return fail(c, RC_ERR_ARG, "rc_set_option: prune must be -1 (automatic), 0 (never) or 1 (always)");
        c->opt_prune = (int)value;
    } else if (!strcmp(name, "lds_point_cache")) {
        if (value != 0 && value != 1) return fail(c, RC_ERR_ARG, "rc_set_option: lds_point_cache must be 0 or 1");
        c->opt_cu_cache = (int)value;
    } else if (!strcmp(name, "chain_workers")) {
        if (value < 0 || value > 1024) return fail(c, RC_ERR_ARG, "rc_set_option: chain_workers must be in 0..1024 (0 = automatic)");
        c->opt_chain_workers = (int)value;
    } else if (!strcmp(name, "chain_depth")) {
        if (value < 0 || value > 64) return fail(c, RC_ERR_ARG, "rc_set_option: chain_depth must be in 0..64 (0 = automatic)");
        c->opt_chain_depth = (int)value;
    } else if (!strcmp(name, "chain_pipeline")) {
        if (value != 0 && value != 1) return fail(c, RC_ERR_ARG, "rc_set_option: chain_pipeline must be 0 or 1");
        c->opt_chain_pipeline = (int)value;
#ifdef RC_DIAG
    } else if (!strcmp(name, "debug_flags")) {   // timing ablations (SweepArgs.dbg): results are wrong on purpose
        c->dbg = (int)value;
#endif
    } else {
        return fail(c, RC_ERR_ARG, "rc_set_option: unknown option '%s' (prune, lds_point_cache, chain_workers, chain_depth, chain_pipeline)", name);
    }
    return RC_OK;
}

// Which row-reduction kernel the last enqueued sweep used (0: k_bulk, full read; 1: one of the symmetric kernels, upper triangle only)
// and the bytes of matrix data it has to read per launch — what bench.py prices the roofline against.
extern "C" int32_t rc_bulk_kernel_info(rc_ctx *c, int32_t *which, double *algorithmic_bytes)
{
    if (!c) return fail(c, RC_ERR_ARG, "rc_bulk_kernel_info: NULL ctx");
    const double n = c->n, esz = c->bits / 8.0;
    if (which) *which = c->last_bulk_kernel;
    // matrices the kernel reads: D and logD, or D alone when logD is derived on the fly
    const double nmat = c->derived ? 1.0 : 2.0;
    const double esz_read = (c->last_bulk_kernel && sym_variant_of(c) == 3 && c->Dq48) ? 6.0 : esz;   // k_bulk_syml2 streams the 48-bit packed copy
    if (algorithmic_bytes) *algorithmic_bytes = c->last_bulk_kernel ? nmat * (n * (n + 1) / 2) * esz_read : nmat * n * n * esz;
    return RC_OK;
}


// name of the row-reduction kernel the last enqueued sweep used, as a profiler shows it
extern "C" const char *rc_bulk_kernel_name(rc_ctx *c)
{
    if (!c) return "";
    if (!c->last_bulk_kernel) return c->derived ? "k_bulk<long long, true>" : (c->bits == 64 ? "k_bulk<long long, false>" : "k_bulk<int, false>");
    if (c->bits != 64) return sym_variant_of(c) == 2 ? "k_bulk_syml32" : "k_bulk_sym32";
    const int v = sym_variant_of(c);
    if (v == 3) return c->derived ? (c->Dq48 ? "k_bulk_syml2<true, true>" : "k_bulk_syml2<true, false>") : "k_bulk_syml2<false, false>";
    if (v == 2) return c->derived ? "k_bulk_syml<true>" : "k_bulk_syml<false>";
    if (v == 1) return c->derived ? "k_bulk_symw<true>" : "k_bulk_symw<false>";
    return c->derived ? "k_bulk_sym<true>" : "k_bulk_sym<false>";
}

extern "C" int32_t rc_layout_info(rc_ctx *c, int32_t *n_relayouts, int32_t *label_runs)
{
    if (!c) return fail(c, RC_ERR_ARG, "rc_layout_info: NULL ctx");
    HIPCHK(c, hipSetDevice(c->dev));
    int32_t rc = sync_and_check(c);
    if (rc != RC_OK) return rc;
    if (n_relayouts) *n_relayouts = c->n_relayouts;
    if (label_runs) *label_runs = c->hsum->runs;
    return RC_OK;
}

// Selects the row-reduction kernel: -1 automatic (by label-run count), 0 k_bulk (full read), 1 k_bulk_sym (upper
// triangle).  Both are exact for every labelling; this only exists for tests and measurements.
extern "C" int32_t rc_set_bulk_kernel(rc_ctx *c, int32_t which)
{
    if (!c) return fail(c, RC_ERR_ARG, "rc_set_bulk_kernel: NULL ctx");
    if (which < -1 || which > 1) return fail(c, RC_ERR_ARG, "rc_set_bulk_kernel: which must be -1, 0 or 1");
    c->bulk_kernel = which;
    return RC_OK;
}


// What is subtracted from every timed launch: 0 — the two events of a timed launch ride in its dispatch (enqueue_bulk) and report the
// kernel's own start and stop; nothing is calibrated any more (round 1 recorded marker pairs and subtracted what a pair reports around
// an empty kernel).  Kept so that the bench line can state it.
extern "C" int32_t rc_event_overhead_ms(rc_ctx *c, double *out)
{
    if (!c || !out) return fail(c, RC_ERR_ARG, "rc_event_overhead_ms: NULL argument");
    *out = c->ev_overhead_ms;
    return RC_OK;
}

#if defined(RC_PROF_SYML) || defined(RC_TRACE_RESOLVE)
extern "C" int32_t rc_debug_prof(rc_ctx *c, int32_t gen, long long *out /* 8192 x 4 */)
{
    HIPCHK(c, hipSetDevice(c->dev));
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(out, (char *)c->work[gen & 1] + 64, 8192 * 128, hipMemcpyDeviceToHost));
    return RC_OK;
}
#endif

// Streaming-read ceiling of the device as this box delivers it ("also report against a measured
// device-to-device streaming-read ceiling"): a buffer far larger than the 256 MiB Infinity Cache read once per launch with
// 16-byte-per-lane non-temporal loads — the access pattern of the row reductions — timed with HIP events; the best of
// `reps` launches.  Independent of any context.
__global__ __launch_bounds__(256) void k_read_ceiling(const ll2 *__restrict__ p, size_t n16, long long *out)
{
    long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const ll2 v = __builtin_nontemporal_load(p + i);
        acc += v.x + v.y;
    }
    if (acc == 0x7fffffffffffffffll) out[0] = acc;
}

extern "C" int32_t rc_measure_read_ceiling(int32_t device, int64_t mib, int32_t reps, double *gbps_out)
{
    if (!gbps_out || mib < 64 || mib > 65536 || reps < 1) return fail(nullptr, RC_ERR_ARG, "rc_measure_read_ceiling: need 64 <= mib <= 65536, reps >= 1 and an output");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, RC_ERR_HIP, "rc_measure_read_ceiling: device %d: %s", device, hipGetErrorString(hipGetLastError()));
    const size_t bytes = (size_t)mib << 20;
    void *buf = nullptr;
    long long *out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() { if (buf) (void)hipFree(buf); if (out) (void)hipFree(out); if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); };
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc((void **)&out, 64) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess ||
        hipMemset(buf, 1, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        const hipError_t e = hipGetLastError();
        cleanup();
        return fail(nullptr, RC_ERR_HIP, "rc_measure_read_ceiling: %s", hipGetErrorString(e));
    }
    double best = 0;
    for (int it = 0; it < reps + 1; ++it) {   // the first launch is a warm-up
        (void)hipEventRecord(e0, 0);
        k_read_ceiling<<<8192, 256>>>((const ll2 *)buf, bytes / 16, out);
        (void)hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) { const hipError_t e = hipGetLastError(); cleanup(); return fail(nullptr, RC_ERR_HIP, "rc_measure_read_ceiling: %s", hipGetErrorString(e)); }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms > 0) best = std::max(best, (double)bytes / (ms * 1e-3) / 1e9);
    }
    cleanup();
    *gbps_out = best;
    return RC_OK;
}

#include "pointestimate.inc.hip"
#include "chain.inc.hip"
#include "chains.inc.hip"
