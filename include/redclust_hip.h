/*
 * redclust_hip.h — C ABI of libredclust_hip.so: the MI355X (gfx950) implementation of RedClust.jl's
 * per-iteration Gibbs label sweep and the observables computed from the same device-resident data.
 *
 * The reference (RedClust.jl v1.2.2, pure Julia) has no FFI seam on this path: sample_labels_Gibbs! is an
 * ordinary Julia function.  This header DEFINES the seam; each entry point cites the reference code it
 * replaces (path:line under the reference checkout).  The Julia-side binding (ccall stubs keeping
 * runsampler / MCMCData / MCMCOptionsList / MCMCResult unchanged) is shown in INTEGRATION.md and
 * julia/RedClustHIP.jl; the Python host (redclust.jl_amd/) binds the same symbols through ctypes.
 *
 * Conventions
 *   - every function returns int32_t: 0 = RC_OK, negative = error class; rc_last_error() gives the text.
 *     Nothing aborts and nothing throws across the boundary.
 *   - labels are Julia Int = int64_t, 1-based, in 1..n (src/types.jl:1,135); matrices are n×n Float64,
 *     column-major == row-major because MCMCData enforces exact symmetry (src/types.jl:149-151).
 *   - the caller owns every host buffer and keeps it alive for the duration of the call; the library owns
 *     all device memory behind rc_ctx.  One rc_ctx is not thread-safe; distinct contexts are independent.
 *   - uniforms: the m uniforms sample_logweights (src/utils.jl:4) draws for point i (0-based) of sweep t
 *     are Philox4x32-10(key = seed)(counter = (key, i, t_lo, t_hi)), key = the candidate cluster's label
 *     (1..n), 0 for the new-cluster candidate; u = (top 52 bits + 0.5) * 2^-52  (DESIGN.md "Uniform stream").
 */
#ifndef REDCLUST_HIP_H
#define REDCLUST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RC_OK 0
#define RC_ERR_ARG (-1)      /* bad argument (null pointer, n < 1, label outside 1..n, r <= 0, p outside (0,1)) */
#define RC_ERR_HIP (-2)      /* HIP runtime error, including "no device" */
#define RC_ERR_OOM (-3)      /* host or device allocation failed */
#define RC_ERR_DOMAIN (-4)   /* D not symmetric / not finite / non-positive off-diagonal entry (log D = -Inf) */
#define RC_ERR_STATE (-5)    /* call sequence error (params or state not set) */
#define RC_ERR_CAPACITY (-6) /* more clusters than the library holds (min(n, 32767)), or than a capacity fixed with RC_KCAP_FIXED */

typedef struct rc_ctx rc_ctx;

/* Fields read by the sweep at src/mcmc.jl:171-178 (δ1 δ2 α β ζ γ repulsion maxK) and by logprior at
 * src/mcmc.jl:65-68 (η σ u v); struct PriorHyperparamsList, src/types.jl:93-108. */
typedef struct rc_params {
    double delta1, delta2, alpha, beta, zeta, gamma;
    double eta, sigma, u, v;
    int64_t maxK;      /* 0 = unbounded */
    uint8_t repulsion; /* Julia Bool */
    uint8_t pad_[7];
} rc_params;

/* Counters of the last rc_gibbs_sweep (diagnostics; no reference counterpart). */
typedef struct rc_sweep_stats {
    int64_t n_changes; /* points whose label changed in the sweep */
    int64_t n_rounds;  /* resolve rounds (tentative scoring passes); between 1 and 1 + n_changes */
    int64_t K;         /* clusters after the sweep (state.K, src/mcmc.jl:254) */
} rc_sweep_stats;

/* MCMCData constructor, src/types.jl:145-157.  Copies the n×n matrix D to HBM once, checks symmetry
 * (types.jl:149-151) and derives logD = log.(D - Diagonal(D) + I) on the device (types.jl:155; diagonal 0,
 * D's own diagonal kept as stored) unless the caller passes its own logD (e.g. MCMCData.logD).
 * With logD_or_null == NULL and 64-bit storage the fixed-point logD is not stored at all: every consumer evaluates it
 * from the stored D with one shared table-based log (DESIGN.md "Derived logD"; results are exact integers either way,
 * the row reduction reads half the bytes).  rc_get_matrix(ctx, 1, ·) returns the values in use.  RC_STORED_LOG=1
 * forces the stored form.
 * storage_bits: 64 (int64 fixed point; the Float64 path) or 32 (int32 fixed point: every entry rounded to
 * 2^-30 of the largest magnitude, half the HBM traffic; all sums stay exact 64-bit integers and scores stay f64
 * — the counterpart of BASELINE config 5's Float32 storage, which the reference itself does not have).
 * kcap: INITIAL slot capacity (clusters the sweep kernel's tables hold).  It grows on demand — rc_set_state with more
 * clusters, a sweep or a split–merge proposal that needs one more slot: the tables are doubled, the sweep is resumed at the
 * point that needed the slot and the sweeps enqueued behind it are replayed; the chain is exactly the one a larger capacity
 * would have produced.  Up to 4096 slots the kernel's tables live in LDS (the fast path).  Beyond — the reference's clustsizes
 * has length n, src/types.jl:131-137, src/mcmc.jl:198-199 — the context becomes WIDE: tables in global memory, one row-sum
 * table corrected in place, the sweep point by point on one workgroup (tens to hundreds of ms per sweep; the same draws), up
 * to min(n, 32767) clusters (slot ids are 16-bit); more is RC_ERR_CAPACITY.  A wide context NARROWS again once its state is down
 * to 1024 clusters (a chain started from all singletons collapses within a sweep or two): rc_set_state, and rc_gibbs_sweep[_async]
 * between two sweeps (also those of rc_run_chain), re-install the labels with a capacity sized by the state — the same chain.
 * 0 = automatic: sized from the first state (twice
 * its cluster count, at least 128, on the fast path while the clusters fit it).  Small capacities are faster (the tables sit
 * beside more row-reduction blocks on a CU); rc_capacity_info reports the current one.
 * device_id: HIP device ordinal. */
int32_t rc_create(int64_t n, const double *D, const double *logD_or_null, int32_t storage_bits,
                  int32_t device_id, int64_t kcap, rc_ctx **out);
/* MCMCData(points) constructor, src/types.jl:159-162: D = pairwise(Euclidean(), makematrix(pnts), dims=2) is
 * computed on the device from the n×dim row-major points (the n×n matrix never exists on the host), then
 * everything proceeds as in rc_create.  SURVEY.md §8f-2. */
int32_t rc_create_from_points(int64_t n, int64_t dim, const double *points, int32_t storage_bits, int32_t device_id,
                              int64_t kcap, rc_ctx **out);
/* The matrix the device actually holds (which = 0: D, 1: logD), as doubles: value = q·2^-e exactly. */
int32_t rc_get_matrix(rc_ctx *ctx, int32_t which, double *out_n_by_n);
/* Selected rows (0-based, caller's point order) of the same matrix: out is nrows×n row-major.  For sizes where the
 * n×n doubles of rc_get_matrix do not fit on the host (BASELINE config 5: 8 GiB). */
int32_t rc_get_matrix_rows(rc_ctx *ctx, int32_t which, const int64_t *rows, int64_t nrows, double *out);
int32_t rc_destroy(rc_ctx *ctx);

/* Last error text of ctx (or of the calling thread's last failed rc_create when ctx == NULL). */
const char *rc_last_error(const rc_ctx *ctx);

int32_t rc_set_params(rc_ctx *ctx, const rc_params *params);

/* MCMCState, src/types.jl:131-137: clustsizes and K are derived from the labels (types.jl:135-136). */
int32_t rc_set_state(rc_ctx *ctx, const int64_t *clusts /* n, 1-based */);
/* Any of the three outputs may be NULL.  With clusts == NULL nothing is copied from the device: sizes and K come
 * from the summary the sweep kernel leaves in host-mapped memory (the per-iteration need of sample_r!/sample_p!). */
int32_t rc_get_state(rc_ctx *ctx, int64_t *clusts /* n */, int64_t *clustsizes /* n, by label */, int64_t *K);

/* sample_labels_Gibbs!(data, state, params), src/mcmc.jl:158-256 — one full sequential sweep with the
 * r, p of the current iteration (mcmc.jl:169-170); mutates the device state exactly as the sequential loop
 * would under the uniform stream (seed, sweep_index).  Blocking. */
int32_t rc_gibbs_sweep(rc_ctx *ctx, double r, double p, uint64_t seed, uint64_t sweep_index);
int32_t rc_last_sweep_stats(rc_ctx *ctx, rc_sweep_stats *out);
/* Current slot capacity, its ceiling min(n, 32767) (beyond 4096 slots the context is wide: see rc_create), the number of growths so far and the resolver's batch capacity (each
 * pointer may be NULL).  Diagnostics; no reference counterpart. */
int32_t rc_capacity_info(rc_ctx *ctx, int64_t *kcap, int64_t *kcap_max, int64_t *n_grows, int64_t *batch_capacity);

/* Data flow of the sweep.  RC_MODE_FULL (default) recomputes the row-sum table S[k][i] = Σ_j D[i,j]·[c_j = k] from
 * the matrices in every sweep — what the reference does (src/mcmc.jl:206-214) and what the HBM roofline metric
 * is defined on.  RC_MODE_INCREMENTAL computes it once and then only corrects it for label changes; because the
 * sums are exact integers both modes give bit-identical results, and a sweep that changes no label reads no
 * matrix data at all.  Can be switched at any time. */
#define RC_MODE_FULL 0
#define RC_MODE_INCREMENTAL 1
int32_t rc_set_mode(rc_ctx *ctx, int32_t mode);

/* Non-blocking sweep for the stationary fast path and for benchmarking: enqueues one sweep on the
 * context's stream and returns; label changes are resolved on the device.  rc_synchronize() (or any
 * blocking call) waits for completion. */
int32_t rc_gibbs_sweep_async(rc_ctx *ctx, double r, double p, uint64_t seed, uint64_t sweep_index);
int32_t rc_synchronize(rc_ctx *ctx);

/* loglik(data, state, params), src/mcmc.jl:1-56. */
int32_t rc_loglik(rc_ctx *ctx, double *out);
/* logprior(state, params), src/mcmc.jl:58-78, with the given r, p. */
int32_t rc_logprior(rc_ctx *ctx, double r, double p, double *out);

/* Recording step of runsampler, src/mcmc.jl:546-553: canonical labels = sortlabels(state.clusts)
 * (src/utils.jl:69-74) written to canonical_out (nullable), and counts += adjacencymatrix(clusts)
 * (src/utils.jl:59-63), the running sum behind src/mcmc.jl:560. */
int32_t rc_record_sample(rc_ctx *ctx, int64_t *canonical_out /* n or NULL */);
/* posterior_coclustering = sum(adjacencymatrix.(result.clusts)) ./ numsamples, src/mcmc.jl:560. */
int32_t rc_cocluster(rc_ctx *ctx, double *out_n_by_n, int64_t numsamples);
/* Raw integer counts (exact), e.g. for cross-chain reduction on the host. */
int32_t rc_cocluster_counts(rc_ctx *ctx, uint32_t *out_n_by_n);
/* Device address and leading dimension (elements) of the uint32 count matrix, for a device-side
 * all-reduce across chains (RCCL through torch.distributed); valid until rc_destroy. */
int32_t rc_cocluster_device_buffer(rc_ctx *ctx, void **dev_ptr, int64_t *ld);
int32_t rc_cocluster_reset(rc_ctx *ctx);

/* ---- split–merge step -------------------------------------------------------------------
 * One proposal of the MH loop of sample_labels!, src/mcmc.jl:374-473 (chaperones, launch state, numGibbs
 * restricted scans sample_labels_Gibbs_restricted! src/mcmc.jl:259-354, split or merge bookkeeping, prior /
 * likelihood / proposal ratios, acceptance), restated as written including quirks Q2/Q3.  The
 * scalar scans run on the host on matrices borrowed with rc_attach_host_matrices (MCMCData.D and .logD; logD may
 * be NULL — the library then derives it); both log-likelihoods of mcmc.jl:462-464 come from the device.
 * Uniforms: Philox keyed (seed_lo, seed_hi ^ 0x4D485F52), counter (draw, mh_counter, iter_lo, iter_hi) — DESIGN.md.
 * On acceptance the device state becomes the proposed state.  The reference's `state = finalstate` (mcmc.jl:470)
 * only rebinds a local name (quirk Q1): a host loop that wants the reference's behaviour as written brackets the
 * iteration with rc_state_checkpoint / rc_state_restore and skips the Gibbs sweep after an acceptance. */
int32_t rc_attach_host_matrices(rc_ctx *ctx, const double *D, const double *logD_or_null);
int32_t rc_splitmerge(rc_ctx *ctx, double r, double p, int64_t numGibbs, uint64_t seed, uint64_t iter,
                      uint64_t mh_counter, uint8_t *accept_out, uint8_t *split_out);
int32_t rc_state_checkpoint(rc_ctx *ctx);
int32_t rc_state_restore(rc_ctx *ctx);

/* Introspection used by the parity tests: exact fixed-point row sums Σ_j D[i,j]·[c_j = label] of the
 * current state (the matsum(D,[i],clust_k) of src/mcmc.jl:210-213 before β is added), value = q·2^-e. */
int32_t rc_debug_rowsums(rc_ctx *ctx, int64_t label, int64_t *sumD_q /* n */, int64_t *sumL_q /* n */,
                         int32_t *eD, int32_t *eL);

/* Fixed-point row totals of the matrices as the caller gave them: totD_q[i] = Σ_j Dq[i,j], totL_q[i] = Σ_j Lq[i,j] (value =
 * q·2^-e, exponents from rc_debug_rowsums).  Computed by a plain per-row kernel that shares nothing with the row-reduction
 * kernels: Σ_labels rc_debug_rowsums(label)[i] must equal it (the full matsum(x) of src/utils.jl:18-24 row by row). */
int32_t rc_debug_rowtotals(rc_ctx *ctx, int64_t *totD_q /* n */, int64_t *totL_q /* n */);

/* The logarithms the sweep kernel scores candidates with, evaluated on the device for m caller-supplied arguments: which = 0
 * log(x) (x > 0, normal), 1 log1p(x) (x >= 0), 2 -log(-log(x)) (0 < x < 1: the Gumbel noise of src/utils.jl:4).  The reference
 * calls Julia's log / log1p (src/mcmc.jl:223-241); the kernel uses one table-driven routine of its own (<= 1.5 ulp on these
 * domains) — this entry lets the tests hold it against libm. */
int32_t rc_debug_flog(rc_ctx *ctx, int32_t which, const double *x, int64_t m, double *out /* m */);

/* Timing of the dominant kernel (row-bucket reduction) measured with HIP events on the stream it is launched
 * on: accumulated milliseconds and number of timed launches since the last reset.  enable: 0 = off, 1 = time
 * every launch, N > 1 = time every N-th launch, negative = just read the counters.  The two events of a timed
 * launch ride in its dispatch (hipExtLaunchKernelGGL) and report the kernel's own start and stop on that stream;
 * no marker packets are added around the kernel. */
int32_t rc_kernel_timing(rc_ctx *ctx, int32_t enable, double *bulk_ms_total, int64_t *bulk_launches);
/* What is subtracted from every timed launch: 0 (round 1 recorded marker pairs around the launch and subtracted
 * what such a pair reports around an empty kernel; kept so that the bench line can say so). */
int32_t rc_event_overhead_ms(rc_ctx *ctx, double *out);

/* Which row-reduction kernel the last enqueued sweep used — *which = 0: k_bulk (reads every entry of D and logD, any point
 * order), 1: one of the symmetric kernels (upper triangle only; chosen automatically when the points of a cluster are
 * contiguous in the internal point order, override with rc_set_bulk_kernel or RC_BULK_KERNEL=perm|sym|auto).  Which symmetric
 * kernel that is depends on the storage: k_bulk_syml2 (64-bit storage, the default: wave-autonomous units; with logD derived
 * it streams the 48-bit packed copy of D), k_bulk_sym (64-bit, logD stored: block-tiled), k_bulk_sym32 (32-bit storage) —
 * rc_bulk_kernel_name says which.  *algorithmic_bytes = the matrix bytes that kernel has to read per launch (what bench.py's
 * roofline prices): every entry of the matrices it reads for k_bulk, the upper triangle incl. diagonal for the symmetric
 * ones, at 6 bytes per entry for the packed copy. */
int32_t rc_bulk_kernel_info(rc_ctx *ctx, int32_t *which, double *algorithmic_bytes);
/* The kernel's name as a profiler shows it (k_bulk_syml2<true, true>: logD derived, packed copy; k_bulk<long long, false>,
 * k_bulk_sym<false>, k_bulk_sym32 ...). */
const char *rc_bulk_kernel_name(rc_ctx *ctx);
/* Force the kernel family: -1 automatic, 0 k_bulk (full read), 1 the symmetric kernel of this context (tests / measurements;
 * results are identical bit for bit). */
int32_t rc_set_bulk_kernel(rc_ctx *ctx, int32_t which);
/* Run-time options of one context.  Defaults come from the environment once, when the context is created (INTEGRATION.md
 * "Environment switches"); nothing reads the environment per sweep or per chain.
 *   "prune"          -1 automatic (default), 0 never, 1 always: candidates that cannot win skip their Gumbel noise (exact either way)
 *   "lds_point_cache" 1 (default) / 0: the sweep kernel keeps the internal indices and clusters of each workgroup's points in LDS
 *                    (possible while a workgroup owns at most 4 chunks of 32 points, i.e. n <= 128 x #CUs; larger problems and 0 read
 *                    them from global memory in every pass; same results)
 *   "chain_workers"  worker threads of rc_run_chain (0 = automatic: the host's cores shared by the chains of this process)
 *   "chain_depth"    iterations rc_run_chain keeps in flight (0 = automatic: 24)
 *   "chain_pipeline" 1 (default): the pipelined loop; 0: its synchronous form (the same chain bit for bit)
 * No reference counterpart (the reference has no tuning knobs on this path). */
int32_t rc_set_option(rc_ctx *ctx, const char *name, int64_t value);
/* Internal point layout.  rc_set_state stores D and logD with the points of a cluster contiguous (a stable sort of
 * the caller's points by label), so that k_bulk_sym applies whatever order the caller's points come in; the sweep
 * still visits the points in the caller's order and every output is in the caller's order.  Label movement
 * fragments the layout; in automatic kernel mode the library re-lays the points out when the number of label runs
 * in internal order exceeds n/32 and a fresh layout would be below it again (at most once per 32 sweeps up to n/64 clusters, per
 * 128 sweeps — doubling when a layout does not last — up to n/36; the chain is bit-identical either way).  Returns the
 * number of layouts built so far and the current run count.  RC_NO_RELAYOUT=1 keeps the caller's order throughout. */
int32_t rc_layout_info(rc_ctx *ctx, int32_t *n_relayouts, int32_t *label_runs);

/* The within- / between-cluster split of the pairwise dissimilarities under the CURRENT labels, as fitprior forms it
 * for its Gamma fits (src/prior.jl:73-75: A = upper-triangle entries of pairs in one cluster, B = the rest;
 * src/prior.jl:96-110 consume |A|, sum(A), sum(log A) and the same of B).  From the K×K block sums of the row-sum
 * table: no pass over the n×n matrices. */
typedef struct rc_wb_stats {
    int64_t count_within, count_between;
    double sum_within, sumlog_within, sum_between, sumlog_between;
} rc_wb_stats;
int32_t rc_within_between(rc_ctx *ctx, rc_wb_stats *out);

/* ---------------------------------------------------------------------------------------------------------------
 * The iteration loop of runsampler (src/mcmc.jl:533-556) as native host code: per iteration sample_r!, sample_p!
 * (src/mcmc.jl:80-155, on the build's counter-based scalar stream — DESIGN.md), sample_labels! (numMH split–merge
 * proposals, then the Gibbs sweep; src/mcmc.jl:356-479) and the recording rule (src/mcmc.jl:546-553).  Recorded
 * samples also enter the device co-clustering counts (read them with rc_cocluster afterwards).
 * ------------------------------------------------------------------------------------------------------------- */
#define RC_SM_AS_WRITTEN 0 /* split–merge exactly as the reference executes it (an accepted
                            * proposal is discarded together with that iteration's Gibbs scan) */
#define RC_SM_INTENDED 1   /* accepted proposals are kept and swept */

typedef struct rc_chain_options {
    int64_t numiters, burnin, thin; /* MCMCOptionsList (src/types.jl:3-43) */
    int64_t numGibbs, numMH;
    int32_t splitmerge_mode;        /* RC_SM_* */
    int32_t pad_;
    uint64_t seed;                  /* keys the label, split–merge and scalar streams */
    uint64_t first_iter;            /* stream index of the first iteration (0 for a fresh chain; continue with numiters) */
    double r0, p0;                  /* MCMCState.r / .p at the start */
    double proposalsd_r;            /* PriorHyperparamsList.proposalsd_r (src/types.jl:102) */
    const double *r_trace, *p_trace;/* both NULL: free-running; else numiters forced values (parity tests) */
    int64_t max_samples;            /* capacity of the per-sample output arrays */
} rc_chain_options;

typedef struct rc_chain_outputs {
    /* per recorded sample (capacity max_samples; each pointer may be NULL) — MCMCResult fields, src/types.jl:193-248 */
    int64_t *clusts;                /* max_samples × n, sortlabels'd (src/mcmc.jl:547) */
    int64_t *K;
    double *r, *p, *loglik, *logposterior;
    /* per iteration (may be NULL) */
    uint8_t *r_acceptances;         /* numiters */
    uint8_t *splitmerge_acceptances, *splitmerge_splits; /* numiters × numMH */
    double *r_all, *p_all;          /* numiters: the r and p every iteration used */
    /* scalars written on return */
    int64_t num_samples;
    double runtime_s;               /* wall time of the loop (MCMCResult.runtime, src/mcmc.jl:586) */
    double r_final, p_final;
} rc_chain_outputs;

int32_t rc_run_chain(rc_ctx *ctx, const rc_chain_options *opt, rc_chain_outputs *out);
/* Counters of the last rc_run_chain on this context (each pointer may be NULL): rollbacks of the speculative split–merge
 * pipeline (an accepted proposal voids the iterations launched behind it), split proposals evaluated off the live state,
 * host worker threads used, capacity growths during the run.  Diagnostics; no reference counterpart. */
int32_t rc_chain_stats(rc_ctx *ctx, int64_t *rollbacks, int64_t *split_evals, int64_t *workers, int64_t *grows);

/* ---------------------------------------------------------------------------------------------------------------
 * Chain-parallel execution: independent chains, one per GPU; the single exchange step is the SUM
 * all-reduce of the n×n uint32 co-clustering counts (and of the numbers of recorded samples) over RCCL.  The reference
 * has no multi-chain driver (runsampler, /root/reference/src/mcmc.jl:501-590, is one chain); the merged estimate is
 * Σ_chains counts / Σ_chains numsamples — what mcmc.jl:560 forms for one chain.
 * ------------------------------------------------------------------------------------------------------------- */
#define RC_COMM_ID_BYTES 128   /* sizeof(ncclUniqueId) */
typedef struct rc_comm rc_comm;

/* Rank 0 of a multi-process job calls this and sends the bytes to the other processes (MPI, a TCP store, a file ...). */
int32_t rc_comm_unique_id(uint8_t *id_out /* RC_COMM_ID_BYTES */);

/* A communicator over world_size chains; this process owns chains rank_offset .. rank_offset + n_local - 1, one per
 * entry of device_ids (distinct devices: one chain per GPU).  unique_id_or_null == NULL: all chains live in this
 * process (rank_offset = 0, world_size = n_local; ncclCommInitAll).  Collective over all participating processes. */
int32_t rc_comm_create(int32_t n_local, const int32_t *device_ids, int32_t rank_offset, int32_t world_size,
                       const uint8_t *unique_id_or_null, rc_comm **out);
int32_t rc_comm_destroy(rc_comm *comm);

/* In-place SUM all-reduce of the co-clustering counts of the n_local contexts (ctxs[i] on device_ids[i]) over all
 * chains, and of their numbers of recorded samples.  Afterwards every context holds the merged counts:
 * rc_cocluster(ctx, out, *total_samples) is the merged posterior co-clustering matrix.  elapsed_ms may be NULL. */
int32_t rc_comm_allreduce_counts(rc_comm *comm, rc_ctx *const *ctxs, const int64_t *num_samples /* n_local */,
                                 int64_t *total_samples, double *elapsed_ms);

typedef struct rc_chains_input {
    int64_t n;
    const double *D;             /* n×n, or NULL when points are given */
    const double *logD_or_null;  /* as rc_create */
    const double *points;        /* n×dim (rc_create_from_points), used when D == NULL */
    int64_t dim;
    int32_t storage_bits;        /* 64 or 32 */
    int32_t pad_;
    int64_t kcap;                /* initial slot capacity as rc_create, 0 = automatic */
    const rc_params *params;
    const int64_t *init_clusts;  /* n labels: every chain starts from them (MCMCState.clusts) */
} rc_chains_input;

/* n_chains chains in this process: one host thread and one context per device runs rc_run_chain with seed
 * opt->seed + chain index (outs[chain] as for rc_run_chain), then the counts are merged over RCCL.
 * posterior_coclustering (n×n, may be NULL): Σ counts / Σ numsamples.  total_samples, allreduce_ms may be NULL. */
int32_t rc_run_chains(int32_t n_chains, const int32_t *device_ids, const rc_chains_input *in, const rc_chain_options *opt,
                      rc_chain_outputs *outs /* n_chains */, double *posterior_coclustering, int64_t *total_samples,
                      double *allreduce_ms);

/* Measured streaming-read ceiling of a device in GB/s (SURVEY.md §8d asks for the roofline fraction against it as well as
 * against the nominal 8 TB/s): `mib` MiB — use far more than the 256 MiB Infinity Cache — read once per launch with the access
 * pattern of the row reductions, best of `reps` launches.  No context needed. */
int32_t rc_measure_read_ceiling(int32_t device, int64_t mib, int32_t reps, double *gbps_out);

/* One sample_r + sample_p pair exactly as rc_run_chain draws them (tests; host only, no GPU needed).  sizes: the K
 * non-empty cluster sizes in ascending label order. */
int32_t rc_scalar_updates(uint64_t seed, uint64_t iter, double r, double p, const int64_t *sizes, int64_t K, int64_t n,
                          double eta, double sigma, double proposalsd_r, double u, double v, double *r_out,
                          double *p_out, uint8_t *accept_out);

/* ---------------------------------------------------------------------------------------------------------------
 * Point estimation — the step after the sampler.  Stand-alone entry points (no rc_ctx): they
 * take host label vectors as MCMCResult.clusts holds them; errors are read with rc_last_error(NULL).
 * ------------------------------------------------------------------------------------------------------------- */

/* loss specifiers of getpointestimate(method = "MPEL") (src/pointestimate.jl:38-47) */
#define RC_LOSS_BINDER 0 /* "binder": randindex(x, y)[3] */
#define RC_LOSS_OMARI 1  /* "omARI":  1 - randindex(x, y)[1] */
#define RC_LOSS_VI 2     /* "VI":     varinfo(x, y) */
#define RC_LOSS_ID 3     /* "ID":     infodist(x, y; normalised = false) */

/* The MPEL search of getpointestimate (src/pointestimate.jl:49-58): lossmatrix[i][j] = loss(clusts[i], clusts[j])
 * for every pair of the m samples (computed for i < j and mirrored, zero diagonal), its column sums and the index
 * (0-based) of the first minimal one.  samples: m×n row-major (sample s = samples[s*n ..]), labels in 1..n.
 * lossmatrix (m×m), colsum (m), argmin and kernel_ms (device time of the pair kernel) may each be NULL. */
int32_t rc_loss_matrix(int32_t device, const int64_t *samples, int64_t m, int64_t n, int32_t loss,
                       double *lossmatrix, double *colsum, int64_t *argmin, double *kernel_ms);

/* Everything evaluateclustering (src/summaries.jl:12-23), binderloss and infodist (src/pointestimate.jl:68-99)
 * derive from a pair of labelings: Clustering.jl's randindex 4-tuple, mutualinfo (normed = false / true), varinfo,
 * the two entropies and the information distance (plain / normalised by max entropy as infodist does). */
typedef struct rc_pair_measures_t {
    double ari, ri, mirkin, hubert;
    double mi, nmi, vi;
    double ha, hb;
    double id, nid;
} rc_pair_measures_t;
int32_t rc_pair_measures(int32_t device, const int64_t *a, const int64_t *b, int64_t n, rc_pair_measures_t *out);

#ifdef __cplusplus
}
#endif
#endif /* REDCLUST_HIP_H */
