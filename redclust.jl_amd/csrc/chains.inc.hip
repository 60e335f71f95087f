// Chain-parallel execution in the library: N independent chains, one per GPU, and the one exchange
// step the path has — the element-wise SUM of the n×n uint32 co-clustering counts (plus the number of recorded samples)
// over RCCL.  The reference has no multi-chain driver (runsampler is one chain, /root/reference/src/mcmc.jl:501-590); the
// merged estimate is Σ_chains counts / Σ_chains numsamples, the same quantity mcmc.jl:560 forms for one chain.
// Included at the end of redclust_hip.hip (same translation unit).
//
// Two layers:
//   rc_comm_*       a communicator over `world` chains of which `n_local` live in this process (one per device).
//                   One process driving all GPUs of a node: ncclCommInitAll.  One process per GPU (torchrun, MPI ...):
//                   the caller moves the 128-byte unique id of rank 0 to the other processes and every process calls
//                   ncclCommInitRank for its chains.
//   rc_run_chains   the single-process driver: one host thread + one context per device runs rc_run_chain, then one
//                   in-place ncclAllReduce(sum, uint32) over the count buffers and rc_cocluster on the merged counts.
//
// RCCL is opened at first use (dlopen, RTLD_LOCAL) rather than linked: a process that already carries another copy of
// librccl (PyTorch bundles its own) keeps the two apart, and single-GPU users do not pay for loading it.  RC_RCCL_PATH
// overrides the library name.

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <thread>

namespace rccl_dl {

struct Api {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

static Api &api()
{
    static Api A;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {rc_env("RC_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) {
            if (!nm || !*nm) continue;
            A.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (A.handle) break;
            A.error = dlerror();
        }
        if (!A.handle) return;
        bool ok = true;
        auto sym = [&](const char *s) { void *p = dlsym(A.handle, s); if (!p) { ok = false; A.error = std::string("missing symbol ") + s; } return p; };
        A.GetUniqueId = (decltype(A.GetUniqueId))sym("ncclGetUniqueId");
        A.CommInitAll = (decltype(A.CommInitAll))sym("ncclCommInitAll");
        A.CommInitRank = (decltype(A.CommInitRank))sym("ncclCommInitRank");
        A.CommDestroy = (decltype(A.CommDestroy))sym("ncclCommDestroy");
        A.AllReduce = (decltype(A.AllReduce))sym("ncclAllReduce");
        A.GroupStart = (decltype(A.GroupStart))sym("ncclGroupStart");
        A.GroupEnd = (decltype(A.GroupEnd))sym("ncclGroupEnd");
        A.GetErrorString = (decltype(A.GetErrorString))sym("ncclGetErrorString");
        if (!ok) { dlclose(A.handle); A.handle = nullptr; }
    });
    return A;
}

}  // namespace rccl_dl

struct rc_comm {
    int n_local = 0, rank_offset = 0, world = 0;
    std::vector<int> devs;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    std::vector<long long *> ns;   // one device int64 per local chain: recorded samples, summed with the counts
};

#define NCCLCHK(expr)                                                                                                  \
    do {                                                                                                               \
        ncclResult_t r_ = (expr);                                                                                      \
        if (r_ != ncclSuccess) return fail(nullptr, RC_ERR_HIP, "%s: RCCL error: %s", #expr, rccl_dl::api().GetErrorString(r_)); \
    } while (0)
#define HIPCHK0(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) return fail(nullptr, RC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));                \
    } while (0)

extern "C" int32_t rc_comm_unique_id(uint8_t *id_out /* RC_COMM_ID_BYTES */)
{
    if (!id_out) return fail(nullptr, RC_ERR_ARG, "rc_comm_unique_id: NULL argument");
    rccl_dl::Api &A = rccl_dl::api();
    if (!A.handle) return fail(nullptr, RC_ERR_STATE, "rc_comm_unique_id: librccl could not be opened: %s", A.error.c_str());
    static_assert(sizeof(ncclUniqueId) == RC_COMM_ID_BYTES, "RC_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
    ncclUniqueId id;
    NCCLCHK(A.GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return RC_OK;
}

extern "C" int32_t rc_comm_destroy(rc_comm *cm)
{
    if (!cm) return RC_OK;
    rccl_dl::Api &A = rccl_dl::api();
    for (int i = 0; i < (int)cm->devs.size(); ++i) {
        (void)hipSetDevice(cm->devs[(size_t)i]);
        if (i < (int)cm->comms.size() && cm->comms[(size_t)i] && A.handle) A.CommDestroy(cm->comms[(size_t)i]);
        if (i < (int)cm->streams.size() && cm->streams[(size_t)i]) (void)hipStreamDestroy(cm->streams[(size_t)i]);
        if (i < (int)cm->ns.size() && cm->ns[(size_t)i]) (void)hipFree(cm->ns[(size_t)i]);
    }
    delete cm;
    return RC_OK;
}

extern "C" int32_t rc_comm_create(int32_t n_local, const int32_t *device_ids, int32_t rank_offset, int32_t world_size,
                                  const uint8_t *unique_id_or_null, rc_comm **out)
{
    if (!out || !device_ids || n_local < 1) return fail(nullptr, RC_ERR_ARG, "rc_comm_create: need n_local >= 1, device_ids and out");
    *out = nullptr;
    if (world_size < n_local || rank_offset < 0 || rank_offset + n_local > world_size)
        return fail(nullptr, RC_ERR_ARG, "rc_comm_create: ranks %d..%d do not fit a world of %d", rank_offset, rank_offset + n_local - 1, world_size);
    if (!unique_id_or_null && (rank_offset != 0 || world_size != n_local))
        return fail(nullptr, RC_ERR_ARG, "rc_comm_create: a communicator that spans processes needs the unique id of rc_comm_unique_id");
    int ndev = 0;
    HIPCHK0(hipGetDeviceCount(&ndev));
    for (int i = 0; i < n_local; ++i) {
        if (device_ids[i] < 0 || device_ids[i] >= ndev)
            return fail(nullptr, RC_ERR_ARG, "rc_comm_create: device %d does not exist (%d visible)", device_ids[i], ndev);
        // one chain per GPU: RCCL refuses duplicate devices, and k_resolve's grid barrier needs the whole chip (two
        // resolvers sharing a GPU from different contexts are serialised only within one process)
        for (int j = 0; j < i; ++j)
            if (device_ids[j] == device_ids[i]) return fail(nullptr, RC_ERR_ARG, "rc_comm_create: device %d is listed twice (one chain per GPU)", device_ids[i]);
    }
    rccl_dl::Api &A = rccl_dl::api();
    if (!A.handle) return fail(nullptr, RC_ERR_STATE, "rc_comm_create: librccl could not be opened: %s", A.error.c_str());
    rc_comm *cm = new (std::nothrow) rc_comm();
    if (!cm) return fail(nullptr, RC_ERR_OOM, "rc_comm_create: out of host memory");
    cm->n_local = n_local; cm->rank_offset = rank_offset; cm->world = world_size;
    cm->devs.assign(device_ids, device_ids + n_local);
    cm->comms.assign((size_t)n_local, nullptr);
    cm->streams.assign((size_t)n_local, nullptr);
    cm->ns.assign((size_t)n_local, nullptr);
    auto bail = [&](int32_t rc) { rc_comm_destroy(cm); return rc; };
    for (int i = 0; i < n_local; ++i) {
        if (hipSetDevice(cm->devs[(size_t)i]) != hipSuccess || hipStreamCreateWithFlags(&cm->streams[(size_t)i], hipStreamNonBlocking) != hipSuccess ||
            hipMalloc(&cm->ns[(size_t)i], sizeof(long long)) != hipSuccess)
            return bail(fail(nullptr, RC_ERR_HIP, "rc_comm_create: stream / buffer on device %d: %s", cm->devs[(size_t)i], hipGetErrorString(hipGetLastError())));
    }
    ncclResult_t r;
    if (!unique_id_or_null) {
        r = A.CommInitAll(cm->comms.data(), n_local, cm->devs.data());
    } else {
        ncclUniqueId id;
        std::memcpy(&id, unique_id_or_null, sizeof(id));
        r = A.GroupStart();
        for (int i = 0; i < n_local && r == ncclSuccess; ++i) {
            if (hipSetDevice(cm->devs[(size_t)i]) != hipSuccess) { r = ncclUnhandledCudaError; break; }
            r = A.CommInitRank(&cm->comms[(size_t)i], world_size, id, rank_offset + i);
        }
        const ncclResult_t r2 = A.GroupEnd();
        if (r == ncclSuccess) r = r2;
    }
    if (r != ncclSuccess) return bail(fail(nullptr, RC_ERR_HIP, "rc_comm_create: RCCL communicator: %s", A.GetErrorString(r)));
    *out = cm;
    return RC_OK;
}

// In-place SUM all-reduce of the co-clustering counts of the local contexts over all chains of the communicator, and of
// the numbers of recorded samples.  Afterwards every context holds the merged counts: rc_cocluster(ctx, out,
// *total_samples) is the merged posterior co-clustering matrix.  elapsed_ms (may be NULL): wall time of the collective.
extern "C" int32_t rc_comm_allreduce_counts(rc_comm *cm, rc_ctx *const *ctxs, const int64_t *num_samples, int64_t *total_samples,
                                            double *elapsed_ms)
{
    if (!cm || !ctxs || !num_samples || !total_samples) return fail(nullptr, RC_ERR_ARG, "rc_comm_allreduce_counts: NULL argument");
    rccl_dl::Api &A = rccl_dl::api();
    std::vector<void *> bufs((size_t)cm->n_local, nullptr);
    size_t count = 0;
    for (int i = 0; i < cm->n_local; ++i) {
        rc_ctx *c = ctxs[i];
        if (!c) return fail(nullptr, RC_ERR_ARG, "rc_comm_allreduce_counts: NULL context %d", i);
        if (c->dev != cm->devs[(size_t)i]) return fail(c, RC_ERR_ARG, "rc_comm_allreduce_counts: context %d lives on device %d, the communicator's chain %d on device %d", i, c->dev, i, cm->devs[(size_t)i]);
        int64_t ld = 0;
        int32_t rc = rc_cocluster_device_buffer(c, &bufs[(size_t)i], &ld);     // flushes the queued samples, drains the streams
        if (rc != RC_OK) return rc;
        const size_t cnt = (size_t)c->n * (size_t)ld;
        if (i > 0 && cnt != count) return fail(c, RC_ERR_ARG, "rc_comm_allreduce_counts: contexts of different size");
        count = cnt;
        const long long ns = num_samples[i];
        HIPCHK(c, hipMemcpy(cm->ns[(size_t)i], &ns, sizeof(ns), hipMemcpyHostToDevice));
    }
    const auto t0 = std::chrono::steady_clock::now();
    NCCLCHK(A.GroupStart());
    // inside the group nothing returns early: the first error is kept, the group is always closed (an open group would make
    // every later RCCL call of this thread — rc_comm_destroy included — misbehave or hang), and only then reported
    ncclResult_t first_nccl = ncclSuccess;
    hipError_t first_hip = hipSuccess;
    for (int i = 0; i < cm->n_local && first_nccl == ncclSuccess && first_hip == hipSuccess; ++i) {
        first_hip = hipSetDevice(cm->devs[(size_t)i]);
        if (first_hip != hipSuccess) break;
        first_nccl = A.AllReduce(bufs[(size_t)i], bufs[(size_t)i], count, ncclUint32, ncclSum, cm->comms[(size_t)i], cm->streams[(size_t)i]);
        if (first_nccl != ncclSuccess) break;
        first_nccl = A.AllReduce(cm->ns[(size_t)i], cm->ns[(size_t)i], 1, ncclInt64, ncclSum, cm->comms[(size_t)i], cm->streams[(size_t)i]);
    }
    const ncclResult_t end_nccl = A.GroupEnd();
    if (first_hip != hipSuccess) return fail(nullptr, RC_ERR_HIP, "rc_comm_allreduce_counts: hipSetDevice: %s", hipGetErrorString(first_hip));
    if (first_nccl != ncclSuccess) return fail(nullptr, RC_ERR_HIP, "rc_comm_allreduce_counts: ncclAllReduce: RCCL error: %s", A.GetErrorString(first_nccl));
    if (end_nccl != ncclSuccess) return fail(nullptr, RC_ERR_HIP, "rc_comm_allreduce_counts: ncclGroupEnd: RCCL error: %s", A.GetErrorString(end_nccl));
    for (int i = 0; i < cm->n_local; ++i) {
        HIPCHK0(hipSetDevice(cm->devs[(size_t)i]));
        HIPCHK0(hipStreamSynchronize(cm->streams[(size_t)i]));
    }
    if (elapsed_ms) *elapsed_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    long long tot = 0;
    HIPCHK0(hipSetDevice(cm->devs[0]));
    HIPCHK0(hipMemcpy(&tot, cm->ns[0], sizeof(tot), hipMemcpyDeviceToHost));
    *total_samples = tot;
    return RC_OK;
}

static int32_t run_chains_impl(int32_t n_chains, const int32_t *device_ids, const rc_chains_input *in, const rc_chain_options *opt,
                               rc_chain_outputs *outs, double *posterior_coclustering, int64_t *total_samples, double *allreduce_ms);

// Nothing may be thrown across the C boundary (a std::bad_alloc for the n² host logD at config 5, a std::system_error from
// std::thread): it would terminate the Julia / Python host.
extern "C" int32_t rc_run_chains(int32_t n_chains, const int32_t *device_ids, const rc_chains_input *in, const rc_chain_options *opt,
                                 rc_chain_outputs *outs, double *posterior_coclustering, int64_t *total_samples, double *allreduce_ms)
{
    try {
        return run_chains_impl(n_chains, device_ids, in, opt, outs, posterior_coclustering, total_samples, allreduce_ms);
    } catch (const std::bad_alloc &) {
        return fail(nullptr, RC_ERR_OOM, "rc_run_chains: out of host memory");
    } catch (const std::exception &e) {
        return fail(nullptr, RC_ERR_HIP, "rc_run_chains: %s", e.what());
    }
}

static int32_t run_chains_impl(int32_t n_chains, const int32_t *device_ids, const rc_chains_input *in, const rc_chain_options *opt,
                               rc_chain_outputs *outs, double *posterior_coclustering, int64_t *total_samples, double *allreduce_ms)
{
    if (n_chains < 1 || !device_ids || !in || !opt || !outs) return fail(nullptr, RC_ERR_ARG, "rc_run_chains: need n_chains >= 1, device_ids, input, options and outputs");
    if (!in->params || !in->init_clusts || (!in->D && !in->points)) return fail(nullptr, RC_ERR_ARG, "rc_run_chains: input needs params, init_clusts and D or points");
    if (opt->numMH > 0 && !in->D) return fail(nullptr, RC_ERR_ARG, "rc_run_chains: numMH > 0 needs the host matrix D (the split-merge scans read it)");
    // the communicator first: it validates the device list (existing, distinct) before any chain starts
    rc_comm *cm = nullptr;
    int32_t rc = rc_comm_create(n_chains, device_ids, 0, n_chains, nullptr, &cm);
    if (rc != RC_OK) return rc;
    std::vector<rc_ctx *> ctxs((size_t)n_chains, nullptr);
    std::vector<int32_t> rcs((size_t)n_chains, RC_OK);
    std::vector<std::string> errs((size_t)n_chains);
    std::vector<double> hostL;   // log.(D - Diagonal(D) + I) for the split-merge scans when the caller gave none (types.jl:155)
    const double *Lhost = in->logD_or_null;
    if (opt->numMH > 0 && !Lhost) {
        try { hostL.resize((size_t)in->n * (size_t)in->n); }
        catch (const std::bad_alloc &) { rc_comm_destroy(cm); return fail(nullptr, RC_ERR_OOM, "rc_run_chains: no host memory for the %lld x %lld logD of the split-merge scans (pass logD, or numMH = 0)", (long long)in->n, (long long)in->n); }
        host_log_matrix(in->D, in->n, hostL.data());   // once for all chains, in parallel over the host's cores
        Lhost = hostL.data();
    }
    g_chains_running += n_chains;     // all chains of this call, before any starts: each sizes its worker pool by the share of the host's cores
    struct Announce { int n; ~Announce() { g_chains_running -= n; } } announce{n_chains};
    auto worker = [&](int ci) {
        t_counted_by_driver = true;
        rc_ctx *c = nullptr;
        int32_t r = in->D ? rc_create(in->n, in->D, in->logD_or_null, in->storage_bits, device_ids[ci], in->kcap, &c)
                          : rc_create_from_points(in->n, in->dim, in->points, in->storage_bits, device_ids[ci], in->kcap, &c);
        auto err = [&](rc_ctx *cc) { errs[(size_t)ci] = rc_last_error(cc); };
        if (r != RC_OK) { rcs[(size_t)ci] = r; err(nullptr); return; }
        ctxs[(size_t)ci] = c;
        if ((r = rc_set_params(c, in->params)) != RC_OK || (r = rc_set_state(c, in->init_clusts)) != RC_OK ||
            (r = rc_cocluster_reset(c)) != RC_OK) { rcs[(size_t)ci] = r; err(c); return; }
        if (opt->numMH > 0 && (r = rc_attach_host_matrices(c, in->D, Lhost)) != RC_OK) { rcs[(size_t)ci] = r; err(c); return; }
        rc_chain_options o = *opt;
        o.seed = opt->seed + (uint64_t)ci;                       // chain seeds base, base+1, ... (bench: 1..N)
        r = rc_run_chain(c, &o, &outs[ci]);
        if (r != RC_OK) { rcs[(size_t)ci] = r; err(c); }
    };
    {
        std::vector<std::thread> th;
        th.reserve((size_t)n_chains);
        for (int ci = 0; ci < n_chains; ++ci) {
            try { th.emplace_back(worker, ci); }
            catch (const std::system_error &e) { rcs[(size_t)ci] = RC_ERR_HIP; errs[(size_t)ci] = std::string("could not start the chain's host thread: ") + e.what(); break; }
        }
        for (auto &t : th) t.join();
    }
    auto cleanup = [&]() {
        for (rc_ctx *c : ctxs) if (c) rc_destroy(c);
        rc_comm_destroy(cm);
    };
    for (int ci = 0; ci < n_chains; ++ci)
        if (rcs[(size_t)ci] != RC_OK) {
            const int32_t code = rcs[(size_t)ci];
            const std::string msg = errs[(size_t)ci];
            cleanup();
            return fail(nullptr, code, "rc_run_chains: chain %d (device %d): %s", ci, device_ids[ci], msg.c_str());
        }
    std::vector<int64_t> ns((size_t)n_chains);
    for (int ci = 0; ci < n_chains; ++ci) ns[(size_t)ci] = outs[ci].num_samples;
    int64_t tot = 0;
    rc = rc_comm_allreduce_counts(cm, ctxs.data(), ns.data(), &tot, allreduce_ms);
    if (rc == RC_OK && total_samples) *total_samples = tot;
    if (rc == RC_OK && posterior_coclustering && tot > 0) {
        rc = rc_cocluster(ctxs[0], posterior_coclustering, tot);      // Σ counts / Σ numsamples (mcmc.jl:560 over all chains)
        if (rc != RC_OK) { const std::string msg = rc_last_error(ctxs[0]); cleanup(); return fail(nullptr, rc, "rc_run_chains: %s", msg.c_str()); }
    }
    if (rc != RC_OK) { const std::string msg = rc_last_error(nullptr); cleanup(); return fail(nullptr, rc, "%s", msg.c_str()); }
    cleanup();
    return RC_OK;
}
