// group.hip -- several GPUs of one node driven from ONE process: RCCL over xGMI behind the C ABI (mmg_group_*).
//
// The reference's only parallelism is OpenMP inside one process (src/mmseq.cpp:834-838, :864).  Here the unit is a device:
//   read-shard mode   every device holds a contiguous range of the stored rows; per iteration K1 on every device, one
//                     ncclAllReduce(int32, sum) of the count vectors (0.8 MB at 200 k transcripts) in place, then the identical
//                     K2 everywhere (same Philox key => same mu, no broadcast).  Integer sums: the sharded chain is
//                     bit-identical to the chain of the unsharded problem.
//   chains mode       every device runs its own chains over the full matrix; nothing is exchanged until the end, when ONE
//                     ncclAllReduce(fp64, sum) pools the posterior moments.
// One communicator per device from ncclCommInitAll, every collective enqueued on the sampler's own stream inside a
// ncclGroupStart / ncclGroupEnd bracket, so kernels and collectives of a device stay ordered without host synchronisation.
// RCCL is loaded on first use (dlopen): a process that never forms a group does not need it, and inside a PyTorch process the
// copy PyTorch already loaded is the one that gets used.
#include "mmg_host.h"
#include "mmg_launch.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <vector>

using namespace mmg;

namespace {
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool load()
    {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
#define RSYM(field, sym) do { *(void **)&field = dlsym(lib, sym); if (!field) { err = std::string("librccl lacks ") + sym; lib = nullptr; return false; } } while (0)
        RSYM(CommInitAll, "ncclCommInitAll");
        RSYM(CommDestroy, "ncclCommDestroy");
        RSYM(AllReduce, "ncclAllReduce");
        RSYM(GroupStart, "ncclGroupStart");
        RSYM(GroupEnd, "ncclGroupEnd");
        RSYM(GetErrorString, "ncclGetErrorString");
#undef RSYM
        return true;
    }
};
Rccl g_rccl;
} // namespace

struct mmg_group {
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
};

#define NCCL_TRY(expr)                                                                                                   \
    do {                                                                                                                 \
        ncclResult_t _r = (expr);                                                                                        \
        if (_r != ncclSuccess) return fail(MMG_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(_r));          \
    } while (0)

extern "C" int mmg_group_create(const int *devices, int n, mmg_group **out)
{
    if (!devices || !out || n < 1) return fail(MMG_ERR_ARG, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return fail(MMG_ERR_NO_DEVICE, "no HIP device available: libmmgibbs has no CPU fallback"); }
    for (int i = 0; i < n; ++i) {
        if (devices[i] < 0 || devices[i] >= ndev) return fail(MMG_ERR_ARG, "device index out of range");
        for (int j = 0; j < i; ++j) if (devices[j] == devices[i]) return fail(MMG_ERR_ARG, "a device may appear once in a group");
    }
    if (!g_rccl.load()) return fail(MMG_ERR_STATE, g_rccl.err);
    mmg_group *g = new mmg_group();
    g->devices.assign(devices, devices + n);
    g->comms.resize(n);
    ncclResult_t r = g_rccl.CommInitAll(g->comms.data(), n, devices);
    if (r != ncclSuccess) { delete g; return fail(MMG_ERR_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r)); }
    *out = g;
    return MMG_OK;
}

extern "C" int mmg_group_size(const mmg_group *g, int *n)
{
    if (!g || !n) return fail(MMG_ERR_ARG, "NULL argument");
    *n = (int)g->devices.size();
    return MMG_OK;
}

extern "C" void mmg_group_destroy(mmg_group *g)
{
    if (!g) return;
    for (size_t i = 0; i < g->comms.size(); ++i) { (void)hipSetDevice(g->devices[i]); if (g->comms[i]) (void)g_rccl.CommDestroy(g->comms[i]); }
    delete g;
}

// every sampler on its group device, all with the same transcript count
static int check_samplers(const mmg_group *g, mmg_sampler *const *s, std::vector<SamplerView> &v)
{
    if (!g || !s) return fail(MMG_ERR_ARG, "NULL argument");
    v.resize(g->devices.size());
    for (size_t i = 0; i < v.size(); ++i) {
        if (!s[i]) return fail(MMG_ERR_ARG, "NULL sampler in the group");
        int rc = sampler_view(s[i], &v[i]);
        if (rc) return rc;
        if (v[i].p->device != g->devices[i]) return fail(MMG_ERR_ARG, "sampler i must live on device i of the group");
        if (v[i].p->n != v[0].p->n || v[i].cfg.n_chains != v[0].cfg.n_chains) return fail(MMG_ERR_ARG, "samplers of a group must agree on transcripts and chains");
    }
    return MMG_OK;
}

// in-place sum over the devices of `count` elements at ptrs[i], each on its sampler's stream
static int all_reduce(const mmg_group *g, const std::vector<SamplerView> &v, void *const *ptrs, size_t count, ncclDataType_t type)
{
    NCCL_TRY(g_rccl.GroupStart());
    for (size_t i = 0; i < v.size(); ++i) {
        HIP_TRY(hipSetDevice(g->devices[i]));
        NCCL_TRY(g_rccl.AllReduce(ptrs[i], ptrs[i], count, type, ncclSum, g->comms[i], v[i].stream));
    }
    NCCL_TRY(g_rccl.GroupEnd());
    return MMG_OK;
}

extern "C" int mmg_group_run_sharded(mmg_group *g, mmg_sampler *const *samplers, int n_iter)
{
    std::vector<SamplerView> v;
    int rc = check_samplers(g, samplers, v);
    if (rc) return rc;
    if (n_iter < 0) return fail(MMG_ERR_ARG, "bad argument");
    const size_t G = v.size();
    for (size_t i = 1; i < G; ++i)
        if (v[i].cfg.seed != v[0].cfg.seed || v[i].cfg.chain_base != v[0].cfg.chain_base || v[i].iter != v[0].iter)
            return fail(MMG_ERR_ARG, "read shards of one chain need the same seed, chain_base and iteration on every device");
    std::vector<void *> cnt(G);
    uint64_t count = 0;
    for (size_t i = 0; i < G; ++i) if ((rc = mmg_sampler_counts_devptr(samplers[i], &cnt[i], &count)) != MMG_OK) return rc;
    for (int it = 0; it < n_iter; ++it) {
        for (size_t i = 0; i < G; ++i) if ((rc = mmg_sampler_sample(samplers[i])) != MMG_OK) return rc;      // src/mmseq.cpp:857-891 on the device's rows
        if (G > 1 && (rc = all_reduce(g, v, cnt.data(), (size_t)count, ncclInt32)) != MMG_OK) return rc;      // :896-899 across devices
        for (size_t i = 0; i < G; ++i) if ((rc = mmg_sampler_update(samplers[i])) != MMG_OK) return rc;      // :905-917, identical everywhere
    }
    return MMG_OK;
}

extern "C" int mmg_group_run_chains(mmg_group *g, mmg_sampler *const *samplers, int n_iter)
{
    std::vector<SamplerView> v;
    int rc = check_samplers(g, samplers, v);
    if (rc) return rc;
    if (n_iter < 0) return fail(MMG_ERR_ARG, "bad argument");
    // interleaved so that every device has work queued from the first iteration on
    for (int it = 0; it < n_iter; ++it)
        for (size_t i = 0; i < v.size(); ++i) if ((rc = mmg_sampler_run(samplers[i], 1)) != MMG_OK) return rc;
    return MMG_OK;
}

extern "C" int mmg_group_pool_moments(mmg_group *g, mmg_sampler *const *samplers, double *sum_log, double *sum_log2, int64_t *n_samples)
{
    std::vector<SamplerView> v;
    int rc = check_samplers(g, samplers, v);
    if (rc) return rc;
    const size_t G = v.size();
    std::vector<void *> mom(G);
    uint64_t count = 0;
    for (size_t i = 0; i < G; ++i) if ((rc = mmg_sampler_moments_devptr(samplers[i], &mom[i], &count)) != MMG_OK) return rc;
    if (G > 1 && (rc = all_reduce(g, v, mom.data(), (size_t)count, ncclDouble)) != MMG_OK) return rc;
    // every device now holds the sums over devices; chains of a device are added up on the host
    const uint32_t n = v[0].p->n;
    const int C = v[0].cfg.n_chains;
    std::vector<double> a(n), b(n);
    int64_t ns = 0, total = 0;
    if (sum_log) std::fill(sum_log, sum_log + n, 0.0);
    if (sum_log2) std::fill(sum_log2, sum_log2 + n, 0.0);
    for (int c = 0; c < C; ++c) {
        if ((rc = mmg_sampler_get_moments(samplers[0], c, a.data(), b.data(), &ns)) != MMG_OK) return rc;
        for (uint32_t t = 0; t < n; ++t) { if (sum_log) sum_log[t] += a[t]; if (sum_log2) sum_log2[t] += b[t]; }
        total += ns * (int64_t)G;
    }
    if (n_samples) *n_samples = total;
    return MMG_OK;
}

// Contiguous row ranges of (nearly) equal hit counts: bounds[i] = first row of part i, bounds[parts] = m.  Boundaries are even
// row indices (a Philox block serves the rows 2q and 2q+1: shards that start on even rows keep every block on one device).
extern "C" int mmg_shard_bounds(const uint64_t *row_ptr, uint64_t m, int parts, uint64_t *bounds)
{
    if (!row_ptr || !bounds || parts < 1) return fail(MMG_ERR_ARG, "bad argument");
    const uint64_t nnz = row_ptr[m];
    bounds[0] = 0;
    for (int i = 1; i < parts; ++i) {
        const uint64_t target = (uint64_t)(((unsigned __int128)nnz * (uint64_t)i) / (uint64_t)parts);
        uint64_t r = (uint64_t)(std::lower_bound(row_ptr, row_ptr + m + 1, target) - row_ptr);
        r = std::min<uint64_t>(m, (r + 1) & ~(uint64_t)1);
        bounds[i] = std::max<uint64_t>(r, bounds[i - 1]);
    }
    bounds[parts] = m;
    return MMG_OK;
}
