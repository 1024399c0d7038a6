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
// copy PyTorch already loaded (or, through MMG_RCCL_LIBRARY, is going to load) is the one that gets used.
#include "mmg_host.h"
#include "mmg_launch.h"

#include <cstdlib>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

using namespace mmg;

namespace {
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool load()
    {
        if (lib) return true;
        // MMG_RCCL_LIBRARY names the copy to use: a process must not end up with two RCCL builds (a host program that loads its own
        // later -- PyTorch's bundled one -- next to /opt/rocm's crashed in the exit handlers); mmseq_amd/_lib.py points it at PyTorch's.
        const char *wanted = getenv("MMG_RCCL_LIBRARY");
        for (const char *name : {wanted ? wanted : "librccl.so.1", "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            // RTLD_LOCAL: with RCCL's symbols in the global namespace a PyTorch imported LATER bound some of its own to them and the
            // process died in the exit handlers ("double free"), whichever RCCL build it was (tools/exit_probe.py)
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
#define RSYM(field, sym) do { *(void **)&field = dlsym(lib, sym); if (!field) { err = std::string("librccl lacks ") + sym; lib = nullptr; return false; } } while (0)
        RSYM(CommInitAll, "ncclCommInitAll");
        RSYM(CommDestroy, "ncclCommDestroy");
        RSYM(CommAbort, "ncclCommAbort");
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
    double last_enqueue_us = 0.0; // host time per device-iteration of the last run_* call (slowest driver thread)
    bool aborted = false;         // a member failed inside a run call: the communicators were aborted, the group only remains to be destroyed
    // The wire is verified ONCE per kind of exchange, the first time it carries data (verify_*): what every device holds after the
    // all-reduce must be the sum (maximum) of what the devices held before it, computed on the host from downloads.  No N > 1 run
    // of this code existed when it was written (one-GPU boxes only): the first node that has peers must not produce a wrong table
    // silently if its transport -- xGMI peer access, the RCCL build -- misbehaves.  Cost: two downloads of the exchanged vector per
    // device, once.  MMG_OPT_WIRE_CHECK = 1 runs the check in groups of one device too (tests), 2 corrupts a word behind the exchange
    // (the failure path), 0 switches it off.
    bool counts_verified = false;
    bool em_verified[3] = {false, false, false};
};

// After a failure on one device its peers may already have enqueued the collective of that iteration: their streams would wait on the
// GPU for a rank that never arrives, and every later synchronisation -- a sampler's destructor, the runtime's teardown at exit --
// with them.  ncclCommAbort makes the collectives in flight return; the group is unusable afterwards.
static void abort_group(mmg_group *g)
{
    if (g->aborted) return;
    g->aborted = true;
    for (size_t i = 0; i < g->comms.size(); ++i) {
        (void)hipSetDevice(g->devices[i]);
        if (g->comms[i]) (void)g_rccl.CommAbort(g->comms[i]);
        g->comms[i] = nullptr;
    }
}

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
    if (g->aborted) return fail(MMG_ERR_STATE, "a device of this group failed in an earlier call: its communicators were aborted, destroy the group");
    v.resize(g->devices.size());
    for (size_t i = 0; i < v.size(); ++i) {
        if (!s[i]) return fail(MMG_ERR_ARG, "NULL sampler in the group");
        int rc = sampler_view(s[i], &v[i]);
        if (rc) return rc;
        if (v[i].p->device != g->devices[i]) return fail(MMG_ERR_ARG, "sampler i must live on device i of the group");
        if (v[i].p->n != v[0].p->n || v[i].cfg.n_chains != v[0].cfg.n_chains) return fail(MMG_ERR_ARG, "samplers of a group must agree on transcripts and chains");
        if (v[i].iter != v[0].iter || v[i].cfg.gibbs_iter != v[0].cfg.gibbs_iter || v[i].cfg.trace_len != v[0].cfg.trace_len)
            return fail(MMG_ERR_ARG, "samplers of a group must be at the same iteration with the same gibbs_iter and trace_len");
    }
    return MMG_OK;
}

// One driver thread per device: thread i binds device i once and enqueues that device's kernels and collectives in order.  A
// single host thread that visits the devices in turn pays a hipSetDevice and three launches per device and iteration -- at
// config-2 scale (40 us per iteration) eight devices would wait for the host.  RCCL accepts concurrent enqueues on the
// communicators of one ncclCommInitAll from one thread each, without a group bracket.  f(i, iteration) returns an MMG code; the first
// failure (with its message) is what the caller sees, the other threads stop at their next iteration, and the communicators are
// aborted: a peer that had already enqueued the iteration's collective must not wait for the failed device for ever (abort_group).
template <typename F>
static int drive_devices(mmg_group *g, size_t G, int n_iter, F body)
{
    std::vector<int> rc(G, MMG_OK);
    std::vector<std::string> msg(G);
    std::vector<double> us(G, 0.0);
    std::atomic<bool> stop{false};
    auto run = [&](size_t i) {
        if (hipSetDevice(g->devices[i]) != hipSuccess) { rc[i] = MMG_ERR_HIP; msg[i] = "hipSetDevice"; stop = true; return; }
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < n_iter && !stop.load(std::memory_order_relaxed); ++it) {
            int r = body(i);
            if (r == MMG_OK && opt(MMG_OPT_GROUP_FAIL) >= 0 && (size_t)opt(MMG_OPT_GROUP_FAIL) % G == i && it == 1) r = fail(MMG_ERR_HIP, "injected failure (MMG_OPT_GROUP_FAIL)");
            if (r != MMG_OK) { rc[i] = r; msg[i] = mmg_last_error(); stop = true; return; }
        }
        us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    };
    if (G == 1) run(0);
    else {
        std::vector<std::thread> th;
        for (size_t i = 1; i < G; ++i) th.emplace_back(run, i);
        run(0);
        for (auto &t : th) t.join();
    }
    for (size_t i = 0; i < G; ++i) if (rc[i] != MMG_OK) { abort_group(g); return fail(rc[i], msg[i] + " (device " + std::to_string(g->devices[i]) + "; the group's communicators were aborted)"); }
    g->last_enqueue_us = n_iter > 0 ? *std::max_element(us.begin(), us.end()) / n_iter : 0.0;
    return MMG_OK;
}

// sum over the devices of `count` elements: send[i] -> recv[i] (may be the same buffer), each on its sampler's stream; one thread
// enqueues for every device, hence the group bracket -- which is closed on the error path as well
static int all_reduce(const mmg_group *g, const std::vector<SamplerView> &v, void *const *send, void *const *recv, size_t count, ncclDataType_t type)
{
    NCCL_TRY(g_rccl.GroupStart());
    int rc = MMG_OK;
    for (size_t i = 0; i < v.size() && rc == MMG_OK; ++i) {
        if (hipSetDevice(g->devices[i]) != hipSuccess) { rc = fail(MMG_ERR_HIP, "hipSetDevice"); break; }
        const ncclResult_t r = g_rccl.AllReduce(send[i], recv[i], count, type, ncclSum, g->comms[i], v[i].stream);
        if (r != ncclSuccess) rc = fail(MMG_ERR_HIP, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
    }
    const ncclResult_t e = g_rccl.GroupEnd();
    if (rc == MMG_OK && e != ncclSuccess) rc = fail(MMG_ERR_HIP, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(e));
    return rc;
}

// The first iteration of a sharded chain with its count exchange verified: K1 on every device, the devices' own column sums
// downloaded and added up on the host, the all-reduce, every device's result compared with that sum (and the sum's total with the
// reads of all shards: src/mmseq.cpp:896-899 -- every read is counted once), then K2.  Any difference fails loudly and aborts the
// group; nothing of the iteration has reached a trace by then.
static int verified_first_iteration(mmg_group *g, mmg_sampler *const *samplers, const std::vector<SamplerView> &v, const std::vector<void *> &cnt, size_t count)
{
    const size_t G = v.size();
    auto bail = [&](int code, const std::string &why) { abort_group(g); return fail(code, why); };
    std::vector<int64_t> want(count, 0);
    std::vector<int32_t> buf(count);
    for (size_t i = 0; i < G; ++i) {
        int rc = mmg_sampler_sample(samplers[i]);
        if (rc != MMG_OK) { const std::string why = mmg_last_error(); return bail(rc, why); }
    }
    for (size_t i = 0; i < G; ++i) {
        hipError_t e = hipSetDevice(g->devices[i]);
        if (e == hipSuccess) e = hipStreamSynchronize(v[i].stream);
        if (e == hipSuccess) e = hipMemcpy(buf.data(), cnt[i], count * sizeof(int32_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return bail(MMG_ERR_HIP, std::string("wire check (counts before the exchange): ") + hipGetErrorString(e));
        for (size_t j = 0; j < count; ++j) want[j] += buf[j];
    }
    int64_t total = 0, reads = 0;
    for (size_t j = 0; j < count; ++j) total += want[j];
    for (size_t i = 0; i < G; ++i) reads += (int64_t)v[i].p->total_k_hit * v[i].cfg.n_chains; // (the reads of rows WITH hits: an empty row is allocated nowhere)
    if (total != reads)
        return bail(MMG_ERR_STATE, "wire check: the shards' first sweep allocated " + std::to_string(total) + " reads, the shards hold " + std::to_string(reads));
    int rc = all_reduce(g, v, cnt.data(), cnt.data(), count, ncclInt32);
    if (rc != MMG_OK) { const std::string why = mmg_last_error(); return bail(rc, why); }
    for (size_t i = 0; i < G; ++i) {
        hipError_t e = hipSetDevice(g->devices[i]);
        if (e == hipSuccess) e = hipStreamSynchronize(v[i].stream);
        if (e == hipSuccess && opt(MMG_OPT_WIRE_CHECK) == 2 && i + 1 == G) { const int32_t bad = -1; e = hipMemcpy((int32_t *)cnt[i] + count / 2, &bad, 4, hipMemcpyHostToDevice); }
        if (e == hipSuccess) e = hipMemcpy(buf.data(), cnt[i], count * sizeof(int32_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return bail(MMG_ERR_HIP, std::string("wire check (counts behind the exchange): ") + hipGetErrorString(e));
        for (size_t j = 0; j < count; ++j)
            if ((int64_t)buf[j] != want[j])
                return bail(MMG_ERR_STATE, "wire check: after the count all-reduce device " + std::to_string(g->devices[i]) + " holds " + std::to_string(buf[j]) +
                            " at element " + std::to_string(j) + " where the devices' own counts add up to " + std::to_string(want[j]) +
                            ": the exchange between the devices (RCCL) does not deliver the sum; nothing was written");
    }
    for (size_t i = 0; i < G; ++i) {
        rc = mmg_sampler_update(samplers[i]);
        if (rc != MMG_OK) { const std::string why = mmg_last_error(); return bail(rc, why); }
    }
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
    if (n_iter > 0 && !g->counts_verified && opt(MMG_OPT_WIRE_CHECK) != 0 && (G > 1 || opt(MMG_OPT_WIRE_CHECK) >= 1)) {
        // the group's first sharded iteration, step by step on this thread, with the exchange checked against the host's own sum
        if ((rc = verified_first_iteration(g, samplers, v, cnt, (size_t)count)) != MMG_OK) return rc;
        g->counts_verified = true;
        --n_iter;
    }
    return drive_devices(g, G, n_iter, [&](size_t i) -> int {
        int r = mmg_sampler_sample(samplers[i]);                                                          // src/mmseq.cpp:857-891 on the device's rows
        if (r != MMG_OK) return r;
        if (G > 1) {                                                                                      // :896-899 across devices
            const ncclResult_t e = g_rccl.AllReduce(cnt[i], cnt[i], (size_t)count, ncclInt32, ncclSum, g->comms[i], v[i].stream);
            if (e != ncclSuccess) return fail(MMG_ERR_HIP, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(e));
        }
        return mmg_sampler_update(samplers[i]);                                                           // :905-917, identical everywhere
    });
}

extern "C" int mmg_group_run_chains(mmg_group *g, mmg_sampler *const *samplers, int n_iter)
{
    std::vector<SamplerView> v;
    int rc = check_samplers(g, samplers, v);
    if (rc) return rc;
    if (n_iter < 0) return fail(MMG_ERR_ARG, "bad argument");
    return drive_devices(g, v.size(), n_iter, [&](size_t i) -> int { return mmg_sampler_run(samplers[i], 1); });
}

extern "C" int mmg_group_enqueue_us(const mmg_group *g, double *us_per_device_iteration)
{
    if (!g || !us_per_device_iteration) return fail(MMG_ERR_ARG, "NULL argument");
    *us_per_device_iteration = g->last_enqueue_us;
    return MMG_OK;
}

extern "C" int mmg_group_pool_moments(mmg_group *g, mmg_sampler *const *samplers, double *sum_log, double *sum_log2, int64_t *n_samples)
{
    std::vector<SamplerView> v;
    int rc = check_samplers(g, samplers, v);
    if (rc) return rc;
    const size_t G = v.size();
    for (size_t i = 1; i < G; ++i)
        if (v[i].n_kept != v[0].n_kept) return fail(MMG_ERR_ARG, "samplers of a group must have kept the same number of samples");
    std::vector<void *> mom(G), red(G, nullptr);
    uint64_t count = 0;
    for (size_t i = 0; i < G; ++i) if ((rc = mmg_sampler_moments_devptr(samplers[i], &mom[i], &count)) != MMG_OK) return rc;
    // reduced into scratch buffers: the samplers keep their own moments, so pooling twice -- or pooling, running on and pooling
    // again -- counts nothing twice
    auto release = [&]() { for (size_t i = 0; i < G; ++i) if (red[i]) { (void)hipSetDevice(g->devices[i]); (void)hipFree(red[i]); } };
    const uint32_t n = v[0].p->n;
    const int C = v[0].cfg.n_chains;
    std::vector<double> pooled((size_t)count);
    if (G > 1) {
        for (size_t i = 0; i < G; ++i) {
            if (hipSetDevice(g->devices[i]) != hipSuccess || hipMalloc(&red[i], (size_t)count * sizeof(double)) != hipSuccess) { release(); return fail(MMG_ERR_HIP, "hipMalloc (pooled moments)"); }
        }
        if ((rc = all_reduce(g, v, mom.data(), red.data(), (size_t)count, ncclDouble)) != MMG_OK) { const std::string why = mmg_last_error(); abort_group(g); release(); return fail(rc, why); }
        hipError_t e = hipSetDevice(g->devices[0]);
        if (e == hipSuccess) e = hipStreamSynchronize(v[0].stream);
        if (e == hipSuccess) e = hipMemcpy(pooled.data(), red[0], (size_t)count * sizeof(double), hipMemcpyDeviceToHost);
        for (size_t i = 1; i < G && e == hipSuccess; ++i) { e = hipSetDevice(g->devices[i]); if (e == hipSuccess) e = hipStreamSynchronize(v[i].stream); }
        release();
        if (e != hipSuccess) return fail(MMG_ERR_HIP, std::string("pooled moments: ") + hipGetErrorString(e));
    } else {
        hipError_t e = hipSetDevice(g->devices[0]);
        if (e == hipSuccess) e = hipStreamSynchronize(v[0].stream);
        if (e == hipSuccess) e = hipMemcpy(pooled.data(), mom[0], (size_t)count * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return fail(MMG_ERR_HIP, std::string("pooled moments: ") + hipGetErrorString(e));
    }
    // pooled: [2][C][n] in device numbering, summed over devices; the chains of a device are added up here, in the caller's numbering
    std::vector<double> a(n), tmp(n);
    for (int w = 0; w < 2; ++w) {
        double *out = w == 0 ? sum_log : sum_log2;
        if (!out) continue;
        std::fill(a.begin(), a.end(), 0.0);
        for (int c = 0; c < C; ++c) {
            const double *src = pooled.data() + ((size_t)w * C + (size_t)c) * n;
            for (uint32_t t = 0; t < n; ++t) a[t] += src[t];
        }
        to_ext(v[0].p, a, out);
    }
    if (n_samples) *n_samples = v[0].n_kept * (int64_t)C * (int64_t)G;
    return MMG_OK;
}

// ---- EM over read shards: one member per device, xe / accumulators / column counts exchanged with RCCL between the phases
namespace {
struct EmGroupCtx { mmg_group *g; };
std::function<int(int)> make_rccl_reduce(const std::vector<mmg_em *> &es, void *ctx)
{
    mmg_group *g = ((EmGroupCtx *)ctx)->g;
    return [es, g](int what) -> int {
        // the first exchange of every kind is verified against the host's own reduction of the members' buffers (mmg_group: wire check)
        const bool verify = what >= 0 && what < 3 && !g->em_verified[what] && opt(MMG_OPT_WIRE_CHECK) != 0 && (es.size() > 1 || opt(MMG_OPT_WIRE_CHECK) >= 1);
        std::vector<uint64_t> want;
        auto fetch = [&](size_t i, std::vector<uint64_t> &out) -> hipError_t { // member i's buffer as 64-bit words (int32 sign-extended)
            void *p = nullptr;
            size_t cnt = 0;
            em_exchange_buffers(es[i], what, &p, &cnt);
            hipError_t e = hipSetDevice(g->devices[i]);
            if (e == hipSuccess) e = hipDeviceSynchronize();
            out.resize(cnt);
            if (what == 0) {
                std::vector<int32_t> t(cnt);
                if (e == hipSuccess) e = hipMemcpy(t.data(), p, cnt * 4, hipMemcpyDeviceToHost);
                for (size_t j = 0; j < cnt; ++j) out[j] = (uint64_t)(int64_t)t[j];
            } else if (e == hipSuccess) e = hipMemcpy(out.data(), p, cnt * 8, hipMemcpyDeviceToHost);
            return e;
        };
        if (verify) {
            std::vector<uint64_t> b;
            for (size_t i = 0; i < es.size(); ++i) {
                const hipError_t e = fetch(i, b);
                if (e != hipSuccess) { abort_group(g); return fail(MMG_ERR_HIP, std::string("wire check (EM buffers before the exchange): ") + hipGetErrorString(e)); }
                if (i == 0) want = b;
                else for (size_t j = 0; j < b.size(); ++j) want[j] = what == 0 ? (uint64_t)std::max((int64_t)want[j], (int64_t)b[j]) : want[j] + b[j];
            }
        }
        NCCL_TRY(g_rccl.GroupStart());
        int rc = MMG_OK;
        for (size_t i = 0; i < es.size() && rc == MMG_OK; ++i) {
            void *p = nullptr;
            size_t cnt = 0;
            em_exchange_buffers(es[i], what, &p, &cnt);
            if (hipSetDevice(g->devices[i]) != hipSuccess) { rc = fail(MMG_ERR_HIP, "hipSetDevice"); break; }
            const ncclResult_t r = g_rccl.AllReduce(p, p, cnt, what == 0 ? ncclInt32 : ncclUint64, what == 0 ? ncclMax : ncclSum, g->comms[i], (hipStream_t)0);
            if (r != ncclSuccess) rc = fail(MMG_ERR_HIP, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
        }
        const ncclResult_t e = g_rccl.GroupEnd();
        if (rc == MMG_OK && e != ncclSuccess) rc = fail(MMG_ERR_HIP, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(e));
        if (rc != MMG_OK) { const std::string why = mmg_last_error(); abort_group(g); return fail(rc, why); } // members that did enqueue must not wait for ever
        if (verify) {
            std::vector<uint64_t> b;
            for (size_t i = 0; i < es.size(); ++i) {
                hipError_t e = hipSuccess;
                if (opt(MMG_OPT_WIRE_CHECK) == 2 && what == 1 && i + 1 == es.size()) { // (failure path: a word of the accumulators is damaged behind the exchange)
                    void *p = nullptr; size_t cnt = 0;
                    em_exchange_buffers(es[i], what, &p, &cnt);
                    const uint64_t bad = ~0ull;
                    e = hipSetDevice(g->devices[i]);
                    if (e == hipSuccess) e = hipDeviceSynchronize();
                    if (e == hipSuccess) e = hipMemcpy((uint64_t *)p + cnt / 2, &bad, 8, hipMemcpyHostToDevice);
                }
                if (e == hipSuccess) e = fetch(i, b);
                if (e != hipSuccess) { abort_group(g); return fail(MMG_ERR_HIP, std::string("wire check (EM buffers behind the exchange): ") + hipGetErrorString(e)); }
                for (size_t j = 0; j < b.size(); ++j)
                    if (b[j] != want[j]) {
                        abort_group(g);
                        return fail(MMG_ERR_STATE, std::string("wire check: after the EM exchange of ") + (what == 0 ? "exponents (max)" : what == 1 ? "accumulators (sum)" : "column counts (sum)") +
                                    " device " + std::to_string(g->devices[i]) + " holds " + std::to_string(b[j]) + " at word " + std::to_string(j) + " where the members' own buffers give " +
                                    std::to_string(want[j]) + ": the exchange between the devices (RCCL) does not deliver the reduction");
                    }
            }
            g->em_verified[what] = true;
        }
        return rc;
    };
}
} // namespace

extern "C" int mmg_group_em_create(mmg_group *g, const mmg_problem *const *shards, const double *mu0, mmg_em **ems, double *loglik0)
{
    if (!g || !shards || !mu0 || !ems) return fail(MMG_ERR_ARG, "NULL argument");
    if (g->aborted) return fail(MMG_ERR_STATE, "a device of this group failed in an earlier call: its communicators were aborted, destroy the group");
    for (size_t i = 0; i < g->devices.size(); ++i)
        if (!shards[i] || shards[i]->device != g->devices[i]) return fail(MMG_ERR_ARG, "shard i must live on device i of the group");
    EmGroupCtx ctx{g};
    return em_create_sharded(shards, (int)g->devices.size(), mu0, make_rccl_reduce, &ctx, ems, loglik0);
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
