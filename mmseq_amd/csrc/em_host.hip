// em_host.hip -- host side of the mmg_em_* entry points: EM sweeps on the device (src/mmseq.cpp:741-811).
#include "mmg_host.h"
#include "mmg_launch.h"

#include <algorithm>
#include <functional>

using namespace mmg;

struct mmg_em {
    mmg_problem *p = nullptr;
    int device = 0;
    double *d_mu = nullptr, *d_pc = nullptr;
    uint32_t *d_word = nullptr;
    uint64_t *d_acc = nullptr;                        // hi[n] | lo[n] | ll[4] in one buffer: one all-reduce per pass when sharded
    uint64_t *d_hi = nullptr, *d_lo = nullptr, *d_ll = nullptr;
    uint64_t *d_colcnt = nullptr;                     // hits per transcript over ALL shards (owned when sharded; else the problem's cache)
    bool owns_colcnt = false;
    // read shards of one problem (mmg_group_em_*): the members advance together, exchanging xe (max), the accumulators and the
    // log-likelihood limbs (exact integer sums): the sharded EM equals the unsharded EM bit for bit
    std::vector<mmg_em *> peers;                      // non-empty on the leader only; peers[0] == this
    std::function<int(int)> reduce;                   // 0: xe (max, int32, n)  1: acc (sum, uint64, 2 n + 3)  2: colcnt (sum, uint64, n)
    int32_t *d_xe = nullptr, *d_sexp = nullptr;
    EmOut *d_out = nullptr;
    uint64_t *d_chunk[2] = {nullptr, nullptr}; // tile ranges of the accumulate / measure kernels
    int grid[2] = {0, 0};
    int path = 0;   // rows-pass kernel: 2 sliced-ELL stream, 0 row per thread from the CSR
    bool first = true;
    int sweeps = 0, repeats = 0;
    double loglik = 0.0;
};

static void em_free(mmg_em *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    for (void *x : {(void *)e->d_mu, (void *)e->d_pc, (void *)e->d_word, (void *)e->d_acc, (void *)(e->owns_colcnt ? e->d_colcnt : nullptr),
                    (void *)e->d_xe, (void *)e->d_sexp, (void *)e->d_out, (void *)e->d_chunk[0], (void *)e->d_chunk[1]})
        if (x) (void)hipFree(x);
    delete e;
}

static int em_launch_rows(mmg_em *e, bool measure)
{
    mmg_problem *p = e->p;
    EmArgs a;
    a.n = p->n; a.mu = e->d_mu; a.word = e->d_word; a.hi = e->d_hi; a.lo = e->d_lo; a.xe = e->d_xe; a.ll = e->d_ll;
    if (p->m == 0) return MMG_OK;
    if (e->path == 2) {
        const int w = measure ? 1 : 0;
        const void *fn = em_sell_kernel(p->idx64, p->d_k != nullptr, measure);
        const void *rp = p->d_row_ptr;
        const uint32_t *col = p->d_col, *kk = p->d_k;
        const SellTile *tiles = p->d_sell_tiles;
        const uint64_t *chunk = e->d_chunk[w];
        const uint8_t *stream = p->d_sell;
        void *args[] = {(void *)&rp, (void *)&col, (void *)&kk, (void *)&tiles, (void *)&chunk, (void *)&stream, (void *)&a};
        HIP_TRY(hipLaunchKernel(fn, dim3((unsigned)e->grid[w]), dim3(EM_SELL_BS), args, 0, 0));
        return MMG_OK;
    }
    launch_em_rows_global(p->idx64, measure, p->d_row_ptr, p->d_col, p->d_k, p->m, a, 0);
    HIP_TRY(hipGetLastError());
    return MMG_OK;
}

// One validated rows pass for the current mu: accumulators, log-likelihood.  Carried exponents first (unless this is the first
// pass), repeated on measured exponents if a check failed.  With read shards every phase runs on every member before the exchange
// that follows it; the sums are integers, so every member then holds the bits the unsharded problem would have produced, takes
// the same decisions and applies the same update.
static int em_rows_pass(mmg_em *lead)
{
    std::vector<mmg_em *> one(1, lead);
    const std::vector<mmg_em *> &es = lead->peers.empty() ? one : lead->peers;
    const uint32_t n = lead->p->n;
    const unsigned gn = (n + 255) / 256;
    for (int measured = lead->first ? 1 : 0; measured < 2; ++measured) {
        if (measured) {
            for (mmg_em *e : es) {
                HIP_TRY(hipSetDevice(e->device));
                launch_fill_i32(e->d_xe, n, INT32_MIN, 0);
                int rc = em_launch_rows(e, true);
                if (rc) return rc;
            }
            if (lead->reduce) { int rc = lead->reduce(0); if (rc) return rc; }
        }
        for (mmg_em *e : es) {
            HIP_TRY(hipSetDevice(e->device));
            launch_em_prepare(n, e->d_mu, e->p->d_l, e->d_colcnt, measured ? e->d_xe : e->d_sexp, measured, e->d_word, e->d_hi, e->d_lo,
                              e->d_pc, e->d_ll, e->p->d_int_of_ext, 0);
            int rc = em_launch_rows(e, false);
            if (rc) return rc;
        }
        if (lead->reduce) { int rc = lead->reduce(1); if (rc) return rc; }
        EmOut out0{};
        for (size_t i = 0; i < es.size(); ++i) {
            mmg_em *e = es[i];
            HIP_TRY(hipSetDevice(e->device));
            if (!measured) launch_em_check(n, e->d_word, e->d_hi, e->d_ll, 0);
            launch_em_finish(e->d_pc, gn, e->d_ll, e->d_out, 0);
            EmOut out;
            HIP_TRY(hipMemcpy(&out, e->d_out, sizeof(out), hipMemcpyDeviceToHost));
            if (i == 0) out0 = out;
            else if (out.loglik != out0.loglik || (out.flag != 0) != (out0.flag != 0)) return fail(MMG_ERR_STATE, "EM: the shards disagree after the exchange");
            e->loglik = out.loglik;
        }
        if (!out0.flag) break;
        if (measured) return fail(MMG_ERR_STATE, "EM: a measured pass failed its own check");
        for (mmg_em *e : es) ++e->repeats;
    }
    for (mmg_em *e : es) e->first = false;
    return MMG_OK;
}

// device state of one member (no pass yet)
static int em_alloc(const mmg_problem *cp, const double *mu0, mmg_em **out)
{
    mmg_problem *p = const_cast<mmg_problem *>(cp); // the lazily built column counts are a cache
    HIP_TRY(hipSetDevice(p->device));
    if (!p->d_colcnt) {
        HIP_TRY(hipMalloc((void **)&p->d_colcnt, p->n * sizeof(uint64_t)));
        HIP_TRY(hipMemset(p->d_colcnt, 0, p->n * sizeof(uint64_t)));
        if (p->nnz) {
            const unsigned g = (unsigned)std::min<uint64_t>((p->nnz + 255) / 256, (uint64_t)p->cu_count * 32);
            launch_em_colcount(p->d_col, p->nnz, p->d_colcnt, g, 0);
            HIP_TRY(hipGetLastError());
        }
        p->device_bytes += p->n * 8;
    }
    mmg_em *e = new mmg_em();
    e->p = p;
    e->device = p->device;
    e->d_colcnt = p->d_colcnt;
    const unsigned gn = (p->n + 255) / 256;
#define EM_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { em_free(e); return fail(MMG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } } while (0)
    EM_TRY(hipMalloc((void **)&e->d_mu, p->n * sizeof(double)));
    EM_TRY(hipMalloc((void **)&e->d_pc, gn * sizeof(double)));
    EM_TRY(hipMalloc((void **)&e->d_word, p->n * sizeof(uint32_t)));
    EM_TRY(hipMalloc((void **)&e->d_acc, (2 * (size_t)p->n + 4) * sizeof(uint64_t)));
    e->d_hi = e->d_acc; e->d_lo = e->d_acc + p->n; e->d_ll = e->d_acc + 2 * (size_t)p->n;
    EM_TRY(hipMalloc((void **)&e->d_xe, p->n * sizeof(int32_t)));
    EM_TRY(hipMalloc((void **)&e->d_sexp, p->n * sizeof(int32_t)));
    EM_TRY(hipMalloc((void **)&e->d_out, sizeof(EmOut)));
    {
        std::vector<double> mu_int;
        to_int(p, mu0, mu_int);
        EM_TRY(hipMemcpy(e->d_mu, mu_int.data(), p->n * sizeof(double), hipMemcpyHostToDevice));
    }
    e->path = (p->use_sell && p->n_sell_tiles > 0) ? 2 : 0;
    if (opt(MMG_OPT_EM_KERNEL) == 0) e->path = 0;
    if (e->path == 2) {
        const uint64_t n_tiles = p->n_sell_tiles;
        for (int w = 0; w < 2; ++w) {
            const void *fn = em_sell_kernel(p->idx64, p->d_k != nullptr, w == 1);
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, EM_SELL_BS, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 4; }
            if (per_cu > 32) per_cu = 32;
            uint64_t grid = std::max<uint64_t>(1, std::min<uint64_t>(n_tiles, (uint64_t)p->cu_count * per_cu));
            uint64_t resident = 0; // > 0: the last generation's ranges are halved (weighted_chunks_tapered)
            if (opt(MMG_OPT_EM_GRID) >= 1) {
                if ((uint64_t)opt(MMG_OPT_EM_GRID) < grid) grid = (uint64_t)opt(MMG_OPT_EM_GRID); // tests: long tile ranges on small problems
            } else {
                // several generations of workgroups once ranges are long (about 24 tiles per wave: mmgibbs.hip, problem_build_sell):
                // config 3 has 381 tiles per range in one generation, 48 in eight: 1.81 -> 1.52 ms per sweep
                const uint64_t g = (n_tiles + grid * 24) / (grid * 48);
                resident = grid;
                grid = std::min<uint64_t>(n_tiles, grid * std::min<uint64_t>(16, std::max<uint64_t>(1, g)));
            }
            std::vector<uint64_t> chunk(grid + 1);
            if (p->h_sell_cum.size() == n_tiles + 1) { weighted_chunks_tapered(p->h_sell_cum, grid, resident, chunk); grid = chunk.size() - 1; }
            else for (uint64_t c = 0; c <= grid; ++c) chunk[c] = (uint64_t)(((unsigned __int128)n_tiles * c) / grid);
            EM_TRY(hipMalloc((void **)&e->d_chunk[w], chunk.size() * sizeof(uint64_t)));
            EM_TRY(hipMemcpy(e->d_chunk[w], chunk.data(), chunk.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
            e->grid[w] = (int)grid;
        }
    }
#undef EM_TRY
    *out = e;
    return MMG_OK;
}

extern "C" int mmg_em_create(const mmg_problem *cp, const double *mu0, mmg_em **out, double *loglik0)
{
    if (!cp || !mu0 || !out) return fail(MMG_ERR_ARG, "NULL argument");
    mmg_em *e = nullptr;
    int rc = em_alloc(cp, mu0, &e);
    if (rc) return rc;
    // log-likelihood of the start value (src/mmseq.cpp:745-754)
    rc = em_rows_pass(e);
    if (rc) { em_free(e); return rc; }
    if (loglik0) *loglik0 = e->loglik;
    *out = e;
    return MMG_OK;
}

// Read shards of one problem as one EM (used by mmg_group_em_create and, with a same-device exchange, by the self test): member i
// runs on shards[i]; `reduce` (see mmg_em::reduce) is called between the phases.  ems[0] is the leader: step it, read mu from it.
int mmg::em_create_sharded(const mmg_problem *const *shards, int n_shards, const double *mu0, std::function<int(int)> (*make_reduce)(const std::vector<mmg_em *> &, void *),
                           void *ctx, mmg_em **ems, double *loglik0)
{
    if (!shards || n_shards < 1 || !mu0 || !ems || !make_reduce) return fail(MMG_ERR_ARG, "bad argument");
    std::vector<mmg_em *> es(n_shards, nullptr);
    auto bail = [&](int code) { for (mmg_em *e : es) em_free(e); return code; };
    for (int i = 0; i < n_shards; ++i) {
        if (!shards[i] || shards[i]->n != shards[0]->n) return bail(fail(MMG_ERR_ARG, "shards of one problem have the same transcripts"));
        int rc = em_alloc(shards[i], mu0, &es[i]);
        if (rc) return bail(rc);
    }
    mmg_em *lead = es[0];
    if (n_shards > 1 || opt(MMG_OPT_WIRE_CHECK) >= 1) { // (the wire check of a group of one device takes the exchange path too: tests)
        // the scale words need the hits per transcript of the WHOLE problem: a sum of the shards' counts, held per member
        for (mmg_em *e : es) {
            if (hipSetDevice(e->device) != hipSuccess || hipMalloc((void **)&e->d_colcnt, e->p->n * sizeof(uint64_t)) != hipSuccess) { e->d_colcnt = nullptr; return bail(fail(MMG_ERR_HIP, "hipMalloc (column counts)")); }
            e->owns_colcnt = true;
            if (hipMemcpy(e->d_colcnt, e->p->d_colcnt, e->p->n * sizeof(uint64_t), hipMemcpyDeviceToDevice) != hipSuccess) return bail(fail(MMG_ERR_HIP, "hipMemcpy (column counts)"));
        }
        lead->peers = es;
        lead->reduce = make_reduce(es, ctx);
        int rc = lead->reduce(2);
        if (rc) return bail(rc);
    }
    int rc = em_rows_pass(lead);
    if (rc) return bail(rc);
    if (loglik0) *loglik0 = lead->loglik;
    for (int i = 0; i < n_shards; ++i) ems[i] = es[i];
    return MMG_OK;
}

void mmg::em_exchange_buffers(mmg_em *e, int what, void **ptr, size_t *count)
{
    const size_t n = e->p->n;
    if (what == 0) { *ptr = e->d_xe; *count = n; }
    else if (what == 1) { *ptr = e->d_acc; *count = 2 * n + 3; }
    else { *ptr = e->d_colcnt; *count = n; }
}
int mmg::em_device(const mmg_em *e) { return e->device; }

extern "C" int mmg_em_step(mmg_em *e, double *loglik)
{
    if (!e) return fail(MMG_ERR_ARG, "NULL argument");
    std::vector<mmg_em *> one(1, e);
    for (mmg_em *q : e->peers.empty() ? one : e->peers) {
        HIP_TRY(hipSetDevice(q->device));
        launch_em_apply(q->p->n, q->d_mu, q->p->d_l, q->d_word, q->d_hi, q->d_lo, q->d_sexp, 0);
        if (q != e) ++q->sweeps;
    }
    int rc = em_rows_pass(e);
    if (rc) return rc;
    ++e->sweeps;
    if (loglik) *loglik = e->loglik;
    return MMG_OK;
}

extern "C" int mmg_em_get_mu(mmg_em *e, double *mu)
{
    if (!e || !mu) return fail(MMG_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(e->device));
    return download_ext(e->p, e->d_mu, mu);
}

extern "C" int mmg_em_stats(const mmg_em *e, int *sweeps, int *repeated_passes, int *stream_kernel)
{
    if (!e) return fail(MMG_ERR_ARG, "NULL argument");
    if (sweeps) *sweeps = e->sweeps;
    if (repeated_passes) *repeated_passes = e->repeats;
    if (stream_kernel) *stream_kernel = e->path;
    return MMG_OK;
}

extern "C" void mmg_em_destroy(mmg_em *e) { em_free(e); }

extern "C" int mmg_problem_em(const mmg_problem *cp, double *mu, int max_iter, double epsilon, int *iters, double *loglik)
{
    if (!cp || !mu) return fail(MMG_ERR_ARG, "NULL argument");
    mmg_em *e = nullptr;
    double ll_prev = 0.0;
    int rc = mmg_em_create(cp, mu, &e, &ll_prev);
    if (rc) return rc;
    double llr = __builtin_huge_val(); // the reference starts from epsilon+1 (src/mmseq.cpp:756): first sweep always runs
    int it = 0;
    while (it < max_iter && llr > epsilon) {
        double ll = 0.0;
        rc = mmg_em_step(e, &ll);
        if (rc) { em_free(e); return rc; }
        llr = ll - ll_prev;
        ll_prev = ll;
        ++it;
    }
    rc = mmg_em_get_mu(e, mu);
    em_free(e);
    if (rc) return rc;
    if (iters) *iters = it;
    if (loglik) *loglik = ll_prev;
    return MMG_OK;
}

// ---- read shards of one problem on ONE device, exchanged with plain kernels: the arithmetic of mmg_group_em_* without RCCL
static std::function<int(int)> make_local_reduce(const std::vector<mmg_em *> &es, void *)
{
    return [es](int what) -> int {
        void *p0 = nullptr;
        size_t cnt = 0;
        em_exchange_buffers(es[0], what, &p0, &cnt);
        const size_t bytes = cnt * (what == 0 ? 4 : 8);
        for (size_t i = 1; i < es.size(); ++i) {
            void *pi = nullptr;
            em_exchange_buffers(es[i], what, &pi, &cnt);
            launch_combine(p0, pi, cnt, what == 0, 0);
        }
        for (size_t i = 1; i < es.size(); ++i) {
            void *pi = nullptr;
            em_exchange_buffers(es[i], what, &pi, &cnt);
            HIP_TRY(hipMemcpyAsync(pi, p0, bytes, hipMemcpyDeviceToDevice, 0));
        }
        HIP_TRY(hipGetLastError());
        return MMG_OK;
    };
}

extern "C" int mmg_selftest_em_shards(const mmg_problem *const *shards, int n_shards, const double *mu0, int sweeps, double *mu, double *loglik, int *repeated_passes)
{
    if (!shards || n_shards < 1 || !mu0 || sweeps < 0) return fail(MMG_ERR_ARG, "bad argument");
    for (int i = 1; i < n_shards; ++i) if (!shards[i] || shards[i]->device != shards[0]->device) return fail(MMG_ERR_ARG, "the self test runs its shards on one device");
    std::vector<mmg_em *> ems(n_shards, nullptr);
    double ll = 0.0;
    int rc = em_create_sharded(shards, n_shards, mu0, make_local_reduce, nullptr, ems.data(), &ll);
    if (rc) return rc;
    for (int it = 0; it < sweeps && rc == MMG_OK; ++it) rc = mmg_em_step(ems[0], &ll);
    if (rc == MMG_OK && mu) rc = mmg_em_get_mu(ems[n_shards - 1], mu);   // any member holds the same mu: read the last one
    if (repeated_passes) *repeated_passes = ems[0]->repeats;
    if (loglik) *loglik = ll;
    for (mmg_em *e : ems) em_free(e);
    return rc;
}
