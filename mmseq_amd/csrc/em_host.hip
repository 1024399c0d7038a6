// em_host.hip -- host side of the mmg_em_* entry points: EM sweeps on the device (src/mmseq.cpp:741-811).
#include "mmg_host.h"
#include "mmg_launch.h"

#include <algorithm>

using namespace mmg;

struct mmg_em {
    mmg_problem *p = nullptr;
    int device = 0;
    double *d_mu = nullptr, *d_pc = nullptr;
    uint32_t *d_word = nullptr;
    uint64_t *d_hi = nullptr, *d_lo = nullptr, *d_ll = nullptr;
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
    for (void *x : {(void *)e->d_mu, (void *)e->d_pc, (void *)e->d_word, (void *)e->d_hi, (void *)e->d_lo, (void *)e->d_ll,
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

// One validated rows pass for the current mu: accumulators, log-likelihood.  Carried exponents first
// (unless this is the first pass), repeated on measured exponents if a check failed.
static int em_rows_pass(mmg_em *e)
{
    mmg_problem *p = e->p;
    const unsigned gn = (p->n + 255) / 256;
    for (int measured = e->first ? 1 : 0; measured < 2; ++measured) {
        if (measured) {
            launch_fill_i32(e->d_xe, p->n, INT32_MIN, 0);
            int rc = em_launch_rows(e, true);
            if (rc) return rc;
        }
        launch_em_prepare(p->n, e->d_mu, p->d_l, p->d_colcnt, measured ? e->d_xe : e->d_sexp, measured, e->d_word, e->d_hi, e->d_lo,
                          e->d_pc, e->d_ll, p->d_int_of_ext, 0);
        int rc = em_launch_rows(e, false);
        if (rc) return rc;
        if (!measured) launch_em_check(p->n, e->d_word, e->d_hi, e->d_ll, 0);
        launch_em_finish(e->d_pc, gn, e->d_ll, e->d_out, 0);
        EmOut out;
        HIP_TRY(hipMemcpy(&out, e->d_out, sizeof(out), hipMemcpyDeviceToHost));
        e->loglik = out.loglik;
        if (!out.flag) break;
        if (measured) return fail(MMG_ERR_STATE, "EM: a measured pass failed its own check");
        ++e->repeats;
    }
    e->first = false;
    return MMG_OK;
}

extern "C" int mmg_em_create(const mmg_problem *cp, const double *mu0, mmg_em **out, double *loglik0)
{
    if (!cp || !mu0 || !out) return fail(MMG_ERR_ARG, "NULL argument");
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
    const unsigned gn = (p->n + 255) / 256;
#define EM_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { em_free(e); return fail(MMG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } } while (0)
    EM_TRY(hipMalloc((void **)&e->d_mu, p->n * sizeof(double)));
    EM_TRY(hipMalloc((void **)&e->d_pc, gn * sizeof(double)));
    EM_TRY(hipMalloc((void **)&e->d_word, p->n * sizeof(uint32_t)));
    EM_TRY(hipMalloc((void **)&e->d_hi, p->n * sizeof(uint64_t)));
    EM_TRY(hipMalloc((void **)&e->d_lo, p->n * sizeof(uint64_t)));
    EM_TRY(hipMalloc((void **)&e->d_ll, 4 * sizeof(uint64_t)));
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
    // log-likelihood of the start value (src/mmseq.cpp:745-754)
    int rc = em_rows_pass(e);
    if (rc) { em_free(e); return rc; }
    if (loglik0) *loglik0 = e->loglik;
    *out = e;
    return MMG_OK;
}

extern "C" int mmg_em_step(mmg_em *e, double *loglik)
{
    if (!e) return fail(MMG_ERR_ARG, "NULL argument");
    mmg_problem *p = e->p;
    HIP_TRY(hipSetDevice(p->device));
    launch_em_apply(p->n, e->d_mu, p->d_l, e->d_word, e->d_hi, e->d_lo, e->d_sexp, 0);
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
