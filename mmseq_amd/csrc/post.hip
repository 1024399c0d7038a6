// post.hip -- device TU + host side of the mmg_summary_* entry points: the posterior summary of the resident Gibbs trace
// (src/mmseq.cpp:927-1008, :1110-1227, :1235-1363; src/sokal.cc:33-87).  Kernels in post_kernels.h.
#include "post_kernels.h"
#include "mmg_host.h"
#include "mmg_launch.h"

#include <cmath>
#include <mutex>
#include <vector>

using namespace mmg;

struct SeriesBuf {            // results of one kind of series, on the device
    uint32_t count = 0;
    double *log_mean = nullptr, *var = nullptr, *tau = nullptr, *pct = nullptr;
    int32_t *rc = nullptr;
};
struct PropBuf {
    uint32_t count = 0;
    double *mean = nullptr, *probit_mean = nullptr, *probit_sd = nullptr, *pct = nullptr;
};

struct mmg_summary {
    int device = 0;
    uint32_t n = 0, nv = 0, ni = 0, ng = 0, np = 0, S = 0;
    // sample-major derived traces the writers stream row by row
    double *d_ident = nullptr;   // [S][ni]
    double *d_gene = nullptr;    // [S][ng]
    double *d_prop = nullptr;    // [S][n], caller's numbering
    SeriesBuf ser[4];            // MMG_SERIES_TRANSCRIPT, _VIRTUAL, _IDENTICAL, _GENE
    PropBuf prop[2];             // MMG_SERIES_TRANSCRIPT, _VIRTUAL
    // the summary is built in steps (mmg_summary_begin / _advance / _finish): what the steps share
    const mmg_problem *p = nullptr;
    const double *trace = nullptr; // the chain's resident trace [S][n], device numbering
    hipStream_t st = nullptr;      // the summary's own stream: its kernels neither wait for nor delay the chain
    uint32_t done = 0;             // samples whose derived rows exist
    bool finished = false;
    std::vector<void *> scratch;   // device buffers that live until _finish
    uint64_t *d_iptr = nullptr, *d_gptr = nullptr;
    uint32_t *d_imem = nullptr, *d_gmem = nullptr, *d_gene_t = nullptr, *d_gene_v = nullptr;
    uint8_t *d_multi_t = nullptr, *d_multi_v = nullptr;
    int32_t *d_pind = nullptr;
    double *d_V = nullptr, *d_propV = nullptr, *d_tw = nullptr;
    // mmg_summary_get_rows: a stream and pinned buffers of its own (the writers of a caller fetch rows while the chain runs)
    hipStream_t rows_st = nullptr;
    std::mutex rows_mu;
    PinnedStage rows_stage;
};

static void summary_free(mmg_summary *q)
{
    if (!q) return;
    (void)hipSetDevice(q->device);
    if (q->st) (void)hipStreamSynchronize(q->st);
    for (void *x : q->scratch) if (x) (void)hipFree(x);
    for (void *x : {(void *)q->d_ident, (void *)q->d_gene, (void *)q->d_prop}) if (x) (void)hipFree(x);
    for (auto &b : q->ser) for (void *x : {(void *)b.log_mean, (void *)b.var, (void *)b.tau, (void *)b.pct, (void *)b.rc}) if (x) (void)hipFree(x);
    for (auto &b : q->prop) for (void *x : {(void *)b.mean, (void *)b.probit_mean, (void *)b.probit_sd, (void *)b.pct}) if (x) (void)hipFree(x);
    if (q->st) (void)hipStreamDestroy(q->st);
    if (q->rows_st) (void)hipStreamDestroy(q->rows_st);
    delete q;
}

static inline unsigned blocks_of(uint64_t n) { return (unsigned)((n + 255) / 256); }

// The per-series summary kernel sorts and transforms in LDS up to 8192 samples (128 KB of the 160 KB a workgroup may hold); longer
// traces take the same steps in a global workspace, SERIES_WS_GROUPS workgroups striding over the series.
static constexpr uint32_t SERIES_WS_GROUPS = 1024;
static size_t series_workspace_bytes(uint32_t S)
{
    if (S <= 8192) return 0;
    size_t sp = 1;
    while (sp < S) sp <<= 1;
    return (size_t)SERIES_WS_GROUPS * 3 * sp * 8;
}
template <bool LOG_MODE>
static int launch_series(uint32_t count, uint32_t S, const double *X, uint32_t np, const int32_t *pind, const uint8_t *multi, const double *tw,
                         SeriesOut o, uint64_t *ws, hipStream_t st)
{
    if (count == 0) return MMG_OK;
#define SERIES_IN_LDS(SMAX) hipLaunchKernelGGL((k_series_summary<SMAX, LOG_MODE>), dim3(count), dim3(256), 0, st, count, S, X, np, pind, multi, tw, o, (uint64_t *)nullptr)
    if (S <= 1024) SERIES_IN_LDS(1024);
    else if (S <= 2048) SERIES_IN_LDS(2048);
    else if (S <= 4096) SERIES_IN_LDS(4096);
    else if (S <= 8192) SERIES_IN_LDS(8192);
    else {
        if (!ws) return fail(MMG_ERR_STATE, "no workspace for the summary of a long trace");
        hipLaunchKernelGGL((k_series_summary<0, LOG_MODE>), dim3(count < SERIES_WS_GROUPS ? count : SERIES_WS_GROUPS), dim3(256), 0, st, count, S, X, np, pind,
                           multi, tw, o, ws);
    }
#undef SERIES_IN_LDS
    HIP_TRY(hipGetLastError());
    return MMG_OK;
}

// Step 1: the description is checked and uploaded, the buffers exist, the simulated traces of isoforms without hits (which do not
// depend on the chain, :971-978) are drawn.  The chain may still be running: nothing of its trace is read here.
extern "C" int mmg_summary_begin(mmg_sampler *smp, const mmg_summary_desc *d, mmg_summary **out)
{
    if (!smp || !d || !out) return fail(MMG_ERR_ARG, "NULL argument");
    SamplerView v;
    int rc = sampler_view(smp, &v);
    if (rc) return rc;
    if (!v.d_trace) return fail(MMG_ERR_STATE, "sampler was created with keep_trace == 0");
    if (d->chain < 0 || d->chain >= v.cfg.n_chains) return fail(MMG_ERR_ARG, "chain index out of range");
    const mmg_problem *p = v.p;
    const uint32_t n = p->n, S = (uint32_t)v.cfg.trace_len, nv = d->n_virtual, ni = d->n_identical, ng = d->n_genes, np = d->n_percentiles;
    if ((nv && (!d->virtual_id || !d->virtual_scale)) || (ni && (!d->identical_ptr || !d->identical_member)) ||
        (ng && (!d->gene_ptr || !d->gene_member)) || (np && !d->percentile_index))
        return fail(MMG_ERR_ARG, "summary description: missing array");
    // groups: member indices in range; gene of every transcript (a transcript in no gene gets NaN proportions)
    std::vector<uint32_t> gene_of_t(n, 0xffffffffu), gene_of_v(nv ? nv : 1, 0xffffffffu);
    std::vector<uint8_t> multi_t(n, 0), multi_v(nv ? nv : 1, 0);
    for (uint32_t g = 0; g < ng; ++g) {
        if (d->gene_ptr[g + 1] < d->gene_ptr[g]) return fail(MMG_ERR_ARG, "gene_ptr must be non-decreasing");
        const bool multi = d->gene_ptr[g + 1] - d->gene_ptr[g] > 1; // :1243 a gene with more than one transcript
        for (uint64_t j = d->gene_ptr[g]; j < d->gene_ptr[g + 1]; ++j) {
            const uint32_t m = d->gene_member[j];
            if (m >= n + nv) return fail(MMG_ERR_ARG, "gene member out of range");
            if (m < n) { gene_of_t[m] = g; multi_t[m] = multi; } else { gene_of_v[m - n] = g; multi_v[m - n] = multi; }
        }
    }
    for (uint32_t g = 0; g < ni; ++g) {
        if (d->identical_ptr[g + 1] < d->identical_ptr[g]) return fail(MMG_ERR_ARG, "identical_ptr must be non-decreasing");
        for (uint64_t j = d->identical_ptr[g]; j < d->identical_ptr[g + 1]; ++j)
            if (d->identical_member[j] >= n + nv) return fail(MMG_ERR_ARG, "identical-set member out of range");
    }
    HIP_TRY(hipSetDevice(p->device));
    mmg_summary *q = new mmg_summary();
    q->device = p->device; q->n = n; q->nv = nv; q->ni = ni; q->ng = ng; q->np = np; q->S = S;
    q->p = p;
    q->trace = v.d_trace + (size_t)d->chain * S * n;
    auto bail = [&](int code) { summary_free(q); return code; };
#define Q_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return bail(fail(MMG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e))); } while (0)
    Q_TRY(hipStreamCreateWithFlags(&q->st, hipStreamNonBlocking));
    hipStream_t st = q->st;
    auto dalloc = [&](void **ptr, size_t bytes) { hipError_t e = hipMalloc(ptr, bytes ? bytes : 8); if (e == hipSuccess) q->scratch.push_back(*ptr); return e; };
    auto upload = [&](void **dst, const void *src, size_t bytes) {
        hipError_t e = dalloc(dst, bytes);
        if (e == hipSuccess && bytes) e = hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, st);
        return e;
    };
    uint64_t *d_vid = nullptr;
    double *d_vscale = nullptr;
    Q_TRY(upload((void **)&d_vid, d->virtual_id, (size_t)nv * 8));
    Q_TRY(upload((void **)&d_vscale, d->virtual_scale, (size_t)nv * 8));
    Q_TRY(upload((void **)&q->d_iptr, d->identical_ptr, ((size_t)ni + 1) * 8 * (ni ? 1 : 0)));
    Q_TRY(upload((void **)&q->d_imem, d->identical_member, ni ? (size_t)d->identical_ptr[ni] * 4 : 0));
    Q_TRY(upload((void **)&q->d_gptr, d->gene_ptr, ((size_t)ng + 1) * 8 * (ng ? 1 : 0)));
    Q_TRY(upload((void **)&q->d_gmem, d->gene_member, ng ? (size_t)d->gene_ptr[ng] * 4 : 0));
    Q_TRY(upload((void **)&q->d_gene_t, gene_of_t.data(), (size_t)n * 4));
    Q_TRY(upload((void **)&q->d_gene_v, gene_of_v.data(), (size_t)nv * 4));
    Q_TRY(upload((void **)&q->d_multi_t, multi_t.data(), (size_t)n));
    Q_TRY(upload((void **)&q->d_multi_v, multi_v.data(), (size_t)nv));
    Q_TRY(upload((void **)&q->d_pind, d->percentile_index, (size_t)np * 4));
    // twiddle factors of host/numerics.hpp:fft_pow2, computed with the host's cos / sin: tw[half + j] = exp(-2 pi i j / (2 half))
    std::vector<double> tw(2 * (size_t)(S ? S : 1), 0.0);
    for (uint32_t len = 2; len <= S; len <<= 1) {
        const double ang = -2.0 * M_PI / (double)len;
        const uint32_t half = len / 2;
        for (uint32_t j = 0; j < half; ++j) { tw[2 * (half + j)] = std::cos(ang * (double)j); tw[2 * (half + j) + 1] = std::sin(ang * (double)j); }
    }
    Q_TRY(upload((void **)&q->d_tw, tw.data(), tw.size() * 8));
    Q_TRY(dalloc((void **)&q->d_V, (size_t)S * nv * 8));
    Q_TRY(dalloc((void **)&q->d_propV, (size_t)S * nv * 8));
    if (nv) hipLaunchKernelGGL(k_virtual_traces, dim3(blocks_of((uint64_t)nv * S)), dim3(256), 0, st, v.cfg.seed, v.cfg.alpha, nv, S, d_vid, d_vscale, q->d_V);
    Q_TRY(hipGetLastError());
    Q_TRY(hipMalloc((void **)&q->d_ident, (size_t)S * (ni ? ni : 1) * 8));
    Q_TRY(hipMalloc((void **)&q->d_gene, (size_t)S * (ng ? ng : 1) * 8));
    Q_TRY(hipMalloc((void **)&q->d_prop, (size_t)S * n * 8));
    Q_TRY(hipStreamSynchronize(st));   // (the host vectors of this call were sources of asynchronous copies)
#undef Q_TRY
    *out = q;
    return MMG_OK;
}

// Step 2: the derived sample rows -- sums over identical sets and genes (:927-1008), proportions (:1014-1031) -- of the samples
// [done, samples_done).  The caller vouches that the chain has finished those samples (it synchronised after iteration
// samples_done * gibbs_ss - 1 or later); iterations enqueued behind them neither are waited for nor delayed.
extern "C" int mmg_summary_advance(mmg_summary *q, int samples_done)
{
    if (!q) return fail(MMG_ERR_ARG, "NULL summary");
    if (q->finished) return fail(MMG_ERR_STATE, "the summary is finished");
    if (samples_done < (int)q->done || samples_done > (int)q->S) return fail(MMG_ERR_ARG, "samples_done out of range");
    if ((uint32_t)samples_done == q->done) return MMG_OK;
    HIP_TRY(hipSetDevice(q->device));
    const uint32_t s0 = q->done, c = (uint32_t)samples_done - s0, n = q->n, nv = q->nv, ni = q->ni, ng = q->ng;
    hipStream_t st = q->st;
    const double *tr = q->trace + (size_t)s0 * n, *V = q->d_V + (size_t)s0 * nv;
    const uint32_t *ioe = q->p->d_int_of_ext;
    if (ni) hipLaunchKernelGGL(k_group_sums, dim3(blocks_of((uint64_t)ni * c)), dim3(256), 0, st, ni, c, n, nv, q->d_iptr, q->d_imem, ioe, tr, V, q->d_ident + (size_t)s0 * ni);
    if (ng) hipLaunchKernelGGL(k_group_sums, dim3(blocks_of((uint64_t)ng * c)), dim3(256), 0, st, ng, c, n, nv, q->d_gptr, q->d_gmem, ioe, tr, V, q->d_gene + (size_t)s0 * ng);
    hipLaunchKernelGGL(k_proportions, dim3(blocks_of((uint64_t)n * c)), dim3(256), 0, st, n, c, n, tr, ioe, q->d_gene_t, ng, q->d_gene + (size_t)s0 * ng, q->d_prop + (size_t)s0 * n);
    if (nv) hipLaunchKernelGGL(k_proportions, dim3(blocks_of((uint64_t)nv * c)), dim3(256), 0, st, nv, c, nv, V, (const uint32_t *)nullptr, q->d_gene_v, ng, q->d_gene + (size_t)s0 * ng, q->d_propV + (size_t)s0 * nv);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    q->done = (uint32_t)samples_done;
    return MMG_OK;
}

// Step 3, once every sample is in: per series the percentiles (:1110-1192), the mean of the logged trace (:1195-1227), Sokal's
// variance and autocorrelation time (:1307-1363), the proportion summaries (:1235-1305).
extern "C" int mmg_summary_finish(mmg_summary *q)
{
    if (!q) return fail(MMG_ERR_ARG, "NULL summary");
    if (q->finished) return MMG_OK;
    if (q->done != q->S) return fail(MMG_ERR_STATE, "mmg_summary_finish before every sample was handed to mmg_summary_advance");
    HIP_TRY(hipSetDevice(q->device));
    const uint32_t n = q->n, nv = q->nv, ni = q->ni, ng = q->ng, np = q->np, S = q->S;
    hipStream_t st = q->st;
    // transpose to series-major, one workgroup per series
    size_t maxcnt = n;
    for (size_t c : {(size_t)nv, (size_t)ni, (size_t)ng}) if (c > maxcnt) maxcnt = c;
    double *d_T = nullptr;
    HIP_TRY(hipMalloc((void **)&d_T, maxcnt * S * 8));
    q->scratch.push_back(d_T);
    uint64_t *d_ws = nullptr;
    if (series_workspace_bytes(S)) {
        HIP_TRY(hipMalloc((void **)&d_ws, series_workspace_bytes(S)));
        q->scratch.push_back(d_ws);
    }
    const uint32_t counts[4] = {n, nv, ni, ng};
    const double *srcs[4] = {q->trace, q->d_V, q->d_ident, q->d_gene};
    for (int k = 0; k < 4; ++k) {
        SeriesBuf &b = q->ser[k];
        b.count = counts[k];
        const size_t c = counts[k] ? counts[k] : 1;
        HIP_TRY(hipMalloc((void **)&b.log_mean, c * 8));
        HIP_TRY(hipMalloc((void **)&b.var, c * 8));
        HIP_TRY(hipMalloc((void **)&b.tau, c * 8));
        HIP_TRY(hipMalloc((void **)&b.rc, c * 4));
        HIP_TRY(hipMalloc((void **)&b.pct, c * (np ? np : 1) * 8));
        if (!counts[k]) continue;
        launch_transpose(srcs[k], d_T, counts[k], S, k == MMG_SERIES_TRANSCRIPT ? q->p->d_int_of_ext : nullptr, st);
        SeriesOut o{b.log_mean, b.var, b.tau, b.rc, b.pct, nullptr, nullptr, nullptr};
        int rc = launch_series<true>(counts[k], S, d_T, np, q->d_pind, nullptr, q->d_tw, o, d_ws, st);
        if (rc) return rc;
    }
    const double *psrc[2] = {q->d_prop, q->d_propV};
    const uint8_t *pmulti[2] = {q->d_multi_t, q->d_multi_v};
    for (int k = 0; k < 2; ++k) {
        PropBuf &b = q->prop[k];
        b.count = counts[k];
        const size_t c = counts[k] ? counts[k] : 1;
        HIP_TRY(hipMalloc((void **)&b.mean, c * 8));
        HIP_TRY(hipMalloc((void **)&b.probit_mean, c * 8));
        HIP_TRY(hipMalloc((void **)&b.probit_sd, c * 8));
        HIP_TRY(hipMalloc((void **)&b.pct, c * (np ? np : 1) * 8));
        if (!counts[k]) continue;
        launch_transpose(psrc[k], d_T, counts[k], S, nullptr, st); // d_prop is in the caller's numbering already
        SeriesOut o{nullptr, nullptr, nullptr, nullptr, b.pct, b.mean, b.probit_mean, b.probit_sd};
        int rc = launch_series<false>(counts[k], S, d_T, np, q->d_pind, pmulti[k], q->d_tw, o, d_ws, st);
        if (rc) return rc;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    for (void *x : q->scratch) if (x) (void)hipFree(x);   // the group tables, the virtual traces: the results no longer need them
    q->scratch.clear();
    q->d_V = q->d_propV = nullptr;
    q->finished = true;
    return MMG_OK;
}

// The three steps at once, after the chain has run.
extern "C" int mmg_summary_create(mmg_sampler *smp, const mmg_summary_desc *d, mmg_summary **out)
{
    if (!out) return fail(MMG_ERR_ARG, "NULL argument");
    mmg_summary *q = nullptr;
    int rc = mmg_summary_begin(smp, d, &q);
    if (rc) return rc;
    rc = mmg_sampler_sync(smp);                      // every sample is final
    if (rc == MMG_OK) rc = mmg_summary_advance(q, (int)q->S);
    if (rc == MMG_OK) rc = mmg_summary_finish(q);
    if (rc) { summary_free(q); return rc; }
    *out = q;
    return MMG_OK;
}

static int check_kind(const mmg_summary *q, int kind, int max_kind)
{
    if (!q) return fail(MMG_ERR_ARG, "NULL summary");
    if (kind < 0 || kind > max_kind) return fail(MMG_ERR_ARG, "series kind out of range");
    return MMG_OK;
}

extern "C" int mmg_summary_get(mmg_summary *q, int kind, double *log_mean, double *var, double *tau, int32_t *sokal_rc, double *percentiles)
{
    int rc = check_kind(q, kind, MMG_SERIES_GENE);
    if (rc) return rc;
    if (!q->finished) return fail(MMG_ERR_STATE, "summary columns exist after mmg_summary_finish");
    HIP_TRY(hipSetDevice(q->device));
    const SeriesBuf &b = q->ser[kind];
    const size_t c = b.count;
    if (!c) return MMG_OK;
    if (log_mean) HIP_TRY(hipMemcpy(log_mean, b.log_mean, c * 8, hipMemcpyDeviceToHost));
    if (var) HIP_TRY(hipMemcpy(var, b.var, c * 8, hipMemcpyDeviceToHost));
    if (tau) HIP_TRY(hipMemcpy(tau, b.tau, c * 8, hipMemcpyDeviceToHost));
    if (sokal_rc) HIP_TRY(hipMemcpy(sokal_rc, b.rc, c * 4, hipMemcpyDeviceToHost));
    if (percentiles && q->np) HIP_TRY(hipMemcpy(percentiles, b.pct, c * q->np * 8, hipMemcpyDeviceToHost));
    return MMG_OK;
}

extern "C" int mmg_summary_get_proportions(mmg_summary *q, int kind, double *mean_prop, double *mean_probit, double *sd_probit, double *percentiles)
{
    int rc = check_kind(q, kind, MMG_SERIES_VIRTUAL);
    if (rc) return rc;
    if (!q->finished) return fail(MMG_ERR_STATE, "summary columns exist after mmg_summary_finish");
    HIP_TRY(hipSetDevice(q->device));
    const PropBuf &b = q->prop[kind];
    const size_t c = b.count;
    if (!c) return MMG_OK;
    if (mean_prop) HIP_TRY(hipMemcpy(mean_prop, b.mean, c * 8, hipMemcpyDeviceToHost));
    if (mean_probit) HIP_TRY(hipMemcpy(mean_probit, b.probit_mean, c * 8, hipMemcpyDeviceToHost));
    if (sd_probit) HIP_TRY(hipMemcpy(sd_probit, b.probit_sd, c * 8, hipMemcpyDeviceToHost));
    if (percentiles && q->np) HIP_TRY(hipMemcpy(percentiles, b.pct, c * q->np * 8, hipMemcpyDeviceToHost));
    return MMG_OK;
}

extern "C" int mmg_summary_get_rows(mmg_summary *q, int kind, int first_sample, int n_samples, double *out)
{
    if (!q || !out) return fail(MMG_ERR_ARG, "NULL argument");
    const double *src = nullptr;
    size_t width = 0;
    switch (kind) {
    case MMG_SERIES_TRANSCRIPT: src = q->d_prop; width = q->n; break;   // proportions of gene expression, caller's numbering
    case MMG_SERIES_IDENTICAL: src = q->d_ident; width = q->ni; break;
    case MMG_SERIES_GENE: src = q->d_gene; width = q->ng; break;
    default: return fail(MMG_ERR_ARG, "rows exist for the proportion, identical-set and gene traces");
    }
    if (first_sample < 0 || n_samples < 0 || (int64_t)first_sample + n_samples > (int64_t)q->S) return fail(MMG_ERR_ARG, "bad sample range");
    if ((uint32_t)(first_sample + n_samples) > q->done) return fail(MMG_ERR_STATE, "rows of samples that were not yet handed to mmg_summary_advance");
    HIP_TRY(hipSetDevice(q->device));
    if (width && n_samples) {
        std::lock_guard<std::mutex> lock(q->rows_mu);
        if (!q->rows_st) HIP_TRY(hipStreamCreateWithFlags(&q->rows_st, hipStreamNonBlocking));
        HIP_TRY(q->rows_stage.copy_out(out, src + (size_t)first_sample * width, (size_t)n_samples * width * 8, q->rows_st));
    }
    return MMG_OK;
}

extern "C" void mmg_summary_destroy(mmg_summary *q) { summary_free(q); }
