// sampler.hip -- host side of the mmg_sampler_* entry points: the Gibbs loop of src/mmseq.cpp:851-918 as a sequence of
// K1 (sample + scatter) and K2 (Gamma redraw + trace) launches on one stream.
#include "mmg_host.h"
#include "mmg_launch.h"
#include <hip/hip_ext.h>

#include <algorithm>
#include <array>
#include <deque>
#include <mutex>

using namespace mmg;

struct mmg_sampler {
    const mmg_problem *p = nullptr;
    int device = 0;
    mmg_config cfg{};
    hipStream_t own = nullptr, cur = nullptr;
    // the launch of the rows on the conditional-binomial chain (k_sample_bigk: few waves, long chains of arithmetic) runs beside the tile
    // kernels (bound by the stream of hits) on a stream of its own, forked from and joined to `cur` inside every sample()
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    double *d_mu = nullptr, *d_scale = nullptr, *d_trace = nullptr, *d_mom = nullptr; // mom: [2][C][n]
    int32_t *d_cnt = nullptr, *d_cnt_last = nullptr;
    int iter = 0;          // completed iterations
    bool sampled = false;  // sample() issued for the current iteration, update() pending
    int64_t n_kept = 0;
    // timing
    // HIP-event pairs around timed launches: a pair is harvested (its time added to the sums, its events back on the free list) as soon
    // as it has completed, so the pool stays as large as the launches in flight however long the chain
    std::vector<hipEvent_t> ev_pool;
    std::vector<int> ev_free;                              // indices into ev_pool
    std::deque<std::array<int, 4>> ev_pending;             // {start, stop, 0 sample | 1 update, stop of the side launch or -1} in enqueue order
    hipStream_t reader = nullptr;                          // mmg_sampler_get_trace_rows_done: copies that do not queue behind the chain
    double *d_reader_tmp = nullptr;                        // its gather buffer, kept (hipFree would wait for the running chain)
    size_t reader_cap = 0;
    std::mutex reader_mu;
    PinnedStage reader_stage;
    // marks: an event behind every 16th stored sample, for mmg_sampler_wait_iterations (events without timing, recycled)
    std::vector<hipEvent_t> mark_pool;
    std::vector<int> mark_free;
    std::deque<std::pair<int, int>> marks;                 // {iterations completed when the event fires, index into mark_pool}
    int fired_upto = 0;                                    // iterations known to have completed: the last mark seen fired (recycled or waited for)
    double acc_sample_ms = 0, acc_update_ms = 0;
    uint64_t acc_sample_n = 0, acc_update_n = 0;
};

static void sampler_free(mmg_sampler *s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->own) { (void)hipStreamSynchronize(s->own); }
    for (auto e : s->ev_pool) (void)hipEventDestroy(e);
    for (auto e : s->mark_pool) (void)hipEventDestroy(e);
    for (void *x : {(void *)s->d_mu, (void *)s->d_scale, (void *)s->d_trace, (void *)s->d_mom, (void *)s->d_cnt, (void *)s->d_cnt_last})
        if (x) (void)hipFree(x);
    if (s->side) { (void)hipStreamSynchronize(s->side); (void)hipStreamDestroy(s->side); }
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    if (s->ev_join) (void)hipEventDestroy(s->ev_join);
    if (s->own) (void)hipStreamDestroy(s->own);
    if (s->reader) (void)hipStreamDestroy(s->reader);
    if (s->d_reader_tmp) (void)hipFree(s->d_reader_tmp);
    delete s;
}

int mmg::sampler_view(mmg_sampler *s, SamplerView *v)
{
    if (!s || !v) return fail(MMG_ERR_ARG, "NULL argument");
    v->p = s->p; v->cfg = s->cfg; v->d_trace = s->d_trace; v->stream = s->cur; v->iter = s->iter; v->n_kept = s->n_kept;
    return MMG_OK;
}

extern "C" int mmg_sampler_create(const mmg_problem *p, const mmg_config *cfg, const double *mu0, mmg_sampler **out)
{
    if (!p || !cfg || !mu0 || !out) return fail(MMG_ERR_ARG, "NULL argument");
    if (cfg->n_chains < 1 || cfg->n_chains > 4096) return fail(MMG_ERR_ARG, "n_chains out of range");
    if (!(cfg->alpha > 0.0) || !(cfg->beta > 0.0)) return fail(MMG_ERR_ARG, "alpha, beta must be > 0");
    if (cfg->trace_len < 1 || cfg->gibbs_iter < 1) return fail(MMG_ERR_ARG, "gibbs_iter and trace_len must be >= 1 (src/mmseq.cpp:286)");
    if (cfg->gibbs_iter % cfg->trace_len != 0) return fail(MMG_ERR_ARG, "gibbs_iter must be a multiple of trace_len (src/mmseq.cpp:278-284)");
    for (uint32_t t = 0; t < p->n; ++t)
        if (!(mu0[t] >= 0.0)) return fail(MMG_ERR_ARG, "mu0 must be finite and >= 0");
    int rc = require_device(p->device);
    if (rc) return rc;
    mmg_sampler *s = new mmg_sampler();
    s->p = p;
    s->device = p->device;
    s->cfg = *cfg;
    const size_t C = (size_t)cfg->n_chains, n = p->n;
    auto bail = [&](int code) { sampler_free(s); return code; };
#define S_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return bail(fail(MMG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e))); } while (0)
    S_TRY(hipStreamCreateWithFlags(&s->own, hipStreamNonBlocking));
    s->cur = s->own;
    if (p->grid_bigk > 0) {
        S_TRY(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
        S_TRY(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
        S_TRY(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
    }
    S_TRY(hipMalloc((void **)&s->d_mu, C * n * sizeof(double)));
    S_TRY(hipMalloc((void **)&s->d_scale, n * sizeof(double)));
    S_TRY(hipMalloc((void **)&s->d_mom, 2 * C * n * sizeof(double)));
    S_TRY(hipMalloc((void **)&s->d_cnt, p->cnt_replicas * C * n * sizeof(int32_t))); // [replicas][C][n]; replica 0 is the public vector
    S_TRY(hipMalloc((void **)&s->d_cnt_last, C * n * sizeof(int32_t)));
    if (cfg->keep_trace) S_TRY(hipMalloc((void **)&s->d_trace, C * n * (size_t)cfg->trace_len * sizeof(double)));
    std::vector<double> scale_ext(n), scale, mu_int;
    for (size_t t = 0; t < n; ++t) scale_ext[t] = 1.0 / (cfg->beta + p->h_l[t]); // src/mmseq.cpp:907 second argument
    to_int(p, scale_ext.data(), scale);
    to_int(p, mu0, mu_int);
    // every fill goes to the sampler's own stream (non-blocking: not ordered against the NULL stream) and is waited for here
    S_TRY(hipMemcpyAsync(s->d_scale, scale.data(), n * sizeof(double), hipMemcpyHostToDevice, s->own));
    for (size_t c = 0; c < C; ++c) S_TRY(hipMemcpyAsync(s->d_mu + c * n, mu_int.data(), n * sizeof(double), hipMemcpyHostToDevice, s->own));
    S_TRY(hipMemsetAsync(s->d_mom, 0, 2 * C * n * sizeof(double), s->own));
    S_TRY(hipMemsetAsync(s->d_cnt, 0, p->cnt_replicas * C * n * sizeof(int32_t), s->own));
    S_TRY(hipMemsetAsync(s->d_cnt_last, 0, C * n * sizeof(int32_t), s->own));
    if (s->d_trace) S_TRY(hipMemsetAsync(s->d_trace, 0, C * n * (size_t)cfg->trace_len * sizeof(double), s->own));
    S_TRY(hipStreamSynchronize(s->own));
#undef S_TRY
    *out = s;
    return MMG_OK;
}

extern "C" int mmg_sampler_set_stream(mmg_sampler *s, void *hip_stream)
{
    if (!s) return fail(MMG_ERR_ARG, "NULL sampler");
    s->cur = hip_stream ? (hipStream_t)hip_stream : s->own;
    return MMG_OK;
}

// pairs at the head of the queue that have completed (all of them after a synchronisation: wait == true)
static int ev_harvest(mmg_sampler *s, bool wait)
{
    while (!s->ev_pending.empty()) {
        const std::array<int, 4> pr = s->ev_pending.front();
        if (!wait) {
            for (int q_i : {1, 3}) {
                if (pr[q_i] < 0) continue;
                const hipError_t q = hipEventQuery(s->ev_pool[pr[q_i]]);
                if (q == hipErrorNotReady) { (void)hipGetLastError(); return MMG_OK; }
                if (q != hipSuccess) return fail(MMG_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(q));
            }
        }
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, s->ev_pool[pr[0]], s->ev_pool[pr[1]]));
        if (pr[3] >= 0) { // the side launch may end after the last launch of the main stream: K1's time is to the later of the two
            float ms2 = 0;
            HIP_TRY(hipEventElapsedTime(&ms2, s->ev_pool[pr[0]], s->ev_pool[pr[3]]));
            ms = std::max(ms, ms2);
            s->ev_free.push_back(pr[3]);
        }
        if (pr[2] == 0) { s->acc_sample_ms += ms; s->acc_sample_n++; } else { s->acc_update_ms += ms; s->acc_update_n++; }
        s->ev_free.push_back(pr[0]); s->ev_free.push_back(pr[1]);
        s->ev_pending.pop_front();
    }
    return MMG_OK;
}

static int ev_get(mmg_sampler *s, int &idx)
{
    if (s->ev_free.empty() && s->ev_pending.size() >= 64) { int rc = ev_harvest(s, false); if (rc) return rc; }
    if (s->ev_free.empty()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        s->ev_pool.push_back(e);
        s->ev_free.push_back((int)s->ev_pool.size() - 1);
    }
    idx = s->ev_free.back();
    s->ev_free.pop_back();
    return MMG_OK;
}

// K1 for every chain of the sampler.  fold: leave the device's column sums in the public count vector (replica 0) -- what a caller
// that exchanges counts between sample and update needs; mmg_sampler_run skips it (K2 sums the replicas itself).
static int sampler_sample(mmg_sampler *s, bool fold)
{
    if (!s) return fail(MMG_ERR_ARG, "NULL sampler");
    if (s->sampled) return fail(MMG_ERR_STATE, "sample() already issued for this iteration; call update()");
    const mmg_problem *p = s->p;
    HIP_TRY(hipSetDevice(s->device));
    int e0 = -1, e1 = -1;
    const bool timed = s->cfg.timing > 0 && s->iter % s->cfg.timing == 0;
    // A timed iteration's events ride on the launches themselves (hipExtLaunchKernel: the dispatch's own start and end time stamps): the
    // first launch of the call carries the start event, every launch the stop event (the last one keeps it).  Recorded into the stream
    // as packets of their own, a pair cost ~9 us of stream time -- 1.8 % of a config-3 step when every fourth step is timed.
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    if (timed) {
        int rc = ev_get(s, e0); if (rc) return rc;
        rc = ev_get(s, e1); if (rc) return rc;
        ev_start = s->ev_pool[e0]; ev_stop = s->ev_pool[e1];
    }
    int n_launched = 0, e2 = -1;
    bool joined = true;
    auto launch = [&](const void *fn, dim3 grid, dim3 block, void **kargs) -> hipError_t {
        const hipError_t e = timed ? hipExtLaunchKernel(fn, grid, block, kargs, 0, s->cur, n_launched == 0 ? ev_start : nullptr, ev_stop, 0)
                                   : hipLaunchKernel(fn, grid, block, kargs, 0, s->cur);
        ++n_launched;
        return e;
    };
    if (p->m > 0) {
        const int C = s->cfg.n_chains;
        auto args_of = [&](int c) {
            SampleArgs a;
            a.seed = s->cfg.seed; a.row_id_base = p->row_id_base; a.n = p->n;
            a.cnt_rep_stride = (uint64_t)C * p->n;
            a.cnt_rep_mask = p->cnt_replicas - 1u;
            a.chain = (uint32_t)(s->cfg.chain_base + c);
            a.iter = (uint32_t)s->iter;
            return a;
        };
        const void *rp = p->d_row_ptr;
        const uint32_t *ci = p->d_col, *kk = p->d_k;
        const uint8_t *ss = p->d_sell;
        // k_sample_sell over one tile list for chains [c0, c0 + nc): grid.y = chain
        // kind: 0 k_sample_sell, 1 its multiplicity instantiation, 2 its far-list instantiation
        auto launch_single = [&](const SellTile *ts, const uint64_t *cs, int grid, int kind, int c0, int nc) -> int {
            SampleArgs a = args_of(c0);
            const double *mu = s->d_mu + (size_t)c0 * p->n;
            int32_t *cnt = s->d_cnt + (size_t)c0 * p->n;
            void *kargs[] = {(void *)&rp, (void *)&ci, (void *)&kk, (void *)&ts, (void *)&cs, (void *)&mu, (void *)&ss, (void *)&cnt, (void *)&a};
            const void *fn = kind == 2 ? k1_sell_far_kernel(p->idx64) : k1_sell_kernel(p->idx64, kind == 1, kind == 0 && p->k1_fixed_walk);
            HIP_TRY(launch(fn, dim3(grid, nc), dim3(64), kargs));
            return MMG_OK;
        };
        if (p->use_sell && p->grid_bigk > 0) {
            // The rows on the conditional-binomial chain, from their list, for every chain (grid.y): few waves with long dependent chains
            // of arithmetic.  They start first, on the side stream (ordered behind everything enqueued on `cur` so far -- the update that
            // wrote mu), and the tile kernels below fill the device around them; sample() ends with `cur` waiting for the side stream.
            SampleArgs a = args_of(0);
            const double *mu = s->d_mu;
            int32_t *cnt = s->d_cnt;
            const uint64_t *list = p->d_bigk_list;
            uint64_t n_list = p->n_bigk;
            uint32_t per = p->bigk_per_wave;
            void *kargs[] = {(void *)&rp, (void *)&ci, (void *)&kk, (void *)&list, (void *)&n_list, (void *)&per, (void *)&mu, (void *)&cnt, (void *)&a};
            // (not beside the legacy NULL stream or the per-thread stream: special handles, which the event calls below do not take here)
            const bool beside = opt(MMG_OPT_BIGK_SIDE_STREAM) != 0 && s->cur != hipStreamLegacy && s->cur != hipStreamPerThread;
            if (beside) {
                HIP_TRY(hipEventRecord(s->ev_fork, s->cur));
                HIP_TRY(hipStreamWaitEvent(s->side, s->ev_fork, 0));
                if (timed) { int rc = ev_get(s, e2); if (rc) return rc; }
                HIP_TRY(timed ? hipExtLaunchKernel(k1_bigk_kernel(p->idx64), dim3(p->grid_bigk, C), dim3(64), kargs, 0, s->side, nullptr, s->ev_pool[e2], 0)
                              : hipLaunchKernel(k1_bigk_kernel(p->idx64), dim3(p->grid_bigk, C), dim3(64), kargs, 0, s->side));
                HIP_TRY(hipEventRecord(s->ev_join, s->side));
                joined = false;
            } else HIP_TRY(launch(k1_bigk_kernel(p->idx64), dim3(p->grid_bigk, C), dim3(64), kargs));
        }
        if (p->use_sell) {
            // Chains are advanced in fused pairs over the register-path tiles without multiplicities (measured at config 3 with 8
            // chains: 3470 chain-iterations/s one chain per launch, 3780 in pairs at 4 waves per SIMD, 2970 in fours at 2 waves
            // per SIMD: fours exist for tests and experiments only, MMG_OPT_FUSE_CHAINS).  The tiles that are neither -- far tiles,
            // CSR-walked tiles, tiles holding rows with multiplicities: what every real hits file has -- take the single-chain
            // kernel, one launch per kind for all the paired chains together.
            const int want = opt(MMG_OPT_FUSE_CHAINS);
            int fuse = want == 1 ? 1 : (want >= 4 ? 4 : 2);
            if (fuse == 4 && p->grid_sell_m[1] <= 0) fuse = 2;
            if (p->grid_sell_m[0] <= 0 || !p->d_sell_tiles_f) fuse = 1;
            const int n_fused = fuse > 1 ? (C / fuse) * fuse : 0, n_rest = fuse == 4 ? ((C - n_fused) / 2) * 2 : 0;
            // one launch per fusion width: grid.y = the groups of `f` chains
            for (int c = 0; c < n_fused + n_rest; ) {
                const int f = c < n_fused ? fuse : 2;
                const int groups = (c < n_fused ? n_fused - c : n_fused + n_rest - c) / f;
                SampleArgs a = args_of(c);
                const SellTile *ts = p->d_sell_tiles_f;
                const uint64_t *cs = p->d_sell_chunk_m[f == 4 ? 1 : 0];
                const double *mu = s->d_mu + (size_t)c * p->n;
                int32_t *cnt = s->d_cnt + (size_t)c * p->n;
                void *kargs[] = {(void *)&rp, (void *)&ci, (void *)&ts, (void *)&cs, (void *)&mu, (void *)&ss, (void *)&cnt, (void *)&a};
                HIP_TRY(launch(k1_sell_multi_kernel(p->idx64, f), dim3(p->grid_sell_m[f == 4 ? 1 : 0], groups), dim3(64), kargs));
                c += f * groups;
            }
            const int n_paired = n_fused + n_rest;
            if (n_paired > 0 && p->grid_sell_x > 0) { // their far / CSR-walked tiles: the far-list instantiation, all paired chains in one launch
                int rc = launch_single(p->d_sell_tiles_x, p->d_sell_chunk_x, p->grid_sell_x, 2, 0, n_paired);
                if (rc) return rc;
            }
            if (n_paired < C) { // chains without a partner: every tile without multiplicities in ONE launch (a launch of its own for the far
                                // tiles costs a single chain as much as it saves: 29 us for 17 k far tiles at 2 % far rows)
                int rc = launch_single(p->grid_sell_k > 0 ? p->d_sell_tiles_1 : p->d_sell_tiles, p->d_sell_chunk, p->grid_sell, 0, n_paired, C - n_paired);
                if (rc) return rc;
            }
            if (p->grid_sell_k > 0) { // the tiles that hold collapsed identical reads, in ranges balanced by their cost, for every chain
                int rc = launch_single(p->d_sell_tiles_k, p->d_sell_chunk_k, p->grid_sell_k, 1, 0, C);
                if (rc) return rc;
            }
        } else {
            for (int c = 0; c < C; ++c) {
                SampleArgs a = args_of(c);
                const TileDesc *td = p->d_tiles;
                const uint64_t *ct = p->d_chunk_tile;
                const double *mu = s->d_mu + (size_t)c * p->n;
                int32_t *cnt = s->d_cnt + (size_t)c * p->n;
                void *kargs[] = {(void *)&rp, (void *)&ci, (void *)&kk, (void *)&td, (void *)&ct, (void *)&mu, (void *)&cnt, (void *)&a};
                HIP_TRY(launch(k1_csr_kernel(p->idx64, p->d_k != nullptr), dim3(p->grid_sample), dim3(K1C_BS), kargs));
            }
        }
    }
    if (!joined) HIP_TRY(hipStreamWaitEvent(s->cur, s->ev_join, 0));
    if (timed) {
        if (n_launched == 0) { HIP_TRY(hipEventRecord(ev_start, s->cur)); HIP_TRY(hipEventRecord(ev_stop, s->cur)); } // (a problem without rows)
        s->ev_pending.push_back({e0, e1, 0, e2});
    }
    if (fold && p->m > 0 && p->cnt_replicas > 1) {
        launch_fold_counts(s->d_cnt, (uint64_t)s->cfg.n_chains * p->n, (size_t)s->cfg.n_chains * p->n, s->cur);
        HIP_TRY(hipGetLastError());
    }
    s->sampled = true;
    return MMG_OK;
}

constexpr int MARK_EVERY = 16;
// An event behind the iteration just enqueued (one that stored a sample): mmg_sampler_wait_iterations waits on these.  Marks that
// have fired are recycled here, so the pool is as large as the samples in flight.
static int mark_iteration(mmg_sampler *s)
{
    while (!s->marks.empty()) {
        if (hipEventQuery(s->mark_pool[s->marks.front().second]) != hipSuccess) { (void)hipGetLastError(); break; }
        s->fired_upto = std::max(s->fired_upto, s->marks.front().first);
        s->mark_free.push_back(s->marks.front().second);
        s->marks.pop_front();
    }
    if (s->mark_free.empty()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        s->mark_pool.push_back(e);
        s->mark_free.push_back((int)s->mark_pool.size() - 1);
    }
    const int idx = s->mark_free.back();
    s->mark_free.pop_back();
    HIP_TRY(hipEventRecord(s->mark_pool[idx], s->cur));
    s->marks.push_back({s->iter, idx});
    return MMG_OK;
}

extern "C" int mmg_sampler_sample(mmg_sampler *s) { return sampler_sample(s, true); }

extern "C" int mmg_sampler_update(mmg_sampler *s)
{
    if (!s) return fail(MMG_ERR_ARG, "NULL sampler");
    if (!s->sampled) return fail(MMG_ERR_STATE, "update() without a preceding sample()");
    const mmg_problem *p = s->p;
    HIP_TRY(hipSetDevice(s->device));
    const int ss = s->cfg.gibbs_iter / s->cfg.trace_len; // src/mmseq.cpp:284
    int sample_idx = -1;
    if (s->iter % ss == 0 && s->iter / ss < s->cfg.trace_len) sample_idx = s->iter / ss; // :911, :914
    const size_t C = (size_t)s->cfg.n_chains, n = p->n;
    UpdateArgs a;
    a.cnt = s->d_cnt; a.cnt_last = s->d_cnt_last; a.scale = s->d_scale; a.mu = s->d_mu; a.trace = s->d_trace;
    a.sum_log = s->d_mom; a.sum_log2 = s->d_mom + C * n;
    a.ext_of_int = p->d_ext_of_int;
    a.seed = s->cfg.seed; a.alpha = s->cfg.alpha; a.n = p->n; a.n_chains = (uint32_t)C;
    a.cnt_rep_stride = (uint64_t)C * n;
    a.cnt_replicas = p->cnt_replicas;
    a.chain_base = (uint32_t)s->cfg.chain_base; a.iter = (uint32_t)s->iter; a.sample_idx = sample_idx;
    a.trace_len = (uint32_t)s->cfg.trace_len;
    int e0 = -1, e1 = -1;
    const bool timed = s->cfg.timing > 0 && s->iter % s->cfg.timing == 0;
    if (timed) {
        int rc = ev_get(s, e0); if (rc) return rc;
        rc = ev_get(s, e1); if (rc) return rc;
    }
    launch_update(a, s->cur, timed ? s->ev_pool[e0] : nullptr, timed ? s->ev_pool[e1] : nullptr); // (the events ride on the launch: sampler_sample)
    HIP_TRY(hipGetLastError());
    if (timed) s->ev_pending.push_back({e0, e1, 1, -1});
    if (sample_idx >= 0) s->n_kept++;
    s->iter++;
    s->sampled = false;
    // (an event per iteration costs the chain 4-5 us of stream time each: every MARK_EVERY-th stored sample is enough for a caller
    // that hands the trace on in pieces)
    if (sample_idx >= 0 && ((sample_idx + 1) % MARK_EVERY == 0 || sample_idx + 1 == s->cfg.trace_len)) { int rc = mark_iteration(s); if (rc) return rc; }
    return MMG_OK;
}

extern "C" int mmg_sampler_run(mmg_sampler *s, int n_iter)
{
    if (!s || n_iter < 0) return fail(MMG_ERR_ARG, "bad argument");
    for (int i = 0; i < n_iter; ++i) {
        int rc = sampler_sample(s, false);
        if (rc) return rc;
        rc = mmg_sampler_update(s);
        if (rc) return rc;
    }
    return MMG_OK;
}

// The sharded chain of mmg_group_run_sharded with every shard on ONE device: per iteration K1 on every shard, the count vectors summed
// by a plain kernel where the group runs ncclAllReduce(int32, sum), the identical K2 everywhere.  Everything is enqueued on the first
// sampler's stream, so kernels and exchange stay ordered without host synchronisation.
extern "C" int mmg_selftest_gibbs_shards(mmg_sampler *const *samplers, int n_shards, int n_iter)
{
    if (!samplers || n_shards < 1 || n_iter < 0) return fail(MMG_ERR_ARG, "bad argument");
    for (int i = 0; i < n_shards; ++i) {
        const mmg_sampler *s = samplers[i], *z = samplers[0];
        if (!s) return fail(MMG_ERR_ARG, "NULL sampler");
        if (s->device != z->device) return fail(MMG_ERR_ARG, "the self test runs its shards on one device");
        if (s->p->n != z->p->n || s->cfg.n_chains != z->cfg.n_chains || s->cfg.seed != z->cfg.seed || s->cfg.chain_base != z->cfg.chain_base ||
            s->iter != z->iter || s->cfg.gibbs_iter != z->cfg.gibbs_iter || s->cfg.trace_len != z->cfg.trace_len || s->sampled)
            return fail(MMG_ERR_ARG, "read shards of one chain need the same transcripts, chains, seed, chain_base, iteration, gibbs_iter and trace_len");
    }
    HIP_TRY(hipSetDevice(samplers[0]->device));
    std::vector<hipStream_t> saved(n_shards);
    for (int i = 0; i < n_shards; ++i) { HIP_TRY(hipStreamSynchronize(samplers[i]->cur)); saved[i] = samplers[i]->cur; samplers[i]->cur = samplers[0]->cur; }
    const hipStream_t st = samplers[0]->cur;
    const size_t count = (size_t)samplers[0]->cfg.n_chains * samplers[0]->p->n;
    int rc = MMG_OK;
    for (int it = 0; it < n_iter && rc == MMG_OK; ++it) {
        for (int i = 0; i < n_shards && rc == MMG_OK; ++i) rc = mmg_sampler_sample(samplers[i]);             // src/mmseq.cpp:857-891 on the shard's rows
        for (int i = 1; i < n_shards && rc == MMG_OK; ++i) {                                                  // :896-899 across shards
            launch_add_i32(samplers[0]->d_cnt, samplers[i]->d_cnt, count, st);
            if (hipGetLastError() != hipSuccess) rc = fail(MMG_ERR_HIP, "count exchange");
        }
        for (int i = 1; i < n_shards && rc == MMG_OK; ++i)
            if (hipMemcpyAsync(samplers[i]->d_cnt, samplers[0]->d_cnt, count * sizeof(int32_t), hipMemcpyDeviceToDevice, st) != hipSuccess) rc = fail(MMG_ERR_HIP, "count exchange");
        for (int i = 0; i < n_shards && rc == MMG_OK; ++i) rc = mmg_sampler_update(samplers[i]);             // :905-917, identical everywhere
    }
    const hipError_t e = hipStreamSynchronize(st);
    for (int i = 0; i < n_shards; ++i) samplers[i]->cur = saved[i];
    if (rc == MMG_OK && e != hipSuccess) rc = fail(MMG_ERR_HIP, std::string("gibbs shards: ") + hipGetErrorString(e));
    return rc;
}

extern "C" int mmg_sampler_counts_devptr(mmg_sampler *s, void **ptr, uint64_t *count)
{
    if (!s || !ptr) return fail(MMG_ERR_ARG, "NULL argument");
    *ptr = s->d_cnt;
    if (count) *count = (uint64_t)s->cfg.n_chains * s->p->n;
    return MMG_OK;
}

extern "C" int mmg_sampler_moments_devptr(mmg_sampler *s, void **ptr, uint64_t *count)
{
    if (!s || !ptr) return fail(MMG_ERR_ARG, "NULL argument");
    *ptr = s->d_mom;
    if (count) *count = 2ull * (uint64_t)s->cfg.n_chains * s->p->n;
    return MMG_OK;
}

extern "C" int mmg_sampler_sync(mmg_sampler *s)
{
    if (!s) return fail(MMG_ERR_ARG, "NULL sampler");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->cur));
    return MMG_OK;
}

// Wait until the first n_done iterations have completed on the device; iterations enqueued behind them are not waited for (a caller
// that streams the trace out enqueues the next stretch BEFORE it waits for this one: the device never idles while the host writes).
extern "C" int mmg_sampler_wait_iterations(mmg_sampler *s, int n_done)
{
    if (!s) return fail(MMG_ERR_ARG, "NULL sampler");
    if (n_done < 0 || n_done > s->iter) return fail(MMG_ERR_ARG, "wait for iterations that were not enqueued");
    HIP_TRY(hipSetDevice(s->device));
    // a mark that has fired covers it (marks are recycled as they fire: without this the fallback below would wait for the chunk the
    // caller has just enqueued AHEAD -- the overlap this function exists for)
    if (n_done <= s->fired_upto) return MMG_OK;
    // the first mark at or behind n_done; without one (n_done behind the last mark: marks follow every 16th stored sample) the stream
    size_t k = 0;
    while (k < s->marks.size() && s->marks[k].first < n_done) ++k;
    if (k == s->marks.size()) {
        HIP_TRY(hipStreamSynchronize(s->cur));
        s->fired_upto = s->iter;
    } else {
        HIP_TRY(hipEventSynchronize(s->mark_pool[s->marks[k].second]));
        s->fired_upto = std::max(s->fired_upto, s->marks[k].first);
        ++k;
    }
    for (size_t i = 0; i < k; ++i) { s->mark_free.push_back(s->marks.front().second); s->marks.pop_front(); }
    return MMG_OK;
}

extern "C" int mmg_sampler_iteration(const mmg_sampler *s, int *iter)
{
    if (!s || !iter) return fail(MMG_ERR_ARG, "NULL argument");
    *iter = s->iter;
    return MMG_OK;
}

static int check_chain(const mmg_sampler *s, int chain)
{
    if (!s) return fail(MMG_ERR_ARG, "NULL sampler");
    if (chain < 0 || chain >= s->cfg.n_chains) return fail(MMG_ERR_ARG, "chain index out of range");
    return MMG_OK;
}

extern "C" int mmg_sampler_get_trace(mmg_sampler *s, int chain, double *out)
{
    int rc = check_chain(s, chain);
    if (rc) return rc;
    if (!out) return fail(MMG_ERR_ARG, "NULL out");
    if (!s->d_trace) return fail(MMG_ERR_STATE, "sampler was created with keep_trace == 0");
    HIP_TRY(hipSetDevice(s->device));
    const size_t n = s->p->n, S = (size_t)s->cfg.trace_len;
    double *d_tmp = nullptr;
    HIP_TRY(hipMalloc((void **)&d_tmp, n * S * sizeof(double)));
    launch_transpose(s->d_trace + (size_t)chain * S * n, d_tmp, (uint32_t)n, (uint32_t)S, s->p->d_int_of_ext, s->cur);
    hipError_t e = hipStreamSynchronize(s->cur);
    if (e == hipSuccess) e = hipMemcpy(out, d_tmp, n * S * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d_tmp);
    if (e != hipSuccess) return fail(MMG_ERR_HIP, std::string("get_trace: ") + hipGetErrorString(e));
    return MMG_OK;
}

extern "C" int mmg_sampler_get_trace_rows(mmg_sampler *s, int chain, int first, int count, double *out)
{
    int rc = check_chain(s, chain);
    if (rc) return rc;
    if (!out || first < 0 || count < 0 || (int64_t)first + count > s->cfg.trace_len) return fail(MMG_ERR_ARG, "bad sample range");
    if (!s->d_trace) return fail(MMG_ERR_STATE, "sampler was created with keep_trace == 0");
    HIP_TRY(hipSetDevice(s->device));
    const size_t n = s->p->n, S = (size_t)s->cfg.trace_len;
    const double *src = s->d_trace + ((size_t)chain * S + (size_t)first) * n;
    if (!s->p->renumbered() || count == 0) {
        HIP_TRY(hipStreamSynchronize(s->cur));
        HIP_TRY(hipMemcpy(out, src, (size_t)count * n * sizeof(double), hipMemcpyDeviceToHost));
        return MMG_OK;
    }
    double *d_tmp = nullptr;
    HIP_TRY(hipMalloc((void **)&d_tmp, (size_t)count * n * sizeof(double)));
    launch_gather_rows(src, d_tmp, (uint32_t)n, (uint32_t)count, 8, s->p->d_int_of_ext, s->cur);
    hipError_t e = hipStreamSynchronize(s->cur);
    if (e == hipSuccess) e = hipMemcpy(out, d_tmp, (size_t)count * n * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d_tmp);
    if (e != hipSuccess) return fail(MMG_ERR_HIP, std::string("get_trace_rows: ") + hipGetErrorString(e));
    return MMG_OK;
}

// The same rows for samples the device has FINISHED (the caller knows: it synchronised after the iteration that produced the last of
// them): gathered and copied on a stream of the sampler's own for reading, so the call neither waits for iterations enqueued behind
// those samples nor delays them.  May be called from another thread than the one driving the sampler.
extern "C" int mmg_sampler_get_trace_rows_done(mmg_sampler *s, int chain, int first, int count, double *out)
{
    int rc = check_chain(s, chain);
    if (rc) return rc;
    if (!out || first < 0 || count < 0 || (int64_t)first + count > s->cfg.trace_len) return fail(MMG_ERR_ARG, "bad sample range");
    if (!s->d_trace) return fail(MMG_ERR_STATE, "sampler was created with keep_trace == 0");
    if (count == 0) return MMG_OK;
    HIP_TRY(hipSetDevice(s->device));
    std::lock_guard<std::mutex> lock(s->reader_mu);
    if (!s->reader) HIP_TRY(hipStreamCreateWithFlags(&s->reader, hipStreamNonBlocking));
    const size_t n = s->p->n, S = (size_t)s->cfg.trace_len;
    const double *src = s->d_trace + ((size_t)chain * S + (size_t)first) * n;
    if (s->p->renumbered()) {
        if (s->reader_cap < (size_t)count * n) {
            if (s->d_reader_tmp) (void)hipFree(s->d_reader_tmp);
            s->d_reader_tmp = nullptr; s->reader_cap = 0;
            HIP_TRY(hipMalloc((void **)&s->d_reader_tmp, (size_t)count * n * sizeof(double)));
            s->reader_cap = (size_t)count * n;
        }
        launch_gather_rows(src, s->d_reader_tmp, (uint32_t)n, (uint32_t)count, 8, s->p->d_int_of_ext, s->reader);
        src = s->d_reader_tmp;
    }
    const hipError_t e = s->reader_stage.copy_out(out, src, (size_t)count * n * sizeof(double), s->reader);
    if (e != hipSuccess) return fail(MMG_ERR_HIP, std::string("get_trace_rows_done: ") + hipGetErrorString(e));
    return MMG_OK;
}

extern "C" int mmg_sampler_get_mu(mmg_sampler *s, int chain, double *mu)
{
    int rc = check_chain(s, chain);
    if (rc) return rc;
    if (!mu) return fail(MMG_ERR_ARG, "NULL out");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->cur));
    return download_ext(s->p, s->d_mu + (size_t)chain * s->p->n, mu);
}

extern "C" int mmg_sampler_get_counts(mmg_sampler *s, int chain, int32_t *cnt)
{
    int rc = check_chain(s, chain);
    if (rc) return rc;
    if (!cnt) return fail(MMG_ERR_ARG, "NULL out");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->cur));
    // between sample() and update() the live counts are the interesting ones
    const int32_t *src = (s->sampled ? s->d_cnt : s->d_cnt_last) + (size_t)chain * s->p->n;
    return download_ext(s->p, src, cnt);
}

extern "C" int mmg_sampler_get_moments(mmg_sampler *s, int chain, double *sum_log, double *sum_log2, int64_t *n_samples)
{
    int rc = check_chain(s, chain);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->cur));
    const size_t C = (size_t)s->cfg.n_chains, n = s->p->n;
    if (sum_log && (rc = download_ext(s->p, s->d_mom + (size_t)chain * n, sum_log)) != MMG_OK) return rc;
    if (sum_log2 && (rc = download_ext(s->p, s->d_mom + (C + (size_t)chain) * n, sum_log2)) != MMG_OK) return rc;
    if (n_samples) *n_samples = s->n_kept;
    return MMG_OK;
}

static int drain_events(mmg_sampler *s)
{
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->cur));
    return ev_harvest(s, true);
}

extern "C" int mmg_sampler_get_timing(mmg_sampler *s, mmg_timing *t)
{
    if (!s || !t) return fail(MMG_ERR_ARG, "NULL argument");
    int rc = drain_events(s);
    if (rc) return rc;
    t->sample_ms = s->acc_sample_ms; t->update_ms = s->acc_update_ms;
    t->sample_launches = s->acc_sample_n; t->update_launches = s->acc_update_n;
    return MMG_OK;
}

extern "C" int mmg_sampler_reset_timing(mmg_sampler *s)
{
    if (!s) return fail(MMG_ERR_ARG, "NULL sampler");
    int rc = drain_events(s);
    if (rc) return rc;
    s->acc_sample_ms = s->acc_update_ms = 0; s->acc_sample_n = s->acc_update_n = 0;
    return MMG_OK;
}

extern "C" void mmg_sampler_destroy(mmg_sampler *s) { sampler_free(s); }
