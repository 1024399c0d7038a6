// mmg_host.h -- shared by the host translation units of libmmgibbs (mmgibbs.hip, sampler.hip, em_host.hip, post_host.hip):
// the problem handle, error plumbing, caller <-> device transcript numbering.
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#include "../../include/mmgibbs.h"
#include "mmg_types.h"

struct mmg_problem {
    int device = 0;
    uint64_t m = 0, nnz = 0, total_k = 0, row_id_base = 0, device_bytes = 0;
    uint64_t total_k_hit = 0;                   // the reads of the rows with at least one hit: what a sweep allocates (an empty row gets no count)
    uint32_t n = 0, max_row_len = 0;
    bool idx64 = false;
    int cu_count = 256;
    int layout = MMG_LAYOUT_CANONICAL;
    void *d_row_ptr = nullptr;
    uint32_t *d_col = nullptr;
    uint32_t *d_k = nullptr;
    double *d_l = nullptr;                      // device numbering
    std::vector<double> h_l;                    // caller numbering
    // transcript renumbering (tx_order): empty / nullptr = identity
    std::vector<uint32_t> h_int_of_ext, h_ext_of_int;
    uint32_t *d_int_of_ext = nullptr, *d_ext_of_int = nullptr;
    // sliced-ELL stream of k_sample_sell / k_em_sell
    uint8_t *d_sell = nullptr;
    uint64_t sell_bytes = 0, n_sell_tiles = 0, n_fast_tiles = 0, n_far_tiles = 0, padded_slots = 0;
    mmg::SellTile *d_sell_tiles = nullptr;
    uint64_t *d_sell_chunk = nullptr;
    int grid_sell = 0;
    // problems with multiplicities: the SELL_HASK tiles and the others as two descriptor lists with their own ranges (null / 0: no
    // SELL_HASK tile -- the one launch over d_sell_tiles does everything)
    mmg::SellTile *d_sell_tiles_1 = nullptr, *d_sell_tiles_k = nullptr;
    uint64_t *d_sell_chunk_k = nullptr;
    int grid_sell_k = 0;
    uint64_t n_hask_tiles = 0;
    // the rows on the conditional-binomial chain (mmg_types.h: bigk_row): their stored positions, ascending -- sampled by k_sample_bigk in
    // pieces of bigk_per_wave list entries per workgroup (the tile kernel skips them); h_bigk_list: the host's copy (the timed shard cut)
    uint64_t *d_bigk_list = nullptr;
    uint64_t n_bigk = 0;
    uint32_t bigk_per_wave = 0;
    int grid_bigk = 0;
    std::vector<uint64_t> h_bigk_list;
    // chains advanced in pairs: k_sample_sell_multi walks the register-path tiles without multiplicities (d_sell_tiles_f; the whole
    // list when every tile is like that) in its own ranges (2 and 4 chains: fewer resident waves); the other tiles without
    // multiplicities -- far tiles, CSR-walked tiles -- are a third list (d_sell_tiles_x) that k_sample_sell walks for all the
    // chains of the sampler in ONE launch (grid.y = chain), and so are the SELL_HASK tiles
    mmg::SellTile *d_sell_tiles_f = nullptr, *d_sell_tiles_x = nullptr; // f may alias d_sell_tiles / d_sell_tiles_1 (not owned then)
    bool owns_tiles_f = false;
    uint64_t *d_sell_chunk_m[2] = {nullptr, nullptr}, *d_sell_chunk_x = nullptr;
    int grid_sell_m[2] = {0, 0}, grid_sell_x = 0;
    uint64_t n_x_tiles = 0;
    bool use_sell = false;
    std::vector<uint64_t> h_sell_cum;           // cumulative tile cost, kept for the EM kernel's own ranges
    // read shards (mmg_problem_shard_bounds): first row of every tile and the cumulative modelled cost of a sweep over the tiles
    // [0, t) -- multiplicities included, a register-path tile priced by its groups (the stream bounds K1: time follows bytes)
    std::vector<uint64_t> h_tile_row, h_shard_cum;
    // the tile lists of the one-chain sample launches as the host built them (mmg_problem_shard_bounds_timed launches the kernels over
    // sub-intervals of them): entry e of the main list is tile h_list1_tile[e] (empty: the identity) and costs h_cum1[e + 1] - h_cum1[e]
    // in the units of the launch's ranges; the same for the list of multiplicity tiles
    std::vector<uint64_t> h_cum1, h_cumk;
    std::vector<uint32_t> h_list1_tile, h_listk_tile;
    uint64_t resident1 = 0, residentk = 0;      // workgroups of the two kernels the device holds at once
    uint32_t cnt_replicas = 1;                  // replicas of a sampler's count vectors (mmg_types.h: CNT_REPLICAS), 1 or CNT_REPLICAS
    bool canonical_rows = false;                // the rows are in canonical order: a canonical problem, or a shard cut from one
    // CSR tiles of the fallback kernel k_sample (built only when the sliced-ELL stream is not used)
    mmg::TileDesc *d_tiles = nullptr;
    uint64_t n_tiles = 0;
    uint64_t *d_chunk_tile = nullptr;
    int grid_sample = 1;
    uint64_t *d_colcnt = nullptr;               // hits per transcript, for the EM scale words (lazy)
    bool order_derived = false;                 // the renumbering came from the hit graph (order.hip), not from the caller's tx_order
    bool k1_fixed_walk = false;    // the k = 1 sample kernel's straight-line instantiation (fewer than 5 groups per register-path tile on average)
    bool groups_reordered = false; // tx_order given and the library reordered its groups (spec version 7)
    bool order_skipped = false;    // an order from the hit graph was called for but not tried (device memory) or could not be built: the caller's order stands
    bool renumbered() const { return !h_int_of_ext.empty(); }
};

struct mmg_sampler;
struct mmg_em;

namespace mmg {

// Device-to-host copies into the caller's (pageable) memory through two pinned buffers the handle keeps: the runtime would otherwise
// pin and unpin the destination pages for every copy, and a caller that fetches trace rows while the chain runs -- and while its other
// threads allocate and write files -- had the chain's kernels stall for 20-40 ms at a time around those (un)pinnings (the 50 M-read
// CLI run: the chain took 4.5 s instead of 3.9 s).  One copy at a time per stage (the owner's mutex).
struct PinnedStage {
    static constexpr size_t CHUNK = 16u << 20;
    void *buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    ~PinnedStage()
    {
        for (int i = 0; i < 2; ++i) { if (buf[i]) (void)hipHostFree(buf[i]); if (ev[i]) (void)hipEventDestroy(ev[i]); }
    }
    hipError_t copy_out(void *dst, const void *dsrc, size_t bytes, hipStream_t st)
    {
        for (int i = 0; i < 2; ++i) {
            if (!buf[i]) { hipError_t e = hipHostMalloc(&buf[i], CHUNK, hipHostMallocDefault); if (e != hipSuccess) { buf[i] = nullptr; return e; } }
            if (!ev[i]) { hipError_t e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming); if (e != hipSuccess) { ev[i] = nullptr; return e; } }
        }
        const size_t chunks = (bytes + CHUNK - 1) / CHUNK;
        auto issue = [&](size_t c) {
            const size_t off = c * CHUNK, len = bytes - off < CHUNK ? bytes - off : CHUNK;
            hipError_t e = hipMemcpyAsync(buf[c & 1], (const char *)dsrc + off, len, hipMemcpyDeviceToHost, st);
            return e == hipSuccess ? hipEventRecord(ev[c & 1], st) : e;
        };
        if (chunks) { hipError_t e = issue(0); if (e != hipSuccess) return e; }
        for (size_t c = 0; c < chunks; ++c) {
            if (c + 1 < chunks) { hipError_t e = issue(c + 1); if (e != hipSuccess) return e; } // (its buffer was emptied an iteration ago)
            hipError_t e = hipEventSynchronize(ev[c & 1]);
            if (e != hipSuccess) return e;
            const size_t off = c * CHUNK, len = bytes - off < CHUNK ? bytes - off : CHUNK;
            std::memcpy((char *)dst + off, buf[c & 1], len);
        }
        return hipSuccess;
    }
};

// what the summary code needs to see of a sampler (sampler.hip)
struct SamplerView {
    const mmg_problem *p;
    mmg_config cfg;
    const double *d_trace;   // [C][trace_len][n] sample-major, device numbering; nullptr without keep_trace
    hipStream_t stream;
    int iter;
    int64_t n_kept;          // samples kept so far
};
int sampler_view(mmg_sampler *s, SamplerView *v);

// read shards of one problem as one EM (em_host.hip): make_reduce(members, ctx) returns the exchange called between the phases with
// what = 0 (xe: max, int32), 1 (accumulators + log-likelihood limbs: sum, uint64), 2 (column counts: sum, uint64); the buffers of a
// member come from em_exchange_buffers
int em_create_sharded(const mmg_problem *const *shards, int n_shards, const double *mu0,
                      std::function<int(int)> (*make_reduce)(const std::vector<mmg_em *> &, void *), void *ctx, mmg_em **ems, double *loglik0);
void em_exchange_buffers(mmg_em *e, int what, void **ptr, size_t *count);
int em_device(const mmg_em *e);

int fail(int code, const std::string &msg);   // records the thread-local message behind mmg_last_error(), returns code
int opt(int option);                          // mmg_selftest_option value, -1 = default
int require_device(int device);
void weighted_chunks(const std::vector<uint64_t> &cum, uint64_t grid, std::vector<uint64_t> &chunk);
void weighted_chunks_tapered(const std::vector<uint64_t> &cum, uint64_t grid, uint64_t resident, std::vector<uint64_t> &chunk);

#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess)                                                                                 \
            return mmg::fail(MMG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                 \
    } while (0)

// caller numbering <-> device numbering of per-transcript host arrays
template <typename T>
inline void to_int(const mmg_problem *p, const T *ext, std::vector<T> &out)
{
    out.resize(p->n);
    if (p->renumbered()) for (uint32_t i = 0; i < p->n; ++i) out[i] = ext[p->h_ext_of_int[i]];
    else std::memcpy(out.data(), ext, p->n * sizeof(T));
}
template <typename T>
inline void to_ext(const mmg_problem *p, const std::vector<T> &in, T *ext)
{
    if (p->renumbered()) for (uint32_t t = 0; t < p->n; ++t) ext[t] = in[p->h_int_of_ext[t]];
    else std::memcpy(ext, in.data(), p->n * sizeof(T));
}
template <typename T>
inline int download_ext(const mmg_problem *p, const T *d_src, T *ext)
{
    if (!p->renumbered()) { HIP_TRY(hipMemcpy(ext, d_src, p->n * sizeof(T), hipMemcpyDeviceToHost)); return MMG_OK; }
    std::vector<T> tmp(p->n);
    HIP_TRY(hipMemcpy(tmp.data(), d_src, p->n * sizeof(T), hipMemcpyDeviceToHost));
    to_ext(p, tmp, ext);
    return MMG_OK;
}

} // namespace mmg
