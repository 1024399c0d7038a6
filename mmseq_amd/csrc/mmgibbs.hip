// mmgibbs.hip -- implementation of the C ABI in include/mmgibbs.h on HIP / gfx950.
// Host side of the device boundary that replaces src/mmseq.cpp:833-925 of the reference.
#include "../../include/mmgibbs.h"
#include "gibbs_kernels.h"
#include "sell_kernels.h"
#include "em_kernels.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace mmg;

// ------------------------------------------------------------------------------ K1 variants
// Tile geometry variants.  Each names the caps the host tile builder uses and the k_sample instance
// (u32 stream) that serves as fallback; k1_s16_kernel() maps the same id to the k_sample16 instance.
// elems: hits per tile (LDS staging), win: LDS window width, unr: walk unroll of k_sample, bs: workgroup
// size, rows: rows per tile.  MMG_K1_VARIANT selects one (experiments); 0 is the default.
struct K1Variant { int elems, win, unr, mode, bs, rows; };
#define K1_VARIANT_LIST(X) \
    X(0, 2560, 256, 4, 0, 128, 128) \
    X(1, 4096, 512, 4, 0, 256, 256) \
    X(2, 2560, 512, 4, 0, 128, 128) \
    X(3, 3072, 256, 4, 0, 128, 128) \
    X(4, 1280, 256, 4, 0, 64, 64)   \
    X(5, 2560, 256, 4, K1M_NO_PHASE2, 128, 128)
static const K1Variant k1_variants[] = {
#define X(id, e, w, u, m, bs, rows) {e, w, u, m, bs, rows},
    K1_VARIANT_LIST(X)
#undef X
};
static const int K1_DEFAULT_VARIANT = 0;

template <typename IdxT, bool HAS_K>
static const void *k1_kernel_for(int variant)
{
    switch (variant) {
#define X(id, e, w, u, m, bs, rows) case id: return (const void *)&k_sample<IdxT, HAS_K, e, w, u, m, bs, rows>;
        K1_VARIANT_LIST(X)
#undef X
    }
    return nullptr;
}
static const void *k1_kernel(int variant, bool idx64, bool has_k)
{
    if (idx64) return has_k ? k1_kernel_for<uint64_t, true>(variant) : k1_kernel_for<uint64_t, false>(variant);
    return has_k ? k1_kernel_for<uint32_t, true>(variant) : k1_kernel_for<uint32_t, false>(variant);
}

// 16-bit-stream kernel instances, keyed by the k_sample variant whose tile caps they share
template <typename IdxT, bool HAS_K>
static const void *k1_s16_kernel_for(int variant, int fuse = 1)
{
    if (variant == 0 && fuse == 2) return (const void *)&k_sample16<IdxT, HAS_K, 2560, 256, 128, 128, 0, 2>;
    if (variant == 0 && fuse == 4) return (const void *)&k_sample16<IdxT, HAS_K, 2560, 256, 128, 128, 0, 4>;
    if (variant == 0 && fuse == 8) return (const void *)&k_sample16<IdxT, HAS_K, 2560, 256, 128, 128, 0, 8>;
    if (fuse != 1) return nullptr;
    switch (variant) {
    case 0: return (const void *)&k_sample16<IdxT, HAS_K, 2560, 256, 128, 128, 0>;
    case 1: return (const void *)&k_sample16<IdxT, HAS_K, 4096, 512, 256, 256, 0>;
    case 2: return (const void *)&k_sample16<IdxT, HAS_K, 2560, 512, 128, 128, 0>;
    case 3: return (const void *)&k_sample16<IdxT, HAS_K, 3072, 256, 128, 128, 0>;
    case 4: return (const void *)&k_sample16<IdxT, HAS_K, 1280, 256, 64, 64, 0>;
    case 5: return (const void *)&k_sample16<IdxT, HAS_K, 2560, 256, 128, 128, K1M_NO_PHASE2>; // ablation (timing only)
    }
    return nullptr;
}
static const void *k1_s16_kernel(int variant, bool idx64, bool has_k, int fuse = 1)
{
    if (idx64) return has_k ? k1_s16_kernel_for<uint64_t, true>(variant, fuse) : k1_s16_kernel_for<uint64_t, false>(variant, fuse);
    return has_k ? k1_s16_kernel_for<uint32_t, true>(variant, fuse) : k1_s16_kernel_for<uint32_t, false>(variant, fuse);
}

static const int k1_n_variants = (int)(sizeof(k1_variants) / sizeof(k1_variants[0]));

// ------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess)                                                                                 \
            return fail(MMG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                      \
    } while (0)

extern "C" const char *mmg_last_error(void) { return g_err.c_str(); }
extern "C" int mmg_abi_version(void) { return MMG_ABI_VERSION; }

extern "C" int mmg_device_count(int *count)
{
    if (!count) return fail(MMG_ERR_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return MMG_OK;
}

static int require_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(MMG_ERR_NO_DEVICE, "no HIP device available: libmmgibbs has no CPU fallback");
    }
    if (device < 0 || device >= n) return fail(MMG_ERR_ARG, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    return MMG_OK;
}

// ------------------------------------------------------------------------------ problem
struct mmg_problem {
    int device = 0;
    uint64_t m = 0, nnz = 0, total_k = 0, row_id_base = 0, n_tiles = 0, device_bytes = 0;
    uint32_t n = 0, max_row_len = 0;
    bool idx64 = false;
    int cu_count = 256;
    void *d_row_ptr = nullptr;
    uint32_t *d_col = nullptr;
    uint32_t *d_k = nullptr;
    double *d_l = nullptr;
    TileDesc *d_tiles = nullptr;
    uint16_t *d_stream16 = nullptr; // 16-bit tile stream of k_sample16 (built by k_encode16)
    uint64_t stream16_bytes = 0;
    S16Tile *d_s16tiles = nullptr;
    uint64_t *d_chunk_tile16 = nullptr;
    int grid16 = 0;
    bool use16 = false;
    double s16_fast_fraction = 0.0;
    uint64_t *d_colcnt = nullptr;   // hits per transcript, for the EM scale words (lazy)
    // SELL-64 stream of k_sample_sell
    uint8_t *d_sell = nullptr;
    uint64_t sell_bytes = 0, n_sell_tiles = 0;
    SellTile *d_sell_tiles = nullptr;
    uint64_t *d_sell_chunk = nullptr;
    int grid_sell = 0;
    bool use_sell = false;
    double sell_fast_fraction = 0.0;
    std::vector<uint64_t> h_sell_cum; // cumulative tile cost (weighted_chunks), kept for the EM kernels' own ranges
    std::vector<uint8_t> h_sell_ng;   // per 64-row tile: groups of the longest row (0: empty, 255: too long); consumed by problem_build_desc
    uint64_t *d_chunk_tile = nullptr;
    uint64_t n_chunks = 0;
    int grid_sample = 1;
    int variant = K1_DEFAULT_VARIANT;
    std::vector<double> h_l;
    std::vector<uint64_t> h_tile_row; // consumed by problem_build_desc
};

static void problem_free(mmg_problem *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->d_row_ptr) (void)hipFree(p->d_row_ptr);
    if (p->d_col) (void)hipFree(p->d_col);
    if (p->d_k) (void)hipFree(p->d_k);
    if (p->d_l) (void)hipFree(p->d_l);
    if (p->d_tiles) (void)hipFree(p->d_tiles);
    if (p->d_stream16) (void)hipFree(p->d_stream16);
    if (p->d_s16tiles) (void)hipFree(p->d_s16tiles);
    if (p->d_chunk_tile16) (void)hipFree(p->d_chunk_tile16);
    if (p->d_colcnt) (void)hipFree(p->d_colcnt);
    if (p->d_sell) (void)hipFree(p->d_sell);
    if (p->d_sell_tiles) (void)hipFree(p->d_sell_tiles);
    if (p->d_sell_chunk) (void)hipFree(p->d_sell_chunk);
    if (p->d_chunk_tile) (void)hipFree(p->d_chunk_tile);
    delete p;
}

// Tiles of consecutive rows: <= tile_nnz hits and <= tile_rows rows; a longer row is alone.
static void build_tiles(const uint64_t *row_ptr, uint64_t m, uint64_t tile_nnz, uint64_t tile_rows, std::vector<uint64_t> &tile_row,
                        uint32_t &max_len)
{
    tile_row.clear();
    tile_row.push_back(0);
    uint64_t cur_nnz = 0, cur_rows = 0;
    max_len = 0;
    for (uint64_t r = 0; r < m; ++r) {
        const uint64_t L = row_ptr[r + 1] - row_ptr[r], L4 = (L + 3) & ~(uint64_t)3; // the 16-bit stream pads rows to 4 hits
        if (L > max_len) max_len = (uint32_t)std::min<uint64_t>(L, 0xffffffffu);
        if (cur_rows > 0 && (cur_nnz + L4 > tile_nnz || cur_rows >= tile_rows)) {
            tile_row.push_back(r);
            cur_nnz = 0;
            cur_rows = 0;
        }
        cur_nnz += L4;
        cur_rows += 1;
    }
    if (m > 0) tile_row.push_back(m);
}

// uploads row_ptr (narrowed to u32 when nnz fits), tiles; fills sizes
static int problem_finish(mmg_problem *p, const uint64_t *h_row_ptr)
{
    std::vector<uint64_t> tiles;
    if (const char *ev = getenv("MMG_K1_VARIANT")) {
        const int v = atoi(ev);
        if (v < 0 || v >= k1_n_variants) return fail(MMG_ERR_ARG, "MMG_K1_VARIANT out of range");
        p->variant = v;
    }
    {
        const K1Variant &kv = k1_variants[p->variant];
        const uint64_t rows_cap = kv.rows > 0 ? std::min<uint64_t>(kv.rows, kv.elems / 4) : (uint64_t)kv.elems / 4;
        build_tiles(h_row_ptr, p->m, (uint64_t)kv.elems - 8, rows_cap, tiles, p->max_row_len);
    }
    p->n_tiles = tiles.empty() ? 0 : tiles.size() - 1;
    p->idx64 = p->nnz >= 0xffffffffull || getenv("MMG_FORCE_IDX64") != nullptr; // the env knob lets small tests cover the 64-bit path
    if (p->idx64) {
        HIP_TRY(hipMalloc(&p->d_row_ptr, (p->m + 1) * sizeof(uint64_t)));
        HIP_TRY(hipMemcpy(p->d_row_ptr, h_row_ptr, (p->m + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
        p->device_bytes += (p->m + 1) * 8;
    } else {
        std::vector<uint32_t> rp32(p->m + 1);
        for (uint64_t i = 0; i <= p->m; ++i) rp32[i] = (uint32_t)h_row_ptr[i];
        HIP_TRY(hipMalloc(&p->d_row_ptr, (p->m + 1) * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(p->d_row_ptr, rp32.data(), (p->m + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
        p->device_bytes += (p->m + 1) * 4;
    }
    p->h_tile_row.swap(tiles);
    if (p->m) { // SELL-64 tiling: fixed 64-row slices, groups of the longest row
        const uint64_t nt64 = (p->m + 63) / 64;
        p->h_sell_ng.assign(nt64, 0);
        for (uint64_t t = 0; t < nt64; ++t) {
            uint64_t mx = 0;
            const uint64_t r1 = std::min<uint64_t>(p->m, (t + 1) * 64);
            for (uint64_t r = t * 64; r < r1; ++r) mx = std::max<uint64_t>(mx, h_row_ptr[r + 1] - h_row_ptr[r]);
            p->h_sell_ng[t] = mx > 255 ? 255 : (uint8_t)((mx + 3) / 4);
        }
    }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, p->device));
    p->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // Persistent grid: every resident workgroup (4 per CU at 40 KiB LDS) walks q contiguous
    // chunks of tiles, so its LDS window slides monotonically over the sorted rows.
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k1_kernel(p->variant, p->idx64, false), k1_variants[p->variant].bs, 0) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 4;
    }
    const int max_blocks = 2048 / k1_variants[p->variant].bs > 32 ? 32 : 2048 / k1_variants[p->variant].bs; // 32 waves per CU
    if (per_cu > max_blocks) per_cu = max_blocks;
    if (const char *ev = getenv("MMG_K1_BLOCKS_PER_CU")) { const int v = atoi(ev); if (v >= 1 && v <= 32) per_cu = v; }
    const uint64_t resident = (uint64_t)p->cu_count * (uint64_t)per_cu;
    const uint64_t grid = std::max<uint64_t>(1, std::min<uint64_t>(p->n_tiles, resident));
    p->n_chunks = p->n_tiles ? grid : 0; // one contiguous tile range per workgroup
    p->grid_sample = (int)grid;
    std::vector<uint64_t> chunk(p->n_chunks + 1, 0);
    for (uint64_t c = 0; c <= p->n_chunks; ++c)
        chunk[c] = p->n_chunks ? (uint64_t)(((unsigned __int128)p->n_tiles * c) / p->n_chunks) : 0;
    HIP_TRY(hipMalloc((void **)&p->d_chunk_tile, chunk.size() * sizeof(uint64_t)));
    HIP_TRY(hipMemcpy(p->d_chunk_tile, chunk.data(), chunk.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    return MMG_OK;
}

// per-tile descriptors, built on the device from the RESIDENT CSR (col_idx must be final)
// Contiguous tile ranges of (nearly) equal COST: cum[t] = cost of tiles [0, t).  A tile that cannot run on the register path
// is walked from the CSR and costs many times more; with equal tile counts a tail of such tiles (wide rows are sorted
// last) lands on a few workgroups that finish long after the rest.
static void weighted_chunks(const std::vector<uint64_t> &cum, uint64_t grid, std::vector<uint64_t> &chunk)
{
    const uint64_t nt = cum.size() - 1, total = cum[nt];
    chunk.assign(grid + 1, 0);
    uint64_t t = 0;
    for (uint64_t c = 1; c < grid; ++c) {
        const uint64_t target = (uint64_t)(((unsigned __int128)total * c) / grid);
        while (t < nt && cum[t] < target) ++t;
        chunk[c] = t;
    }
    chunk[grid] = nt;
}
constexpr uint64_t SELL_SLOW_TILE_COST = 24; // measured: a CSR-walked tile against a register-path tile

static const void *k1_sell_kernel(bool idx64, bool has_k)
{
    if (!idx64 && !has_k) {
        if (const char *rp = getenv("MMG_K1_SELL_REP")) { // count replicas (experiments)
            switch (atoi(rp)) {
            case 4: return (const void *)k_sample_sell<uint32_t, false, 8, 4>;
            case 2: return (const void *)k_sample_sell<uint32_t, false, 8, 2>;
            case 8: return (const void *)k_sample_sell<uint32_t, false, 8, 8>;
            }
        }
    }
    if (idx64) return has_k ? (const void *)k_sample_sell<uint64_t, true, 8> : (const void *)k_sample_sell<uint64_t, false, 8>;
    return has_k ? (const void *)k_sample_sell<uint32_t, true, 8> : (const void *)k_sample_sell<uint32_t, false, 8>;
}

// SELL-64 stream for k_sample_sell: tiles of 64 rows, the window policy of the 16-bit stream, one block per tile.
static int problem_build_sell(mmg_problem *p)
{
    std::vector<uint8_t> ngs;
    ngs.swap(p->h_sell_ng);
    const char *ev = getenv("MMG_K1_SELL");
    if (ngs.empty() || (ev && atoi(ev) == 0)) return MMG_OK;
    const uint64_t nt = ngs.size();
    const uint32_t WIN = SELL_WIN;
    std::vector<uint64_t> tile_row(nt + 1);
    for (uint64_t t = 0; t <= nt; ++t) tile_row[t] = std::min<uint64_t>(p->m, t * 64);
    uint64_t *d_tile_row = nullptr;
    TileDesc *d_td = nullptr;
    auto cleanup = [&]() { if (d_tile_row) (void)hipFree(d_tile_row); if (d_td) (void)hipFree(d_td); };
#define SELL_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); return fail(MMG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } } while (0)
    SELL_TRY(hipMalloc((void **)&d_tile_row, tile_row.size() * sizeof(uint64_t)));
    SELL_TRY(hipMemcpy(d_tile_row, tile_row.data(), tile_row.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    SELL_TRY(hipMalloc((void **)&d_td, nt * sizeof(TileDesc)));
    if (p->idx64) hipLaunchKernelGGL(k_tile_desc<uint64_t>, dim3((unsigned)nt), dim3(64), 0, 0, (const uint64_t *)p->d_row_ptr, p->d_col, d_tile_row, nt, d_td);
    else hipLaunchKernelGGL(k_tile_desc<uint32_t>, dim3((unsigned)nt), dim3(64), 0, 0, (const uint32_t *)p->d_row_ptr, p->d_col, d_tile_row, nt, d_td);
    SELL_TRY(hipGetLastError());
    std::vector<TileDesc> td(nt);
    SELL_TRY(hipMemcpy(td.data(), d_td, nt * sizeof(TileDesc), hipMemcpyDeviceToHost));
    cleanup();
    d_tile_row = nullptr; d_td = nullptr;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k1_sell_kernel(p->idx64, p->d_k != nullptr), 64, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 16; }
    if (per_cu > 32) per_cu = 32;
    if (const char *e2 = getenv("MMG_K1_SELL_WAVES_PER_CU")) { const int v = atoi(e2); if (v >= 1 && v <= 32) per_cu = v; }
    const uint64_t grid = std::max<uint64_t>(1, std::min<uint64_t>(nt, (uint64_t)p->cu_count * per_cu));
    // cost estimate before the window policy runs: a tile whose columns span more than the window cannot qualify
    p->h_sell_cum.assign(nt + 1, 0);
    for (uint64_t t = 0; t < nt; ++t) {
        const bool slow = td[t].nnz && ((uint64_t)td[t].cmax - td[t].call >= WIN || ngs[t] > 64 || ngs[t] == 0);
        p->h_sell_cum[t + 1] = p->h_sell_cum[t] + (td[t].nnz == 0 ? 0 : slow ? SELL_SLOW_TILE_COST : 1);
    }
    std::vector<uint64_t> chunk;
    weighted_chunks(p->h_sell_cum, grid, chunk);
    std::vector<SellTile> st(nt);
    uint64_t n_fast = 0, n_live = 0, pos = 0;
    for (uint64_t c = 0; c < grid; ++c) {
        bool have = false;
        uint32_t cur = 0;
        for (uint64_t t = chunk[c]; t < chunk[c + 1]; ++t) if (td[t].nnz) { cur = td[t].cmin & ~15u; break; }
        for (uint64_t t = chunk[c]; t < chunk[c + 1]; ++t) {
            const TileDesc &d = td[t];
            SellTile &q = st[t];
            q.off16 = 0; q.r0 = d.r0; q.meta = sell_meta(d.nrows, 0, 0);
            if (d.nnz == 0) { q.meta = sell_meta(d.nrows, 0, S16_EMPTY); q.wbase = cur; continue; }
            const bool keep = have && d.cmin >= cur && (uint64_t)d.clast + K1_WIN_MARGIN <= (uint64_t)cur + WIN;
            if (!keep) { cur = d.cmin & ~15u; have = true; }
            q.wbase = cur;
            const bool inwin = d.call >= cur && (uint64_t)d.cmax < (uint64_t)cur + WIN;
            if (inwin && ngs[t] >= 1 && ngs[t] <= 64) { // rows of at most 255 hits (the length byte), all columns inside the window
                q.meta = sell_meta(d.nrows, ngs[t], S16_FAST);
                q.off16 = pos;
                pos += 4 + 16 * (uint64_t)ngs[t];
                ++n_fast;
            }
            ++n_live;
        }
    }
    p->sell_fast_fraction = n_live ? (double)n_fast / (double)n_live : 0.0;
    // tiles that do not qualify are walked from the CSR inside the same kernel; half the tiles on the register path already
    // beats the fallback kernels (callers sort wide rows last, so the slow tiles are a contiguous tail)
    p->use_sell = p->sell_fast_fraction >= 0.5 || (ev && atoi(ev) == 2);
    if (!p->use_sell) return MMG_OK;
    p->sell_bytes = pos * 16;
    p->n_sell_tiles = nt;
    HIP_TRY(hipMalloc((void **)&p->d_sell, p->sell_bytes + 64));
    HIP_TRY(hipMalloc((void **)&p->d_sell_tiles, nt * sizeof(SellTile)));
    HIP_TRY(hipMemcpy(p->d_sell_tiles, st.data(), nt * sizeof(SellTile), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void **)&p->d_sell_chunk, chunk.size() * sizeof(uint64_t)));
    HIP_TRY(hipMemcpy(p->d_sell_chunk, chunk.data(), chunk.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    if (p->idx64) hipLaunchKernelGGL(k_encode_sell<uint64_t>, dim3((unsigned)nt), dim3(64), 0, 0, (const uint64_t *)p->d_row_ptr, p->d_col, p->d_sell_tiles, nt, p->d_sell);
    else hipLaunchKernelGGL(k_encode_sell<uint32_t>, dim3((unsigned)nt), dim3(64), 0, 0, (const uint32_t *)p->d_row_ptr, p->d_col, p->d_sell_tiles, nt, p->d_sell);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    p->grid_sell = (int)grid;
    p->device_bytes += p->sell_bytes + nt * sizeof(SellTile);
#undef SELL_TRY
    return MMG_OK;
}

static int problem_build_desc(mmg_problem *p)
{
    std::vector<uint64_t> &tiles = p->h_tile_row;
        uint64_t *d_tile_row = nullptr;
        const size_t tb = std::max<size_t>(tiles.size(), 1) * sizeof(uint64_t);
        HIP_TRY(hipMalloc((void **)&d_tile_row, tb));
        hipError_t e = hipSuccess;
        if (!tiles.empty()) e = hipMemcpy(d_tile_row, tiles.data(), tiles.size() * sizeof(uint64_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void **)&p->d_tiles, std::max<uint64_t>(p->n_tiles, 1) * sizeof(TileDesc));
        if (e == hipSuccess && p->n_tiles) {
            if (p->idx64) hipLaunchKernelGGL(k_tile_desc<uint64_t>, dim3((unsigned)p->n_tiles), dim3(64), 0, 0, (const uint64_t *)p->d_row_ptr, p->d_col, d_tile_row, p->n_tiles, p->d_tiles);
            else hipLaunchKernelGGL(k_tile_desc<uint32_t>, dim3((unsigned)p->n_tiles), dim3(64), 0, 0, (const uint32_t *)p->d_row_ptr, p->d_col, d_tile_row, p->n_tiles, p->d_tiles);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipDeviceSynchronize();
        }
        (void)hipFree(d_tile_row);
        if (e != hipSuccess) return fail(MMG_ERR_HIP, std::string("tile descriptors: ") + hipGetErrorString(e));
        p->device_bytes += p->n_tiles * sizeof(TileDesc);
        // ---- 16-bit stream kernel: grid, contiguous tile ranges, per-tile window policy, stream placement, encode
        const void *k16 = k1_s16_kernel(p->variant, p->idx64, p->d_k != nullptr);
        const char *ev16 = getenv("MMG_K1_S16");
        if (p->n_tiles && k16 && !(ev16 && atoi(ev16) == 0)) {
            const K1Variant &kv = k1_variants[p->variant];
            std::vector<TileDesc> td(p->n_tiles);
            HIP_TRY(hipMemcpy(td.data(), p->d_tiles, p->n_tiles * sizeof(TileDesc), hipMemcpyDeviceToHost));
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k16, kv.bs, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 4; }
            if (per_cu > 2048 / kv.bs) per_cu = 2048 / kv.bs;
            if (const char *e2 = getenv("MMG_K1_BLOCKS_PER_CU")) { const int v = atoi(e2); if (v >= 1 && v <= 32) per_cu = v; }
            const uint64_t grid = std::max<uint64_t>(1, std::min<uint64_t>(p->n_tiles, (uint64_t)p->cu_count * per_cu));
            std::vector<uint64_t> chunk(grid + 1);
            for (uint64_t c = 0; c <= grid; ++c) chunk[c] = (uint64_t)(((unsigned __int128)p->n_tiles * c) / grid);
            std::vector<S16Tile> st(p->n_tiles);
            uint64_t n_fast = 0, n_live = 0, pos = 0;
            const uint32_t WIN = (uint32_t)kv.win;
            const uint32_t rows_cap = kv.rows > 0 ? (uint32_t)std::min(kv.rows, kv.elems / 4) : (uint32_t)kv.elems / 4;
            for (uint64_t c = 0; c < grid; ++c) {
                bool have = false;
                uint32_t cur = 0;
                for (uint64_t t = chunk[c]; t < chunk[c + 1]; ++t) if (td[t].nnz) { cur = td[t].cmin & ~15u; break; }
                for (uint64_t t = chunk[c]; t < chunk[c + 1]; ++t) {
                    const TileDesc &d = td[t];
                    S16Tile &q = st[t];
                    q.s16 = 0; q.r0 = d.r0; q.nrows = d.nrows; q.nnz4 = d.nnz4; q.flags = 0;
                    if (d.nnz == 0) { q.flags = S16_EMPTY; q.wbase = cur; continue; }
                    const bool keep = have && d.cmin >= cur && (uint64_t)d.clast + K1_WIN_MARGIN <= (uint64_t)cur + WIN;
                    if (!keep) {
                        const uint32_t nb = d.cmin & ~15u;
                        if (have && nb != cur) q.flags |= S16_SHIFT;
                        cur = nb;
                        have = true;
                    }
                    q.wbase = cur;
                    const bool inwin = d.call >= cur && (uint64_t)d.cmax < (uint64_t)cur + WIN;
                    if (inwin && d.nrows <= rows_cap && d.nnz4 <= (uint32_t)kv.elems && d.nnz4 > 0) {
                        q.flags |= S16_FAST;
                        q.s16 = pos;
                        pos += ((d.nrows + 8) >> 3) + ((d.nnz4 + 7) >> 3);
                        ++n_fast;
                    }
                    ++n_live;
                }
            }
            p->s16_fast_fraction = n_live ? (double)n_fast / (double)n_live : 0.0;
            p->use16 = p->s16_fast_fraction >= 0.9 || (ev16 && atoi(ev16) == 2);
            if (p->use16) {
                p->stream16_bytes = pos * 16;
                HIP_TRY(hipMalloc((void **)&p->d_stream16, p->stream16_bytes + 64));
                HIP_TRY(hipMemset(p->d_stream16, 0, p->stream16_bytes + 64));
                HIP_TRY(hipMalloc((void **)&p->d_s16tiles, p->n_tiles * sizeof(S16Tile)));
                HIP_TRY(hipMemcpy(p->d_s16tiles, st.data(), p->n_tiles * sizeof(S16Tile), hipMemcpyHostToDevice));
                HIP_TRY(hipMalloc((void **)&p->d_chunk_tile16, chunk.size() * sizeof(uint64_t)));
                HIP_TRY(hipMemcpy(p->d_chunk_tile16, chunk.data(), chunk.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
                if (p->idx64) hipLaunchKernelGGL(k_encode16<uint64_t>, dim3((unsigned)p->n_tiles), dim3(64), 0, 0, (const uint64_t *)p->d_row_ptr, p->d_col, p->d_s16tiles, p->n_tiles, WIN, p->d_stream16);
                else hipLaunchKernelGGL(k_encode16<uint32_t>, dim3((unsigned)p->n_tiles), dim3(64), 0, 0, (const uint32_t *)p->d_row_ptr, p->d_col, p->d_s16tiles, p->n_tiles, WIN, p->d_stream16);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipDeviceSynchronize());
                p->grid16 = (int)grid;
                p->device_bytes += p->stream16_bytes + p->n_tiles * sizeof(S16Tile);
            }
        }
        std::vector<uint64_t>().swap(tiles);
    return problem_build_sell(p);
}

extern "C" int mmg_problem_create(const mmg_problem_desc *d, int device, mmg_problem **out)
{
    if (!d || !out) return fail(MMG_ERR_ARG, "NULL argument");
    if (!d->row_ptr || !d->l || d->n == 0) return fail(MMG_ERR_ARG, "row_ptr/l missing or n == 0");
    if (d->row_ptr[0] != 0) return fail(MMG_ERR_ARG, "row_ptr[0] must be 0");
    const uint64_t nnz = d->row_ptr[d->m];
    if (nnz > 0 && !d->col_idx) return fail(MMG_ERR_ARG, "col_idx missing");
    for (uint64_t r = 0; r < d->m; ++r)
        if (d->row_ptr[r + 1] < d->row_ptr[r]) return fail(MMG_ERR_ARG, "row_ptr must be non-decreasing");
    for (uint64_t j = 0; j < nnz; ++j)
        if (d->col_idx[j] >= d->n) return fail(MMG_ERR_ARG, "col_idx entry out of range");
    for (uint32_t t = 0; t < d->n; ++t)
        if (!(d->l[t] > 0.0)) return fail(MMG_ERR_ARG, "l[t] must be > 0 (src/mmseq.cpp:604)");
    int rc = require_device(device);
    if (rc) return rc;
    mmg_problem *p = new mmg_problem();
    p->device = device;
    p->m = d->m; p->n = d->n; p->nnz = nnz; p->row_id_base = d->row_id_base;
    p->h_l.assign(d->l, d->l + d->n);
    if (d->k) { for (uint64_t r = 0; r < d->m; ++r) p->total_k += d->k[r]; } else p->total_k = d->m;
    auto bail = [&](int code) { problem_free(p); return code; };
    const size_t col_bytes = (nnz + 16) * sizeof(uint32_t); // padded: the 16-byte stream may over-read
    if (hipMalloc((void **)&p->d_col, col_bytes) != hipSuccess) return bail(fail(MMG_ERR_HIP, "hipMalloc col_idx"));
    if (hipMemset(p->d_col, 0, col_bytes) != hipSuccess) return bail(fail(MMG_ERR_HIP, "hipMemset col_idx"));
    if (nnz && hipMemcpy(p->d_col, d->col_idx, nnz * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
        return bail(fail(MMG_ERR_HIP, "hipMemcpy col_idx"));
    p->device_bytes += col_bytes;
    if (d->k && d->m) {
        if (hipMalloc((void **)&p->d_k, d->m * sizeof(uint32_t)) != hipSuccess) return bail(fail(MMG_ERR_HIP, "hipMalloc k"));
        if (hipMemcpy(p->d_k, d->k, d->m * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
            return bail(fail(MMG_ERR_HIP, "hipMemcpy k"));
        p->device_bytes += d->m * 4;
    }
    if (hipMalloc((void **)&p->d_l, d->n * sizeof(double)) != hipSuccess) return bail(fail(MMG_ERR_HIP, "hipMalloc l"));
    if (hipMemcpy(p->d_l, d->l, d->n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
        return bail(fail(MMG_ERR_HIP, "hipMemcpy l"));
    p->device_bytes += d->n * 8;
    rc = problem_finish(p, d->row_ptr);
    if (rc) return bail(rc);
    rc = problem_build_desc(p);
    if (rc) return bail(rc);
    *out = p;
    return MMG_OK;
}

// Host-built transcript tables of the synthetic generator.
static void synth_tables(uint64_t seed, uint32_t T, double lambda, std::vector<double> &efflen, std::vector<double> &cdf,
                         std::vector<double> &len_cdf)
{
    efflen.resize(T);
    cdf.resize(T);
    double run = 0.0;
    for (uint32_t t = 0; t < T; ++t) {
        Stream s(seed, 0, TAG_SYNTH_TX, (uint64_t)t, 0);
        const double z1 = normal(s), z2 = normal(s);
        double ua, ub;
        s.pair(ua, ub);
        double e = dfloor(dexp(7.3132203870903014 + 0.6 * z1) + 0.5);
        if (e < 50.0) e = 50.0;
        const double th = (ua < 0.3) ? 0.0 : dexp(2.0 * z2);
        efflen[t] = e;
        run += th * e;
        cdf[t] = run;
    }
    len_cdf.resize(99);
    double p = dexp(-lambda), acc = 0.0;
    for (int j = 0; j < 99; ++j) {
        acc += p;
        len_cdf[j] = acc;
        p = p * lambda / (double)(j + 1);
    }
}

extern "C" int mmg_problem_create_synthetic(const mmg_synth_desc *d, int device, mmg_problem **out)
{
    if (!d || !out) return fail(MMG_ERR_ARG, "NULL argument");
    if (d->n == 0 || d->rows == 0 || !(d->avg_hits >= 1.0)) return fail(MMG_ERR_ARG, "bad synthetic spec");
    int rc = require_device(device);
    if (rc) return rc;
    std::vector<double> efflen, cdf, len_cdf;
    synth_tables(d->seed, d->n, d->avg_hits - 1.0, efflen, cdf, len_cdf);
    if (!(cdf[d->n - 1] > 0.0)) return fail(MMG_ERR_ARG, "synthetic abundance table is all zero");
    mmg_problem *p = new mmg_problem();
    p->device = device;
    p->m = d->rows; p->n = d->n; p->row_id_base = d->row0; p->total_k = d->rows;
    const double N = (double)(d->mapped_reads ? d->mapped_reads : d->rows);
    p->h_l.resize(d->n);
    for (uint32_t t = 0; t < d->n; ++t) p->h_l[t] = efflen[t] * N / 1000000000.0; // src/mmseq.cpp:603
    double *d_cdf = nullptr, *d_len_cdf = nullptr;
    uint32_t *d_lens = nullptr, *d_keys = nullptr, *d_perm = nullptr;
    auto cleanup = [&]() { if (d_cdf) (void)hipFree(d_cdf); if (d_len_cdf) (void)hipFree(d_len_cdf); if (d_lens) (void)hipFree(d_lens);
                           if (d_keys) (void)hipFree(d_keys); if (d_perm) (void)hipFree(d_perm); };
    if (d->sorted && d->rows >= 0xffffffffull) return fail(MMG_ERR_ARG, "sorted synthetic problems need rows < 2^32 per device");
    auto bail = [&](int code) { cleanup(); problem_free(p); return code; };
#define SYN_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return bail(fail(MMG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e))); } while (0)
    SYN_TRY(hipMalloc((void **)&d_cdf, d->n * sizeof(double)));
    SYN_TRY(hipMalloc((void **)&d_len_cdf, 99 * sizeof(double)));
    SYN_TRY(hipMalloc((void **)&d_lens, d->rows * sizeof(uint32_t)));
    SYN_TRY(hipMemcpy(d_cdf, cdf.data(), d->n * sizeof(double), hipMemcpyHostToDevice));
    SYN_TRY(hipMemcpy(d_len_cdf, len_cdf.data(), 99 * sizeof(double), hipMemcpyHostToDevice));
    SynthArgs sa{d->seed, d->row0, d->rows, d->n, d->uniform, d_cdf, d_len_cdf};
    const unsigned gb = (unsigned)((d->rows + 255) / 256);
    if (d->sorted) SYN_TRY(hipMalloc((void **)&d_keys, d->rows * sizeof(uint32_t)));
    hipLaunchKernelGGL(k_synth_len, dim3(gb), dim3(256), 0, 0, sa, d_lens, d_keys);
    SYN_TRY(hipGetLastError());
    std::vector<uint32_t> lens(d->rows);
    SYN_TRY(hipMemcpy(lens.data(), d_lens, d->rows * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::vector<uint64_t> rp(d->rows + 1);
    rp[0] = 0;
    if (d->sorted) {
        // stable two-pass (LSD) counting sort by (leading transcript, row length): deterministic.
        // Equal-length neighbours keep a wave's 64 rows in step; the leading transcript keeps the
        // LDS window of the sample kernel sliding monotonically.
        std::vector<uint32_t> keys(d->rows), perm(d->rows), tmp(d->rows);
        SYN_TRY(hipMemcpy(keys.data(), d_keys, d->rows * sizeof(uint32_t), hipMemcpyDeviceToHost));
        {
            std::vector<uint64_t> pos(102, 0);
            for (uint64_t r = 0; r < d->rows; ++r) pos[(size_t)lens[r] + 1]++;
            for (int i = 0; i < 101; ++i) pos[i + 1] += pos[i];
            for (uint64_t r = 0; r < d->rows; ++r) tmp[pos[lens[r]]++] = (uint32_t)r;
        }
        {
            std::vector<uint64_t> pos((size_t)d->n + 1, 0);
            for (uint64_t r = 0; r < d->rows; ++r) pos[(size_t)keys[r] + 1]++;
            for (uint32_t t = 0; t < d->n; ++t) pos[t + 1] += pos[t];
            for (uint64_t i = 0; i < d->rows; ++i) { const uint32_t r = tmp[i]; perm[pos[keys[r]]++] = r; }
        }
        std::vector<uint32_t>().swap(tmp);
        for (uint64_t r = 0; r < d->rows; ++r) rp[r + 1] = rp[r] + lens[perm[r]];
        SYN_TRY(hipMalloc((void **)&d_perm, d->rows * sizeof(uint32_t)));
        SYN_TRY(hipMemcpy(d_perm, perm.data(), d->rows * sizeof(uint32_t), hipMemcpyHostToDevice));
    } else {
        for (uint64_t r = 0; r < d->rows; ++r) rp[r + 1] = rp[r] + lens[r];
    }
    std::vector<uint32_t>().swap(lens);
    p->nnz = rp[d->rows];
    const size_t col_bytes = (p->nnz + 16) * sizeof(uint32_t);
    SYN_TRY(hipMalloc((void **)&p->d_col, col_bytes));
    SYN_TRY(hipMemset(p->d_col, 0, col_bytes));
    p->device_bytes += col_bytes;
    SYN_TRY(hipMalloc((void **)&p->d_l, d->n * sizeof(double)));
    SYN_TRY(hipMemcpy(p->d_l, p->h_l.data(), d->n * sizeof(double), hipMemcpyHostToDevice));
    p->device_bytes += d->n * 8;
    rc = problem_finish(p, rp.data());
    if (rc) return bail(rc);
    if (p->idx64) hipLaunchKernelGGL(k_synth_fill<uint64_t>, dim3(gb), dim3(256), 0, 0, sa, (const uint64_t *)p->d_row_ptr, (const uint32_t *)d_perm, p->d_col);
    else hipLaunchKernelGGL(k_synth_fill<uint32_t>, dim3(gb), dim3(256), 0, 0, sa, (const uint32_t *)p->d_row_ptr, (const uint32_t *)d_perm, p->d_col);
    SYN_TRY(hipGetLastError());
    SYN_TRY(hipDeviceSynchronize());
#undef SYN_TRY
    rc = problem_build_desc(p);
    if (rc) return bail(rc);
    cleanup();
    *out = p;
    return MMG_OK;
}

extern "C" int mmg_problem_info_get(const mmg_problem *p, mmg_problem_info *info)
{
    if (!p || !info) return fail(MMG_ERR_ARG, "NULL argument");
    info->m = p->m; info->nnz = p->nnz; info->total_k = p->total_k; info->row_id_base = p->row_id_base;
    info->n = p->n; info->max_row_len = p->max_row_len; info->n_tiles = p->n_tiles;
    info->device_bytes = p->device_bytes; info->index_bits = p->idx64 ? 64 : 32;
    info->sample_kernel = p->use_sell ? 2 : (p->use16 ? 1 : 0);
    info->stream_bytes = p->use_sell ? p->sell_bytes : (p->use16 ? p->stream16_bytes : 0);
    return MMG_OK;
}

extern "C" int mmg_problem_download(const mmg_problem *p, uint64_t *row_ptr, uint32_t *col_idx)
{
    if (!p) return fail(MMG_ERR_ARG, "NULL problem");
    HIP_TRY(hipSetDevice(p->device));
    if (row_ptr) {
        if (p->idx64) {
            HIP_TRY(hipMemcpy(row_ptr, p->d_row_ptr, (p->m + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
        } else {
            std::vector<uint32_t> rp32(p->m + 1);
            HIP_TRY(hipMemcpy(rp32.data(), p->d_row_ptr, (p->m + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i <= p->m; ++i) row_ptr[i] = rp32[i];
        }
    }
    if (col_idx && p->nnz) HIP_TRY(hipMemcpy(col_idx, p->d_col, p->nnz * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return MMG_OK;
}

extern "C" int mmg_problem_get_l(const mmg_problem *p, double *l)
{
    if (!p || !l) return fail(MMG_ERR_ARG, "NULL argument");
    std::memcpy(l, p->h_l.data(), p->n * sizeof(double));
    return MMG_OK;
}

extern "C" int mmg_problem_start_values(const mmg_problem *p, double *mu0, int32_t *unique_hits)
{
    if (!p) return fail(MMG_ERR_ARG, "NULL problem");
    HIP_TRY(hipSetDevice(p->device));
    double *d_acc = nullptr;
    int32_t *d_uh = nullptr;
    HIP_TRY(hipMalloc((void **)&d_acc, p->n * sizeof(double)));
    if (hipMalloc((void **)&d_uh, p->n * sizeof(int32_t)) != hipSuccess) { (void)hipFree(d_acc); return fail(MMG_ERR_HIP, "hipMalloc"); }
    int rc = MMG_OK;
    do {
        if (hipMemset(d_acc, 0, p->n * sizeof(double)) != hipSuccess || hipMemset(d_uh, 0, p->n * sizeof(int32_t)) != hipSuccess) { rc = fail(MMG_ERR_HIP, "hipMemset"); break; }
        if (p->m) {
            const unsigned gb = (unsigned)((p->m + 255) / 256);
            if (p->idx64) hipLaunchKernelGGL(k_start_values<uint64_t>, dim3(gb), dim3(256), 0, 0, (const uint64_t *)p->d_row_ptr, p->d_col, p->d_k, p->m, d_acc, d_uh);
            else hipLaunchKernelGGL(k_start_values<uint32_t>, dim3(gb), dim3(256), 0, 0, (const uint32_t *)p->d_row_ptr, p->d_col, p->d_k, p->m, d_acc, d_uh);
        }
        hipLaunchKernelGGL(k_div, dim3((p->n + 255) / 256), dim3(256), 0, 0, d_acc, p->d_l, p->n);
        if (hipGetLastError() != hipSuccess) { rc = fail(MMG_ERR_HIP, "start_values launch"); break; }
        if (mu0 && hipMemcpy(mu0, d_acc, p->n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(MMG_ERR_HIP, "hipMemcpy mu0"); break; }
        if (unique_hits && hipMemcpy(unique_hits, d_uh, p->n * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(MMG_ERR_HIP, "hipMemcpy uh"); break; }
    } while (0);
    (void)hipFree(d_acc);
    (void)hipFree(d_uh);
    return rc;
}

// CSC transpose (rows ascending within a column), built on the host from the resident CSR on first use
// ------------------------------------------------------------------------------ EM
struct mmg_em {
    mmg_problem *p = nullptr;
    double *d_mu = nullptr, *d_pc = nullptr;
    uint32_t *d_word = nullptr;
    uint64_t *d_hi = nullptr, *d_lo = nullptr, *d_ll = nullptr;
    int32_t *d_xe = nullptr, *d_sexp = nullptr;
    EmOut *d_out = nullptr;
    uint64_t *d_chunk[2] = {nullptr, nullptr}; // tile ranges of the accumulate / measure kernels
    int grid[2] = {0, 0};
    bool fast = false;
    int path = 0;   // rows-pass kernel: 2 sliced-ELL stream, 1 16-bit tile stream, 0 row per thread from the CSR
    bool first = true;
    int sweeps = 0, repeats = 0;
    double loglik = 0.0;
};

static void em_free(mmg_em *e)
{
    if (!e) return;
    (void)hipSetDevice(e->p->device);
    for (void *x : {(void *)e->d_mu, (void *)e->d_pc, (void *)e->d_word, (void *)e->d_hi, (void *)e->d_lo, (void *)e->d_ll,
                    (void *)e->d_xe, (void *)e->d_sexp, (void *)e->d_out, (void *)e->d_chunk[0], (void *)e->d_chunk[1]})
        if (x) (void)hipFree(x);
    delete e;
}

// the stream kernel is instantiated for the default K1 tile shape only; anything else takes the row-per-thread kernel
template <bool MEASURE>
static const void *em16_kernel(const mmg_problem *p)
{
    if (!p->use16 || p->variant != 0) return nullptr;
    const bool hk = p->d_k != nullptr;
    if (!MEASURE && !hk && !p->idx64) {
        if (const char *rp = getenv("MMG_EM_REP")) { // accumulator replicas (experiments)
            switch (atoi(rp)) {
            case 1: return (const void *)k_em16<uint32_t, false, 2560, 256, 128, 128, false, 0, 1>;
            case 4: return (const void *)k_em16<uint32_t, false, 2560, 256, 128, 128, false, 0, 4>;
            case 8: return (const void *)k_em16<uint32_t, false, 2560, 256, 128, 128, false, 0, 8>;
            }
        }
        if (const char *ab = getenv("MMG_EM_ABL")) { // timing ablations (wrong results by design)
            switch (atoi(ab)) {
            case 1: return (const void *)k_em16<uint32_t, false, 2560, 256, 128, 128, false, 1>;
            case 2: return (const void *)k_em16<uint32_t, false, 2560, 256, 128, 128, false, 2>;
            case 3: return (const void *)k_em16<uint32_t, false, 2560, 256, 128, 128, false, 3>;
            }
        }
    }
    if (p->idx64) return hk ? (const void *)k_em16<uint64_t, true, 2560, 256, 128, 128, MEASURE> : (const void *)k_em16<uint64_t, false, 2560, 256, 128, 128, MEASURE>;
    return hk ? (const void *)k_em16<uint32_t, true, 2560, 256, 128, 128, MEASURE> : (const void *)k_em16<uint32_t, false, 2560, 256, 128, 128, MEASURE>;
}

// k_em_sell runs as 2 waves per workgroup with 4 accumulator replicas (1.93 ms per sweep at cfg 3; 1 wave x 2 replicas: 2.12,
// 2 x 2: 2.01, 4 x 4: 1.99, 2 x 1: 2.34); MMG_EM_WAVES=1 selects the single-wave form for comparison.
static int em_sell_waves()
{
    const char *ev = getenv("MMG_EM_WAVES");
    return ev && atoi(ev) == 1 ? 1 : 2;
}

template <bool MEASURE>
static const void *em_sell_kernel(const mmg_problem *p)
{
    if (!p->use_sell) return nullptr;
    const bool hk = p->d_k != nullptr;
    const bool one = em_sell_waves() == 1;
#define EMS_PICK(IDX, HK) (MEASURE ? (one ? (const void *)k_em_sell<IDX, HK, true, 1, 1> : (const void *)k_em_sell<IDX, HK, true, 1, 2>) \
                                   : (one ? (const void *)k_em_sell<IDX, HK, MEASURE, 2, 1> : (const void *)k_em_sell<IDX, HK, MEASURE, 4, 2>))
    if (p->idx64) return hk ? EMS_PICK(uint64_t, true) : EMS_PICK(uint64_t, false);
    return hk ? EMS_PICK(uint32_t, true) : EMS_PICK(uint32_t, false);
#undef EMS_PICK
}

static int em_launch_rows(mmg_em *e, bool measure)
{
    mmg_problem *p = e->p;
    EmArgs a;
    a.n = p->n; a.mu = e->d_mu; a.word = e->d_word; a.hi = e->d_hi; a.lo = e->d_lo; a.xe = e->d_xe; a.ll = e->d_ll;
    if (p->m == 0) return MMG_OK;
    if (e->path == 2) {
        const int w = measure ? 1 : 0;
        const void *fn = measure ? em_sell_kernel<true>(p) : em_sell_kernel<false>(p);
        const void *rp = p->d_row_ptr;
        const uint32_t *col = p->d_col, *kk = p->d_k;
        const SellTile *tiles = p->d_sell_tiles;
        const uint64_t *chunk = e->d_chunk[w];
        const uint8_t *stream = p->d_sell;
        void *args[] = {(void *)&rp, (void *)&col, (void *)&kk, (void *)&tiles, (void *)&chunk, (void *)&stream, (void *)&a};
        HIP_TRY(hipLaunchKernel(fn, dim3((unsigned)e->grid[w]), dim3(64 * em_sell_waves()), args, 0, 0));
        return MMG_OK;
    }
    if (e->fast) {
        const int w = measure ? 1 : 0;
        const void *fn = measure ? em16_kernel<true>(p) : em16_kernel<false>(p);
        const void *rp = p->d_row_ptr;
        const uint32_t *col = p->d_col, *kk = p->d_k;
        const S16Tile *tiles = p->d_s16tiles;
        const uint64_t *chunk = e->d_chunk[w];
        const u32x4 *stream = (const u32x4 *)p->d_stream16;
        void *args[] = {(void *)&rp, (void *)&col, (void *)&kk, (void *)&tiles, (void *)&chunk, (void *)&stream, (void *)&a};
        HIP_TRY(hipLaunchKernel(fn, dim3((unsigned)e->grid[w]), dim3(128), args, 0, 0));
        return MMG_OK;
    }
    const unsigned gr = (unsigned)((p->m + 255) / 256);
    if (p->idx64) {
        if (measure) hipLaunchKernelGGL((k_em_rows_global<uint64_t, true>), dim3(gr), dim3(256), 0, 0, (const uint64_t *)p->d_row_ptr, p->d_col, p->d_k, p->m, a);
        else hipLaunchKernelGGL((k_em_rows_global<uint64_t, false>), dim3(gr), dim3(256), 0, 0, (const uint64_t *)p->d_row_ptr, p->d_col, p->d_k, p->m, a);
    } else {
        if (measure) hipLaunchKernelGGL((k_em_rows_global<uint32_t, true>), dim3(gr), dim3(256), 0, 0, (const uint32_t *)p->d_row_ptr, p->d_col, p->d_k, p->m, a);
        else hipLaunchKernelGGL((k_em_rows_global<uint32_t, false>), dim3(gr), dim3(256), 0, 0, (const uint32_t *)p->d_row_ptr, p->d_col, p->d_k, p->m, a);
    }
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
            hipLaunchKernelGGL(k_fill_i32, dim3(gn), dim3(256), 0, 0, e->d_xe, p->n, INT32_MIN);
            int rc = em_launch_rows(e, true);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(k_em_prepare, dim3(gn), dim3(256), 0, 0, p->n, e->d_mu, p->d_l, p->d_colcnt,
                           measured ? e->d_xe : e->d_sexp, measured, e->d_word, e->d_hi, e->d_lo, e->d_pc, e->d_ll);
        int rc = em_launch_rows(e, false);
        if (rc) return rc;
        if (!measured) hipLaunchKernelGGL(k_em_check, dim3(gn), dim3(256), 0, 0, p->n, e->d_word, e->d_hi, e->d_ll);
        hipLaunchKernelGGL(k_em_finish, dim3(1), dim3(1), 0, 0, e->d_pc, gn, e->d_ll, e->d_out);
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
            hipLaunchKernelGGL(k_em_colcount, dim3(g), dim3(256), 0, 0, p->d_col, p->nnz, p->d_colcnt);
            HIP_TRY(hipGetLastError());
        }
        p->device_bytes += p->n * 8;
    }
    mmg_em *e = new mmg_em();
    e->p = p;
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
    EM_TRY(hipMemcpy(e->d_mu, mu0, p->n * sizeof(double), hipMemcpyHostToDevice));
    e->fast = em16_kernel<false>(p) != nullptr && p->n_tiles > 0;
    e->path = (em_sell_kernel<false>(p) != nullptr && p->n_sell_tiles > 0) ? 2 : (e->fast ? 1 : 0);
    if (const char *ev = getenv("MMG_EM_STREAM")) { // 0: row-per-thread kernel, 1: 16-bit tile stream (if the problem has one)
        if (atoi(ev) == 0) e->path = 0;
        if (atoi(ev) == 1) e->path = e->fast ? 1 : 0;
    }
    e->fast = e->path == 1;
    if (e->path) {
        const uint64_t n_tiles = e->path == 2 ? p->n_sell_tiles : p->n_tiles;
        const unsigned bs = e->path == 2 ? 64 * em_sell_waves() : 128;
        for (int w = 0; w < 2; ++w) {
            const void *fn = e->path == 2 ? (w ? em_sell_kernel<true>(p) : em_sell_kernel<false>(p)) : (w ? em16_kernel<true>(p) : em16_kernel<false>(p));
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, bs, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 4; }
            if (per_cu > 32) per_cu = 32;
            uint64_t grid = std::max<uint64_t>(1, std::min<uint64_t>(n_tiles, (uint64_t)p->cu_count * per_cu));
            if (const char *eg = getenv("MMG_EM_GRID")) { const long v = atol(eg); if (v >= 1 && (uint64_t)v < grid) grid = (uint64_t)v; } // tests: long tile ranges on small problems
            std::vector<uint64_t> chunk(grid + 1);
            if (e->path == 2 && p->h_sell_cum.size() == n_tiles + 1) weighted_chunks(p->h_sell_cum, grid, chunk);
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
    hipLaunchKernelGGL(k_em_apply, dim3((p->n + 255) / 256), dim3(256), 0, 0, p->n, e->d_mu, p->d_l, e->d_word, e->d_hi, e->d_lo, e->d_sexp);
    int rc = em_rows_pass(e);
    if (rc) return rc;
    ++e->sweeps;
    if (loglik) *loglik = e->loglik;
    return MMG_OK;
}

extern "C" int mmg_em_get_mu(mmg_em *e, double *mu)
{
    if (!e || !mu) return fail(MMG_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(e->p->device));
    HIP_TRY(hipMemcpy(mu, e->d_mu, e->p->n * sizeof(double), hipMemcpyDeviceToHost));
    return MMG_OK;
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

extern "C" void mmg_problem_destroy(mmg_problem *p) { problem_free(p); }

// ------------------------------------------------------------------------------ sampler
struct mmg_sampler {
    const mmg_problem *p = nullptr;
    mmg_config cfg{};
    hipStream_t own = nullptr, cur = nullptr;
    double *d_mu = nullptr, *d_scale = nullptr, *d_trace = nullptr, *d_mom = nullptr; // mom: [2][C][n]
    int32_t *d_cnt = nullptr, *d_cnt_last = nullptr;
    int iter = 0;          // completed iterations
    bool sampled = false;  // sample() issued for the current iteration, update() pending
    int64_t n_kept = 0;
    // timing
    std::vector<hipEvent_t> ev_pool;
    std::vector<std::pair<int, int>> ev_sample, ev_update; // indices into ev_pool
    size_t ev_used = 0;
    double acc_sample_ms = 0, acc_update_ms = 0;
    uint64_t acc_sample_n = 0, acc_update_n = 0;
};

static void sampler_free(mmg_sampler *s)
{
    if (!s) return;
    (void)hipSetDevice(s->p->device);
    if (s->own) { (void)hipStreamSynchronize(s->own); }
    for (auto e : s->ev_pool) (void)hipEventDestroy(e);
    if (s->d_mu) (void)hipFree(s->d_mu);
    if (s->d_scale) (void)hipFree(s->d_scale);
    if (s->d_trace) (void)hipFree(s->d_trace);
    if (s->d_mom) (void)hipFree(s->d_mom);
    if (s->d_cnt) (void)hipFree(s->d_cnt);
    if (s->d_cnt_last) (void)hipFree(s->d_cnt_last);
    if (s->own) (void)hipStreamDestroy(s->own);
    delete s;
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
    s->cfg = *cfg;
    const size_t C = (size_t)cfg->n_chains, n = p->n;
    auto bail = [&](int code) { sampler_free(s); return code; };
#define S_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return bail(fail(MMG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e))); } while (0)
    S_TRY(hipStreamCreateWithFlags(&s->own, hipStreamNonBlocking));
    s->cur = s->own;
    S_TRY(hipMalloc((void **)&s->d_mu, C * n * sizeof(double)));
    S_TRY(hipMalloc((void **)&s->d_scale, n * sizeof(double)));
    S_TRY(hipMalloc((void **)&s->d_mom, 2 * C * n * sizeof(double)));
    S_TRY(hipMalloc((void **)&s->d_cnt, C * n * sizeof(int32_t)));
    S_TRY(hipMalloc((void **)&s->d_cnt_last, C * n * sizeof(int32_t)));
    if (cfg->keep_trace) S_TRY(hipMalloc((void **)&s->d_trace, C * n * (size_t)cfg->trace_len * sizeof(double)));
    std::vector<double> scale(n);
    for (size_t t = 0; t < n; ++t) scale[t] = 1.0 / (cfg->beta + p->h_l[t]); // src/mmseq.cpp:907 second argument
    S_TRY(hipMemcpy(s->d_scale, scale.data(), n * sizeof(double), hipMemcpyHostToDevice));
    for (size_t c = 0; c < C; ++c) S_TRY(hipMemcpy(s->d_mu + c * n, mu0, n * sizeof(double), hipMemcpyHostToDevice));
    S_TRY(hipMemset(s->d_mom, 0, 2 * C * n * sizeof(double)));
    S_TRY(hipMemset(s->d_cnt, 0, C * n * sizeof(int32_t)));
    S_TRY(hipMemset(s->d_cnt_last, 0, C * n * sizeof(int32_t)));
    if (s->d_trace) S_TRY(hipMemset(s->d_trace, 0, C * n * (size_t)cfg->trace_len * sizeof(double)));
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

static int ev_get(mmg_sampler *s, int &idx)
{
    if (s->ev_used == s->ev_pool.size()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        s->ev_pool.push_back(e);
    }
    idx = (int)s->ev_used++;
    return MMG_OK;
}

extern "C" int mmg_sampler_sample(mmg_sampler *s)
{
    if (!s) return fail(MMG_ERR_ARG, "NULL sampler");
    if (s->sampled) return fail(MMG_ERR_STATE, "sample() already issued for this iteration; call update()");
    const mmg_problem *p = s->p;
    HIP_TRY(hipSetDevice(p->device));
    int e0 = -1, e1 = -1;
    const bool timed = s->cfg.timing > 0 && s->iter % s->cfg.timing == 0; // every timing-th iteration: an event pair costs ~9 us of stream time
    if (timed) {
        int rc = ev_get(s, e0); if (rc) return rc;
        rc = ev_get(s, e1); if (rc) return rc;
        HIP_TRY(hipEventRecord(s->ev_pool[e0], s->cur));
    }
    if (p->n_tiles > 0) {
        int fuse_cap = 8;
        if (const char *ev = getenv("MMG_K1_FUSE")) { fuse_cap = atoi(ev); if (fuse_cap < 1) fuse_cap = 1; }
        for (int c = 0; c < s->cfg.n_chains;) {
            // chains are advanced in fused groups of 8 / 4 / 2 / 1 (the walk reads each hit's offset once per group)
            int fuse = 1;
            if (p->use16 && !p->use_sell) {
                for (int f = 8; f > 1; f >>= 1)
                    if (f <= fuse_cap && c + f <= s->cfg.n_chains && k1_s16_kernel(p->variant, p->idx64, p->d_k != nullptr, f)) { fuse = f; break; }
            }
            SampleArgs a;
            a.seed = s->cfg.seed; a.row_id_base = p->row_id_base; a.n = p->n;
            a.chain = (uint32_t)(s->cfg.chain_base + c);
            a.iter = (uint32_t)s->iter;
            const void *rp = p->d_row_ptr;
            const uint32_t *ci = p->d_col, *kk = p->d_k;
            const TileDesc *td = p->d_tiles;
            const uint64_t *ct = p->d_chunk_tile;
            const double *mu = s->d_mu + (size_t)c * p->n;
            int32_t *cnt = s->d_cnt + (size_t)c * p->n;
            if (p->use_sell) {
                const SellTile *ts = p->d_sell_tiles;
                const uint64_t *cs = p->d_sell_chunk;
                const uint8_t *ss = p->d_sell;
                void *kargs[] = {(void *)&rp, (void *)&ci, (void *)&kk, (void *)&ts, (void *)&cs, (void *)&mu, (void *)&ss, (void *)&cnt, (void *)&a};
                HIP_TRY(hipLaunchKernel(k1_sell_kernel(p->idx64, p->d_k != nullptr), dim3(p->grid_sell), dim3(64), kargs, 0, s->cur));
            } else if (p->use16) {
                const S16Tile *t16 = p->d_s16tiles;
                const uint64_t *c16 = p->d_chunk_tile16;
                const void *s16 = p->d_stream16;
                void *kargs[] = {(void *)&rp, (void *)&ci, (void *)&kk, (void *)&t16, (void *)&c16, (void *)&mu, (void *)&s16, (void *)&cnt, (void *)&a};
                // fused instances need more LDS per workgroup: fewer fit per CU, the tile ranges stay those of grid16
                HIP_TRY(hipLaunchKernel(k1_s16_kernel(p->variant, p->idx64, p->d_k != nullptr, fuse), dim3(p->grid16), dim3(k1_variants[p->variant].bs), kargs, 0, s->cur));
            } else {
                void *kargs[] = {(void *)&rp, (void *)&ci, (void *)&kk, (void *)&td, (void *)&ct, (void *)&mu, (void *)&cnt, (void *)&a};
                HIP_TRY(hipLaunchKernel(k1_kernel(p->variant, p->idx64, p->d_k != nullptr), dim3(p->grid_sample), dim3(k1_variants[p->variant].bs), kargs, 0, s->cur));
            }
            c += fuse;
        }
        HIP_TRY(hipGetLastError());
    }
    if (timed) {
        HIP_TRY(hipEventRecord(s->ev_pool[e1], s->cur));
        s->ev_sample.push_back({e0, e1});
    }
    s->sampled = true;
    return MMG_OK;
}

extern "C" int mmg_sampler_update(mmg_sampler *s)
{
    if (!s) return fail(MMG_ERR_ARG, "NULL sampler");
    if (!s->sampled) return fail(MMG_ERR_STATE, "update() without a preceding sample()");
    const mmg_problem *p = s->p;
    HIP_TRY(hipSetDevice(p->device));
    const int ss = s->cfg.gibbs_iter / s->cfg.trace_len; // src/mmseq.cpp:284
    int sample_idx = -1;
    if (s->iter % ss == 0 && s->iter / ss < s->cfg.trace_len) sample_idx = s->iter / ss; // :911, :914
    const size_t C = (size_t)s->cfg.n_chains, n = p->n;
    UpdateArgs a;
    a.cnt = s->d_cnt; a.cnt_last = s->d_cnt_last; a.scale = s->d_scale; a.mu = s->d_mu; a.trace = s->d_trace;
    a.sum_log = s->d_mom; a.sum_log2 = s->d_mom + C * n;
    a.seed = s->cfg.seed; a.alpha = s->cfg.alpha; a.n = p->n; a.n_chains = (uint32_t)C;
    a.chain_base = (uint32_t)s->cfg.chain_base; a.iter = (uint32_t)s->iter; a.sample_idx = sample_idx;
    a.trace_len = (uint32_t)s->cfg.trace_len;
    int e0 = -1, e1 = -1;
    const bool timed = s->cfg.timing > 0 && s->iter % s->cfg.timing == 0;
    if (timed) {
        int rc = ev_get(s, e0); if (rc) return rc;
        rc = ev_get(s, e1); if (rc) return rc;
        HIP_TRY(hipEventRecord(s->ev_pool[e0], s->cur));
    }
    const unsigned gb = (unsigned)((C * n + 255) / 256);
    hipLaunchKernelGGL(k_update, dim3(gb), dim3(256), 0, s->cur, a);
    HIP_TRY(hipGetLastError());
    if (timed) {
        HIP_TRY(hipEventRecord(s->ev_pool[e1], s->cur));
        s->ev_update.push_back({e0, e1});
    }
    if (sample_idx >= 0) s->n_kept++;
    s->iter++;
    s->sampled = false;
    return MMG_OK;
}

extern "C" int mmg_sampler_run(mmg_sampler *s, int n_iter)
{
    if (!s || n_iter < 0) return fail(MMG_ERR_ARG, "bad argument");
    for (int i = 0; i < n_iter; ++i) {
        int rc = mmg_sampler_sample(s);
        if (rc) return rc;
        rc = mmg_sampler_update(s);
        if (rc) return rc;
    }
    return MMG_OK;
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
    HIP_TRY(hipSetDevice(s->p->device));
    HIP_TRY(hipStreamSynchronize(s->cur));
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
    HIP_TRY(hipSetDevice(s->p->device));
    const size_t n = s->p->n, S = (size_t)s->cfg.trace_len;
    double *d_tmp = nullptr;
    HIP_TRY(hipMalloc((void **)&d_tmp, n * S * sizeof(double)));
    const dim3 g((unsigned)((n + 31) / 32), (unsigned)((S + 31) / 32));
    hipLaunchKernelGGL(k_transpose, g, dim3(256), 0, s->cur, s->d_trace + (size_t)chain * S * n, d_tmp, (uint32_t)n, (uint32_t)S);
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
    if (!out || first < 0 || count < 0 || first + count > s->cfg.trace_len) return fail(MMG_ERR_ARG, "bad sample range");
    if (!s->d_trace) return fail(MMG_ERR_STATE, "sampler was created with keep_trace == 0");
    HIP_TRY(hipSetDevice(s->p->device));
    HIP_TRY(hipStreamSynchronize(s->cur));
    const size_t n = s->p->n, S = (size_t)s->cfg.trace_len;
    HIP_TRY(hipMemcpy(out, s->d_trace + ((size_t)chain * S + (size_t)first) * n, (size_t)count * n * sizeof(double), hipMemcpyDeviceToHost));
    return MMG_OK;
}

extern "C" int mmg_sampler_get_mu(mmg_sampler *s, int chain, double *mu)
{
    int rc = check_chain(s, chain);
    if (rc) return rc;
    if (!mu) return fail(MMG_ERR_ARG, "NULL out");
    HIP_TRY(hipSetDevice(s->p->device));
    HIP_TRY(hipStreamSynchronize(s->cur));
    HIP_TRY(hipMemcpy(mu, s->d_mu + (size_t)chain * s->p->n, s->p->n * sizeof(double), hipMemcpyDeviceToHost));
    return MMG_OK;
}

extern "C" int mmg_sampler_get_counts(mmg_sampler *s, int chain, int32_t *cnt)
{
    int rc = check_chain(s, chain);
    if (rc) return rc;
    if (!cnt) return fail(MMG_ERR_ARG, "NULL out");
    HIP_TRY(hipSetDevice(s->p->device));
    HIP_TRY(hipStreamSynchronize(s->cur));
    // between sample() and update() the live counts are the interesting ones
    const int32_t *src = (s->sampled ? s->d_cnt : s->d_cnt_last) + (size_t)chain * s->p->n;
    HIP_TRY(hipMemcpy(cnt, src, s->p->n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return MMG_OK;
}

extern "C" int mmg_sampler_get_moments(mmg_sampler *s, int chain, double *sum_log, double *sum_log2, int64_t *n_samples)
{
    int rc = check_chain(s, chain);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(s->p->device));
    HIP_TRY(hipStreamSynchronize(s->cur));
    const size_t C = (size_t)s->cfg.n_chains, n = s->p->n;
    if (sum_log) HIP_TRY(hipMemcpy(sum_log, s->d_mom + (size_t)chain * n, n * sizeof(double), hipMemcpyDeviceToHost));
    if (sum_log2) HIP_TRY(hipMemcpy(sum_log2, s->d_mom + (C + (size_t)chain) * n, n * sizeof(double), hipMemcpyDeviceToHost));
    if (n_samples) *n_samples = s->n_kept;
    return MMG_OK;
}

static int drain_events(mmg_sampler *s)
{
    HIP_TRY(hipSetDevice(s->p->device));
    HIP_TRY(hipStreamSynchronize(s->cur));
    for (auto &pr : s->ev_sample) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, s->ev_pool[pr.first], s->ev_pool[pr.second]));
        s->acc_sample_ms += ms; s->acc_sample_n++;
    }
    for (auto &pr : s->ev_update) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, s->ev_pool[pr.first], s->ev_pool[pr.second]));
        s->acc_update_ms += ms; s->acc_update_n++;
    }
    s->ev_sample.clear(); s->ev_update.clear(); s->ev_used = 0;
    return MMG_OK;
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

extern "C" int mmg_host_gamma_trace(uint64_t seed, uint64_t id, double shape, double scale, int n, double *out)
{
    if (n < 0 || !out || !(shape > 0.0)) return fail(MMG_ERR_ARG, "bad argument");
    for (int i = 0; i < n; ++i) {
        Stream s(seed, 0, TAG_SIMU, id, (uint32_t)i);
        out[i] = gamma_unit(s, shape) * scale;
    }
    return MMG_OK;
}

// ------------------------------------------------------------------------------ self tests
extern "C" int mmg_selftest_math(int device, int64_t n, const double *x, double *ol, double *oe, double *os, double *orc)
{
    if (n < 0 || !x || !ol || !oe || !os || !orc) return fail(MMG_ERR_ARG, "bad argument");
    if (device < 0) {
        for (int64_t i = 0; i < n; ++i) { ol[i] = dlog(x[i]); oe[i] = dexp(x[i]); os[i] = dsqrt(x[i]); orc[i] = 1.0 / x[i]; }
        return MMG_OK;
    }
    int rc = require_device(device);
    if (rc) return rc;
    double *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 5 * (size_t)n * sizeof(double) + 8));
    hipError_t e = hipMemcpy(d, x, n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && n) {
        hipLaunchKernelGGL(k_selftest_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, d, d + n, d + 2 * n, d + 3 * n, d + 4 * n);
        e = hipDeviceSynchronize();
    }
    if (e == hipSuccess) e = hipMemcpy(ol, d + n, n * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(oe, d + 2 * n, n * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(os, d + 3 * n, n * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(orc, d + 4 * n, n * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(MMG_ERR_HIP, std::string("selftest_math: ") + hipGetErrorString(e));
    return MMG_OK;
}

extern "C" int mmg_selftest_philox(int device, const uint32_t *ctr, const uint32_t *key, uint32_t *out)
{
    if (!ctr || !key || !out) return fail(MMG_ERR_ARG, "NULL argument");
    if (device < 0) {
        const U4 r = philox4x32_10(U4{ctr[0], ctr[1], ctr[2], ctr[3]}, key[0], key[1]);
        out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
        uint32_t a = ctr[0], b = ctr[1];
        philox2x32_10(a, b, key[0]);
        out[4] = a; out[5] = b;
        return MMG_OK;
    }
    int rc = require_device(device);
    if (rc) return rc;
    uint32_t *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 12 * sizeof(uint32_t)));
    hipError_t e = hipMemcpy(d, ctr, 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + 4, key, 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) { hipLaunchKernelGGL(k_selftest_philox, dim3(1), dim3(1), 0, 0, d, d + 4, d + 6); e = hipDeviceSynchronize(); }
    if (e == hipSuccess) e = hipMemcpy(out, d + 6, 24, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(MMG_ERR_HIP, std::string("selftest_philox: ") + hipGetErrorString(e));
    return MMG_OK;
}

extern "C" int mmg_selftest_gamma(int device, uint64_t seed, double shape, double scale, int64_t n, double *out)
{
    if (n < 0 || !out || !(shape > 0.0)) return fail(MMG_ERR_ARG, "bad argument");
    if (device < 0) {
        for (int64_t i = 0; i < n; ++i) { Stream s(seed, 0, TAG_GAMMA, (uint64_t)i, 0); out[i] = gamma_unit(s, shape) * scale; }
        return MMG_OK;
    }
    int rc = require_device(device);
    if (rc) return rc;
    double *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, (size_t)n * sizeof(double) + 8));
    hipError_t e = hipSuccess;
    if (n) { hipLaunchKernelGGL(k_selftest_gamma, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, seed, shape, scale, n, d); e = hipDeviceSynchronize(); }
    if (e == hipSuccess) e = hipMemcpy(out, d, n * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(MMG_ERR_HIP, std::string("selftest_gamma: ") + hipGetErrorString(e));
    return MMG_OK;
}

extern "C" int mmg_selftest_binomial(int device, uint64_t seed, uint32_t nn, double p, int64_t n, uint32_t *out)
{
    if (n < 0 || !out) return fail(MMG_ERR_ARG, "bad argument");
    if (device < 0) {
        for (int64_t i = 0; i < n; ++i) { Stream2 q(seed, 0, TAG_ROW, (uint64_t)i, 0); out[i] = binomial(q, nn, p); }
        return MMG_OK;
    }
    int rc = require_device(device);
    if (rc) return rc;
    uint32_t *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, (size_t)n * sizeof(uint32_t) + 8));
    hipError_t e = hipSuccess;
    if (n) { hipLaunchKernelGGL(k_selftest_binomial, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, seed, nn, p, n, d); e = hipDeviceSynchronize(); }
    if (e == hipSuccess) e = hipMemcpy(out, d, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(MMG_ERR_HIP, std::string("selftest_binomial: ") + hipGetErrorString(e));
    return MMG_OK;
}
