// mmgibbs.hip -- implementation of the C ABI in include/mmgibbs.h on HIP / gfx950: the host side of the device boundary
// that replaces src/mmseq.cpp:833-925 of the reference.  Kernels live in k1.hip / em.hip / misc.hip / layout.hip
// and are reached through mmg_launch.h; this file owns handles, memory, layout decisions and launch order.
#include "../../include/mmgibbs.h"
#include "mmg_launch.h"
#include "mmg_host.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace mmg;

// ------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
int mmg::fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

extern "C" const char *mmg_last_error(void) { return g_err.c_str(); }
extern "C" int mmg_abi_version(void) { return MMG_ABI_VERSION; }

// self-test overrides (mmg_selftest_option): -1 = the library decides
static std::atomic<int> g_opt[MMG_OPT_COUNT_] = {{-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}};
int mmg::opt(int o) { return g_opt[o].load(std::memory_order_relaxed); }
extern "C" int mmg_selftest_option(int option, int value)
{
    if (option < 0 || option >= MMG_OPT_COUNT_) return fail(MMG_ERR_ARG, "unknown self-test option");
    g_opt[option].store(value < 0 ? -1 : value, std::memory_order_relaxed);
    return MMG_OK;
}

extern "C" int mmg_device_count(int *count)
{
    if (!count) return fail(MMG_ERR_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return MMG_OK;
}

int mmg::require_device(int device)
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

// simple host parallel-for over [0, n) in contiguous slices
template <typename F>
static void parallel_slices(uint64_t n, F f)
{
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 32) nt = 32;
    if (n < 1u << 16) nt = 1;
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back([=] { f(n * t / nt, n * (t + 1) / nt); });
    f(0, n / nt);
    for (auto &x : th) x.join();
}

// ------------------------------------------------------------------------------ problem
static void problem_free(mmg_problem *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (void *x : {(void *)p->d_row_ptr, (void *)p->d_col, (void *)p->d_k, (void *)p->d_l, (void *)p->d_int_of_ext, (void *)p->d_ext_of_int,
                    (void *)p->d_sell, (void *)p->d_sell_tiles, (void *)p->d_sell_chunk, (void *)p->d_sell_chunk_k, (void *)p->d_sell_tiles_1, (void *)p->d_sell_tiles_k, (void *)p->d_sell_chunk_m[0], (void *)p->d_sell_chunk_m[1], (void *)(p->owns_tiles_f ? p->d_sell_tiles_f : nullptr), (void *)p->d_sell_tiles_x, (void *)p->d_sell_chunk_x, (void *)p->d_bigk_list,
                    (void *)p->d_tiles, (void *)p->d_chunk_tile,
                    (void *)p->d_colcnt})
        if (x) (void)hipFree(x);
    delete p;
}

// Contiguous tile ranges of (nearly) equal COST: cum[t] = cost of tiles [0, t).  A tile that cannot run on the register path
// is walked from the CSR and costs many times more; with equal tile counts a tail of such tiles (far rows are sorted
// last) lands on a few workgroups that finish long after the rest.
void mmg::weighted_chunks(const std::vector<uint64_t> &cum, uint64_t grid, std::vector<uint64_t> &chunk)
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
// tile costs for the ranges of the persistent workgroups, in halves of a register-path tile (measured, tools/k1_ab.py --far)
constexpr uint64_t SELL_TILES_PER_RANGE = 36; // target length of a workgroup's tile range once a launch has several generations
constexpr uint64_t SELL_FAST_TILE_COST = 2;
constexpr uint64_t SELL_FAR_TILE_COST = 2;   // plus SELL_FAR_ENTRY_COST per entry of the far list
constexpr uint64_t SELL_FAR_ENTRY_COST = 4;
constexpr uint64_t SELL_SLOW_TILE_COST = 48; // a CSR-walked tile
// the cut into read shards (h_shard_cum): a register-path tile of 5.4 groups (config 3) costs 1 + 2 * 5.4 = 11.8 of these units
constexpr uint64_t SHARD_TILE_COST = 1, SHARD_GROUP_COST = 2;
constexpr uint64_t SHARD_FAR_ENTRY_COST = 24;  // measured (tools/shard_balance.py): a far tile with one entry per lane takes 3 x a register-path tile
constexpr uint64_t SHARD_SLOW_TILE_COST = 280; // 24 x a register-path tile
constexpr uint64_t SHARD_DRAW_COST = 12; // per draw of a tile's largest k (as cumk: 2 halves of a tile)
// a row on the conditional-binomial chain is not walked with its tile but from the list (k_sample_bigk): 1.12 ms for 2 M rows of 20 hits against
// 0.29 ns of device time per register-path tile (11.8 units) -- 5/4 of a unit per hit of such a row
constexpr uint64_t SHARD_CHAIN_HIT_COST_NUM = 5, SHARD_CHAIN_HIT_COST_DEN = 4;
// k_sample (CSR tiles) costs 2.8 of these units per 64 hits (7.4 ms for 1.0 G uniform hits): a problem dearer on the stream kernel runs there

// The same with a tapered end: the last `resident` ranges' worth of cost is cut into twice as many ranges of half the cost (the
// workgroups that start last finish closer together).  grid -> grid + resident ranges; needs grid >= 2 * resident.
void mmg::weighted_chunks_tapered(const std::vector<uint64_t> &cum, uint64_t grid, uint64_t resident, std::vector<uint64_t> &chunk)
{
    if (resident == 0 || grid < 2 * resident) { mmg::weighted_chunks(cum, grid, chunk); return; }
    const uint64_t nt = cum.size() - 1, total = cum[nt], full = grid - resident, n = full + 2 * resident;
    chunk.assign(n + 1, 0);
    uint64_t t = 0;
    for (uint64_t c = 1; c < n; ++c) {
        const uint64_t halves = c <= full ? 2 * c : 2 * full + (c - full); // cost of ranges [0, c) in half-range units
        const uint64_t target = (uint64_t)(((unsigned __int128)total * halves) / (2 * grid));
        while (t < nt && cum[t] < target) ++t;
        chunk[c] = t;
    }
    chunk[n] = nt;
}

// The ranges of a launch as the kernels read them: one 64-byte header per workgroup -- first tile, end tile and the descriptors of the
// range's first two tiles -- so that ONE scalar load gives a workgroup everything it needs to request its window and its first blocks
// (with a plain table of boundaries the descriptors are a second dependent memory round trip: 1-2 us per workgroup, which is what a
// config-2 launch of 29 us is largely made of).
static hipError_t upload_ranges(const std::vector<uint64_t> &chunk, const std::vector<SellTile> &tiles, uint64_t **d_out)
{
    static_assert(sizeof(SellTile) == 24, "three 64-bit words per descriptor");
    const size_t n = chunk.size() - 1;
    std::vector<uint64_t> h(n * 8 + 8, 0);
    SellTile none;
    none.off16 = 0; none.r0 = 0; none.wbase = 0; none.meta = sell_meta(0, 0, SELL_EMPTY);
    for (size_t c = 0; c < n; ++c) {
        h[c * 8] = chunk[c];
        h[c * 8 + 1] = chunk[c + 1];
        for (int j = 0; j < 2; ++j) {
            const SellTile &d = chunk[c] + (uint64_t)j < chunk[c + 1] ? tiles[chunk[c] + j] : none;
            std::memcpy(&h[c * 8 + 2 + 3 * j], &d, sizeof(SellTile));
        }
    }
    hipError_t e = hipMalloc((void **)d_out, h.size() * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMemcpy(*d_out, h.data(), h.size() * sizeof(uint64_t), hipMemcpyHostToDevice);
    return e;
}

// List entries per workgroup of k_sample_bigk (one wave): 64 -- a row per lane, no second helping.  Measured (profiles/r06_bigk_ab.md):
// longer pieces, whose lanes fetch the next row when they finish one, lose to it at every list length (2 M rows: 1.17 ms at 64, 1.18 /
// 1.21 / 1.36 at 128 / 256 / 512; 262 k rows: 0.33 ms at 64, 0.36 / 0.40 / 0.47 at 96 / 128 / 192) -- the canonical order already puts rows
// of equal length side by side, and many short waves balance the SIMDs better than few long ones; thinner waves (16 rows) run every phase
// for a handful of lanes.  MMG_OPT_BIGK_PER_WAVE overrides (tests run 1 ... 300 entries per wave).
static uint32_t bigk_piece(uint64_t n_list, int cu_count)
{
    (void)n_list; (void)cu_count;
    if (opt(MMG_OPT_BIGK_PER_WAVE) >= 1) return (uint32_t)opt(MMG_OPT_BIGK_PER_WAVE);
    return 64;
}

// Sliced-ELL stream: tiles of <= 64 rows that never cross a (near, band) boundary of the canonical order, one window per tile.
static int problem_build_sell(mmg_problem *p, const std::vector<uint64_t> &seg_starts, const uint64_t *d_key)
{
    if (p->m == 0 || opt(MMG_OPT_SAMPLE_KERNEL) == 0) return MMG_OK;
    // Tiles of <= 64 rows inside the runs of equal (near, band).  The rows of a tile must lie within 32 consecutive Philox
    // blocks (a block serves row ids 2q and 2q+1, mmg_math.h: Stream2), so a tile that starts at an odd row id holds 63 rows.
    std::vector<uint64_t> tile_row;
    std::vector<uint64_t> whole(1, 0);
    const std::vector<uint64_t> &segs = seg_starts.empty() ? whole : seg_starts;
    tile_row.reserve(p->m / 64 + segs.size() + 2);
    for (size_t sgi = 0; sgi < segs.size(); ++sgi) {
        const uint64_t s = segs[sgi], e = sgi + 1 < segs.size() ? segs[sgi + 1] : p->m;
        for (uint64_t r = s; r < e; r += 64 - ((p->row_id_base + r) & 1)) tile_row.push_back(r);
    }
    tile_row.push_back(p->m);
    const uint64_t nt = tile_row.size() - 1;
    if (nt >= 0x7fffffffull) return MMG_OK; // a 1-D grid cannot describe them; the CSR kernel takes over
    uint64_t *d_tile_row = nullptr;
    TileDesc *d_td = nullptr;
    auto cleanup = [&]() { if (d_tile_row) (void)hipFree(d_tile_row); if (d_td) (void)hipFree(d_td); d_tile_row = nullptr; d_td = nullptr; };
#define SELL_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); return fail(MMG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } } while (0)
    SELL_TRY(hipMalloc((void **)&d_tile_row, tile_row.size() * sizeof(uint64_t)));
    SELL_TRY(hipMemcpy(d_tile_row, tile_row.data(), tile_row.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    SELL_TRY(hipMalloc((void **)&d_td, nt * sizeof(TileDesc)));
    launch_tile_desc(p->idx64, p->d_row_ptr, p->d_col, p->d_k, d_tile_row, nt, d_td, 0);
    SELL_TRY(hipGetLastError());
    std::vector<TileDesc> td(nt);
    SELL_TRY(hipMemcpy(td.data(), d_td, nt * sizeof(TileDesc), hipMemcpyDeviceToHost));
    const uint32_t WIN = SELL_WIN, BAND_MASK = ~((1u << LAYOUT_BAND_SHIFT) - 1u);
    // a tile qualifies for the register path iff its hits fit one window whose base is a band start and no row exceeds 255 hits
    auto qualifies = [&](const TileDesc &d) {
        return d.nnz > 0 && d.maxlen <= 255 && d.nrows <= 64 && (uint64_t)d.cmax < (uint64_t)(d.call & BAND_MASK) + WIN;
    };
    // the others become far tiles (a fast tile's block for the window hits + a far list) when their rows are short enough and stored
    // window-hits-first: ask the device for the window base their rows were sorted for and the sizes of the two parts
    std::vector<uint32_t> far_wbase(nt, 0), far_nn(nt, 0), far_nf(nt, 0xffffffffu);
    {
        std::vector<uint32_t> cand;
        for (uint64_t t = 0; t < nt; ++t)
            if (td[t].nnz > 0 && !qualifies(td[t]) && td[t].maxlen <= 255 && td[t].nrows <= 64) cand.push_back((uint32_t)t);
        if (!cand.empty() && d_key) {
            uint32_t *d_cand = nullptr, *d_out = nullptr;
            auto cleanup2 = [&]() { if (d_cand) (void)hipFree(d_cand); if (d_out) (void)hipFree(d_out); };
            const size_t nc = cand.size();
            hipError_t e = hipMalloc((void **)&d_cand, nc * 4);
            if (e == hipSuccess) e = hipMalloc((void **)&d_out, nc * 12);
            if (e == hipSuccess) e = hipMemcpy(d_cand, cand.data(), nc * 4, hipMemcpyHostToDevice);
            std::vector<uint32_t> out(nc * 3);
            if (e == hipSuccess) {
                launch_tile_far(p->idx64, p->d_row_ptr, p->d_col, d_key, d_tile_row, d_cand, nc, d_out, 0);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
            cleanup2();
            if (e != hipSuccess) { cleanup(); return fail(MMG_ERR_HIP, std::string("far tiles: ") + hipGetErrorString(e)); }
            for (size_t i = 0; i < nc; ++i) { far_wbase[cand[i]] = out[i]; far_nn[cand[i]] = out[nc + i]; far_nf[cand[i]] = out[2 * nc + i]; }
        }
    }
    auto is_far = [&](uint64_t t) { return far_nf[t] <= 255u; };
    cleanup();
#undef SELL_TRY
    auto resident_raw = [&](bool has_k) { // workgroups of the kernel the device holds at once
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k1_sell_kernel(p->idx64, has_k, false), 64, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 16; }
        if (per_cu > 32) per_cu = 32;
        if (opt(MMG_OPT_SELL_WAVES_PER_CU) >= 1 && opt(MMG_OPT_SELL_WAVES_PER_CU) < per_cu) per_cu = opt(MMG_OPT_SELL_WAVES_PER_CU);
        return (uint64_t)p->cu_count * per_cu;
    };
    auto resident_grid = [&](bool has_k) { return std::max<uint64_t>(1, std::min<uint64_t>(nt, resident_raw(has_k))); };
    // As many workgroups as fit at once while a range is short (every workgroup pays for a window load and the fill of its prefetch
    // pipeline: at config 2 a range is 10 tiles); once ranges are long, several generations of them -- the later generations start as
    // the first ones finish and even out the tail.  Round 2 (K1 bound by instruction issue) measured ranges of about 24 tiles best
    // (config 3: 109 tiles per range in one generation -> 27 in four, -2 to -4 %).  K1 is bound by HBM traffic now, and every range
    // costs a window load and a flush of its counts (0.14 GB written per launch in five generations): 36 tiles per range (three
    // generations at config 3) 0.2387 ms against 0.2467 at 24 and 0.2400 at 48; the pair kernel 1.617 / 1.630 / 1.617 ms.
    auto generations = [&](uint64_t resident) {
        if (opt(MMG_OPT_SELL_WAVES_PER_CU) >= 1) return resident;
        const uint64_t g = (nt + resident * (SELL_TILES_PER_RANGE / 2)) / (resident * SELL_TILES_PER_RANGE); // nearest
        return std::min<uint64_t>(nt, resident * std::min<uint64_t>(16, std::max<uint64_t>(1, g)));
    };
    const uint64_t grid = generations(resident_grid(false));
    // h_sell_cum: what a tile costs whatever its multiplicities (the EM kernel's ranges, the choice of the kernel); cum1 / cumk: the
    // ranges of the two sample launches of a problem with multiplicities -- tiles whose rows all have k = 1, and the others, where a
    // row draws k times (k <= K_SMALL) or runs a binomial per hit.  Identical reads pile up on few hit sets of few abundant
    // transcripts: without their cost in the ranges one workgroup ends up with most of them (20 x the kernel time at config 3).
    auto has_k_rows = [&](uint64_t t) { return p->d_k != nullptr && td[t].knot1 > 0; };
    p->h_sell_cum.assign(nt + 1, 0);
    std::vector<uint64_t> cum1(nt + 1, 0), cumk(nt + 1, 0);
    uint64_t n_hask = 0;
    for (uint64_t t = 0; t < nt; ++t) {
        const uint64_t c = td[t].nnz == 0 ? 0 : qualifies(td[t]) ? SELL_FAST_TILE_COST : is_far(t) ? SELL_FAR_TILE_COST + SELL_FAR_ENTRY_COST * far_nf[t] : SELL_SLOW_TILE_COST;
        p->h_sell_cum[t + 1] = p->h_sell_cum[t] + c;
        const bool hk = has_k_rows(t);
        n_hask += hk;
        cum1[t + 1] = cum1[t] + (hk ? 0 : c);
        // a draw costs about as much as a register-path tile per 64 rows (2)
        // (the tile's rows on the binomial chain are skipped here -- the list kernel draws them: what is left draws at most K_SMALL times)
        cumk[t + 1] = cumk[t] + (!hk ? 0 : 2 * c + 2 * (uint64_t)std::min<uint32_t>(td[t].kmax, K_SMALL));
    }
    // What a sweep over the tiles [0, t) costs, for the cut into read shards (mmg_problem_shard_bounds).  The ranges of ONE launch get by
    // with "2 per register-path tile": several generations of short ranges even out what the model misses.  A shard is one range per
    // device and the slowest device sets the pace, so here a register-path tile is priced by what bounds it -- the bytes of its block,
    // ng groups of 256 (K1's time per tile at 2.4 and 5.4 groups per tile: 2.0 and 4.7 us per million rows, i.e. proportional) -- a far
    // tile by its block plus SHARD_FAR_ENTRY_COST per entry of the far list (64 gathers from L2), a CSR-walked tile by the model above.
    p->h_shard_cum.assign(nt + 1, 0);
    for (uint64_t t = 0; t < nt; ++t) {
        uint64_t c = 0;
        if (td[t].nnz) {
            const uint64_t blk = SHARD_TILE_COST + SHARD_GROUP_COST * (qualifies(td[t]) ? (td[t].maxlen + 3) / 4 : is_far(t) ? (far_nn[t] + 3) / 4 : 0);
            c = qualifies(td[t]) ? blk : is_far(t) ? blk + SHARD_FAR_ENTRY_COST * far_nf[t] : SHARD_SLOW_TILE_COST;
            if (has_k_rows(t)) // draws of the tile's rows that draw; its rows on the chain (at most the knot1 rows with k != 1) by their hits
                c = 2 * c + SHARD_DRAW_COST * (uint64_t)std::min<uint32_t>(td[t].kmax, K_SMALL) +
                    (draws_categoricals(td[t].kmax, td[t].maxlen > 1 ? td[t].maxlen : 2) ? 0 : SHARD_CHAIN_HIT_COST_NUM * (uint64_t)td[t].maxlen * td[t].knot1 / SHARD_CHAIN_HIT_COST_DEN);
        }
        p->h_shard_cum[t + 1] = p->h_shard_cum[t] + c;
    }
    p->h_tile_row = tile_row;
    std::vector<uint64_t> chunk;
    if (n_hask) weighted_chunks(p->h_sell_cum, grid, chunk); // (with SELL_HASK tiles: only the windows below follow these ranges)
    else weighted_chunks_tapered(cum1, grid, opt(MMG_OPT_SELL_WAVES_PER_CU) >= 1 ? 0 : resident_grid(false), chunk);
    const uint64_t n_ranges = chunk.size() - 1;
    std::vector<SellTile> st(nt);
    uint64_t n_fast = 0, n_far = 0, pos = 0, slots = 0;
    for (uint64_t c = 0; c < n_ranges; ++c) {
        bool have = false;
        uint32_t cur = 0;
        for (uint64_t t = chunk[c]; t < chunk[c + 1]; ++t) if (td[t].nnz) { cur = is_far(t) ? far_wbase[t] : td[t].call & BAND_MASK; break; }
        for (uint64_t t = chunk[c]; t < chunk[c + 1]; ++t) {
            const TileDesc &d = td[t];
            SellTile &q = st[t];
            q.off16 = 0; q.r0 = d.r0;
            if (d.nnz == 0) { q.meta = sell_meta(d.nrows, 0, SELL_EMPTY); q.wbase = cur; continue; }
            const uint32_t hask = has_k_rows(t) ? SELL_HASK : 0u;
            if (is_far(t)) { // its own window: the one its bytes are relative to
                cur = far_wbase[t]; have = true;
                q.wbase = cur;
                const uint32_t ng = (far_nn[t] + 3) / 4;
                q.meta = sell_meta(d.nrows, ng, SELL_FAR | hask, far_nf[t]);
                q.off16 = pos;
                pos += 16 * (uint64_t)ng + 4 + 16 * (uint64_t)far_nf[t];
                slots += 256 * (uint64_t)ng + 64 * (uint64_t)far_nf[t];
                ++n_far;
                continue;
            }
            // keep the window in force when the whole tile lies inside it; otherwise slide to the band start of its smallest id
            const bool inside = have && d.call >= cur && (uint64_t)d.cmax < (uint64_t)cur + WIN;
            if (!inside) { cur = d.call & BAND_MASK; have = true; }
            q.wbase = cur;
            const bool fast = d.maxlen <= 255 && d.nrows <= 64 && (uint64_t)d.cmax < (uint64_t)cur + WIN;
            if (fast) {
                const uint32_t ng = (d.maxlen + 3) / 4;
                q.meta = sell_meta(d.nrows, ng, SELL_FAST | hask);
                q.off16 = pos;
                pos += 16 * (uint64_t)ng;
                slots += 256 * (uint64_t)ng;
                ++n_fast;
            } else {
                q.meta = sell_meta(d.nrows, 0, hask);
            }
        }
    }
    // tiles that do not qualify are walked from their far lists or from the CSR inside the same kernel (far rows are sorted last:
    // a contiguous tail); a problem whose tiles cost more than the CSR kernel's on average -- hits uniform over all transcripts -- runs there
    p->use_sell = p->h_sell_cum[nt] <= p->nnz * 7 / 160 || opt(MMG_OPT_SAMPLE_KERNEL) == 2;
    if (!p->use_sell) { p->h_sell_cum.clear(); p->h_shard_cum.clear(); p->h_tile_row.clear(); return MMG_OK; }
    p->sell_bytes = pos * 16;
    p->n_sell_tiles = nt;
    p->n_fast_tiles = n_fast;
    p->n_far_tiles = n_far;
    p->padded_slots = slots;
    p->k1_fixed_walk = n_fast > 0 && slots < 5 * 256 * n_fast; // fewer than 5 groups per register-path tile on average (sell_kernels.h: FIXW)
    const size_t alloc = p->sell_bytes + 64 + 8 * 256; // head room: tiles without a block prefetch the head of the stream
    HIP_TRY(hipMalloc((void **)&p->d_sell, alloc));
    HIP_TRY(hipMemset(p->d_sell, 0, alloc));
    HIP_TRY(hipMalloc((void **)&p->d_sell_tiles, nt * sizeof(SellTile)));
    HIP_TRY(hipMemcpy(p->d_sell_tiles, st.data(), nt * sizeof(SellTile), hipMemcpyHostToDevice));
    p->grid_sell = (int)n_ranges;
    p->n_hask_tiles = n_hask;
    // many ranges per band: the flushes of neighbouring workgroups meet at the same addresses (mmg_types.h: CNT_REPLICAS)
    p->cnt_replicas = n_ranges >= CNT_REPLICA_RANGES_PER_BAND * segs.size() ? CNT_REPLICAS : 1u;
    if (n_hask) { // two launches, each over its own descriptor list and ranges
        std::vector<SellTile> s1, sk;
        std::vector<uint64_t> c1(1, 0), ck(1, 0);
        s1.reserve(nt - n_hask); sk.reserve(n_hask);
        for (uint64_t t = 0; t < nt; ++t) {
            if (st[t].flags() & SELL_HASK) { sk.push_back(st[t]); ck.push_back(ck.back() + (cumk[t + 1] - cumk[t])); p->h_listk_tile.push_back((uint32_t)t); }
            else { s1.push_back(st[t]); c1.push_back(c1.back() + (cum1[t + 1] - cum1[t])); p->h_list1_tile.push_back((uint32_t)t); }
        }
        const uint64_t g1 = std::max<uint64_t>(1, std::min<uint64_t>(s1.size(), grid)), gk = std::max<uint64_t>(1, std::min<uint64_t>(sk.size(), 4 * resident_grid(true))); // (a multiplicity tile costs tens of register-path tiles: short ranges, several generations, even out the tail)
        if (s1.empty()) { SellTile e; e.off16 = 0; e.r0 = 0; e.wbase = 0; e.meta = sell_meta(0, 0, SELL_EMPTY); s1.push_back(e); c1.push_back(0); }
        std::vector<uint64_t> r1, rk;
        weighted_chunks(c1, g1, r1);
        weighted_chunks(ck, gk, rk);
        chunk = r1;
        HIP_TRY(hipMalloc((void **)&p->d_sell_tiles_1, s1.size() * sizeof(SellTile)));
        HIP_TRY(hipMemcpy(p->d_sell_tiles_1, s1.data(), s1.size() * sizeof(SellTile), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&p->d_sell_tiles_k, sk.size() * sizeof(SellTile)));
        HIP_TRY(hipMemcpy(p->d_sell_tiles_k, sk.data(), sk.size() * sizeof(SellTile), hipMemcpyHostToDevice));
        HIP_TRY(upload_ranges(rk, sk, &p->d_sell_chunk_k));
        HIP_TRY(upload_ranges(r1, s1, &p->d_sell_chunk));
        p->h_cum1 = c1; p->h_cumk = ck;
        p->grid_sell_k = (int)gk;
        p->grid_sell = (int)g1;
        p->device_bytes += (s1.size() + sk.size()) * sizeof(SellTile);
    }
    if (!n_hask) p->h_cum1 = cum1;
    p->resident1 = resident_raw(false); p->residentk = resident_raw(true);
    if (!n_hask) HIP_TRY(upload_ranges(chunk, st, &p->d_sell_chunk)); // (with SELL_HASK tiles: uploaded above, over the list without them)
    { // chains in pairs (k_sample_sell_multi): the register-path tiles without multiplicities in their own ranges, the rest apart
        std::vector<SellTile> sf, sx;
        std::vector<uint64_t> cf(1, 0), cx(1, 0);
        for (uint64_t t = 0; t < nt; ++t) {
            const uint32_t fl = st[t].flags();
            if (fl & (SELL_HASK | SELL_EMPTY)) continue;
            const uint64_t c = p->h_sell_cum[t + 1] - p->h_sell_cum[t];
            if (fl & SELL_FAST) { sf.push_back(st[t]); cf.push_back(cf.back() + c); }
            else { sx.push_back(st[t]); cx.push_back(cx.back() + c); }
        }
        p->n_x_tiles = sx.size();
        if (!sf.empty()) {
            if (sf.size() == nt) p->d_sell_tiles_f = p->d_sell_tiles;           // every tile: the list that exists
            else {
                HIP_TRY(hipMalloc((void **)&p->d_sell_tiles_f, sf.size() * sizeof(SellTile)));
                HIP_TRY(hipMemcpy(p->d_sell_tiles_f, sf.data(), sf.size() * sizeof(SellTile), hipMemcpyHostToDevice));
                p->owns_tiles_f = true;
                p->device_bytes += sf.size() * sizeof(SellTile);
            }
            const uint64_t nf = sf.size();
            for (int q = 0; q < 2; ++q) {
                int pc = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, k1_sell_multi_kernel(p->idx64, 2 << q), 64, 0) != hipSuccess || pc < 1) { (void)hipGetLastError(); pc = 8; }
                if (pc > 32) pc = 32;
                if (opt(MMG_OPT_SELL_WAVES_PER_CU) >= 1 && opt(MMG_OPT_SELL_WAVES_PER_CU) < pc) pc = opt(MMG_OPT_SELL_WAVES_PER_CU);
                const uint64_t rq = std::max<uint64_t>(1, std::min<uint64_t>(nf, (uint64_t)p->cu_count * pc));
                const uint64_t gen = opt(MMG_OPT_SELL_WAVES_PER_CU) >= 1 ? rq : std::min<uint64_t>(nf, rq * std::min<uint64_t>(16, std::max<uint64_t>(1, (nf + rq * (SELL_TILES_PER_RANGE / 2)) / (rq * SELL_TILES_PER_RANGE))));
                std::vector<uint64_t> cq;
                weighted_chunks_tapered(cf, gen, opt(MMG_OPT_SELL_WAVES_PER_CU) >= 1 ? 0 : rq, cq);
                HIP_TRY(upload_ranges(cq, sf, &p->d_sell_chunk_m[q]));
                p->grid_sell_m[q] = (int)(cq.size() - 1);
            }
        }
        if (!sx.empty()) {
            HIP_TRY(hipMalloc((void **)&p->d_sell_tiles_x, sx.size() * sizeof(SellTile)));
            HIP_TRY(hipMemcpy(p->d_sell_tiles_x, sx.data(), sx.size() * sizeof(SellTile), hipMemcpyHostToDevice));
            std::vector<uint64_t> rx;
            weighted_chunks(cx, std::max<uint64_t>(1, std::min<uint64_t>(sx.size(), resident_grid(false))), rx);
            HIP_TRY(upload_ranges(rx, sx, &p->d_sell_chunk_x));
            p->grid_sell_x = (int)(rx.size() - 1);
            p->device_bytes += sx.size() * sizeof(SellTile);
        }
    }
    if (p->d_k) { // the rows on the conditional-binomial chain: a list of their own (bigk_kernels.h)
        HIP_TRY(layout_bigk_rows(p->idx64, p->m, p->d_row_ptr, p->d_k, &p->d_bigk_list, &p->n_bigk, 0));
        if (p->n_bigk) {
            p->h_bigk_list.resize(p->n_bigk);
            HIP_TRY(hipMemcpy(p->h_bigk_list.data(), p->d_bigk_list, p->n_bigk * sizeof(uint64_t), hipMemcpyDeviceToHost));
            p->bigk_per_wave = bigk_piece(p->n_bigk, p->cu_count);
            p->grid_bigk = (int)((p->n_bigk + p->bigk_per_wave - 1) / p->bigk_per_wave);
            p->device_bytes += p->n_bigk * sizeof(uint64_t);
        }
    }
    launch_encode_sell(p->idx64, p->d_row_ptr, p->d_col, p->d_sell_tiles, nt, p->d_sell, 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    p->device_bytes += p->sell_bytes + nt * sizeof(SellTile);
    return MMG_OK;
}

// Tiles of consecutive rows for k_sample: <= tile_nnz hits (rows padded to 4) and <= tile_rows rows; a longer row is alone.
static int problem_build_csr_tiles(mmg_problem *p, const uint64_t *d_rp64)
{
    if (p->m == 0) return MMG_OK;
    std::vector<uint64_t> rp(p->m + 1);
    HIP_TRY(hipMemcpy(rp.data(), d_rp64, (p->m + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
    std::vector<uint64_t> tile_row;
    tile_row.push_back(0);
    const uint64_t tile_nnz = (uint64_t)K1C_ELEMS - 8, tile_rows = std::min<uint64_t>(K1C_ROWS, K1C_ELEMS / 4);
    uint64_t cur_nnz = 0, cur_rows = 0;
    for (uint64_t r = 0; r < p->m; ++r) {
        const uint64_t L = rp[r + 1] - rp[r], L4 = (L + 3) & ~(uint64_t)3;
        if (cur_rows > 0 && (cur_nnz + L4 > tile_nnz || cur_rows >= tile_rows)) {
            tile_row.push_back(r);
            cur_nnz = 0;
            cur_rows = 0;
        }
        cur_nnz += L4;
        cur_rows += 1;
    }
    tile_row.push_back(p->m);
    std::vector<uint64_t>().swap(rp);
    p->n_tiles = tile_row.size() - 1;
    if (p->n_tiles >= 0x7fffffffull) return fail(MMG_ERR_ARG, "too many tiles for one device");
    uint64_t *d_tile_row = nullptr;
    HIP_TRY(hipMalloc((void **)&d_tile_row, tile_row.size() * sizeof(uint64_t)));
    hipError_t e = hipMemcpy(d_tile_row, tile_row.data(), tile_row.size() * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_tiles, p->n_tiles * sizeof(TileDesc));
    if (e == hipSuccess) {
        launch_tile_desc(p->idx64, p->d_row_ptr, p->d_col, p->d_k, d_tile_row, p->n_tiles, p->d_tiles, 0);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    (void)hipFree(d_tile_row);
    if (e != hipSuccess) return fail(MMG_ERR_HIP, std::string("tile descriptors: ") + hipGetErrorString(e));
    p->device_bytes += p->n_tiles * sizeof(TileDesc);
    // persistent grid: every resident workgroup walks one contiguous range of tiles
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k1_csr_kernel(p->idx64, p->d_k != nullptr), K1C_BS, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 4; }
    if (per_cu > 2048 / K1C_BS) per_cu = 2048 / K1C_BS;
    const uint64_t grid = std::max<uint64_t>(1, std::min<uint64_t>(p->n_tiles, (uint64_t)p->cu_count * per_cu));
    p->grid_sample = (int)grid;
    std::vector<uint64_t> chunk(grid + 1, 0);
    for (uint64_t c = 0; c <= grid; ++c) chunk[c] = (uint64_t)(((unsigned __int128)p->n_tiles * c) / grid);
    HIP_TRY(hipMalloc((void **)&p->d_chunk_tile, chunk.size() * sizeof(uint64_t)));
    HIP_TRY(hipMemcpy(p->d_chunk_tile, chunk.data(), chunk.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    return MMG_OK;
}

// From a device CSR in the caller's row order (d_rp64: m+1 u64, consumed; p->d_col / p->d_k set) to the finished problem.
static int problem_build(mmg_problem *p, uint64_t *d_rp64)
{
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, p->device);
    if (e != hipSuccess) { (void)hipFree(d_rp64); return fail(MMG_ERR_HIP, "hipGetDeviceProperties"); }
    p->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    uint64_t *d_key = nullptr;
    auto bail = [&](int code) { if (d_key) (void)hipFree(d_key); if (d_rp64) (void)hipFree(d_rp64); return code; };
#define B_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return bail(fail(MMG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e))); } while (0)
    std::vector<uint64_t> seg;
    if (p->m && p->layout == (int)MMG_LAYOUT_CANONICAL && p->d_k) {
        // a row that draws k >= 2 categoricals (mmg_types.h: draws_categoricals) is stored as k rows with k = 1: identical reads then run on the register path like any
        // other read, instead of through the multiplicity kernel (mmg_types.h)
        B_TRY(layout_expand_rows(&p->m, &p->nnz, &d_rp64, &p->d_col, &p->d_k, 16, 0));
    }
    if (p->m) {
        B_TRY(hipMalloc((void **)&d_key, p->m * sizeof(uint64_t)));
        if (p->layout == (int)MMG_LAYOUT_CANONICAL) {
            B_TRY(layout_canonical_sort(p->m, p->nnz, &d_rp64, &p->d_col, p->d_k ? &p->d_k : nullptr, d_key, 16, 0));
        } else {
            B_TRY(layout_row_keys(p->m, d_rp64, p->d_col, p->d_k, d_key, 0));
        }
        // kept rows may be in any order: band-aligned tiles only while the runs of equal band stay long
        // (a shard cut from a canonical problem is in canonical order whatever its size: the canonical bound, not the one for rows in any order)
        if (p->layout == (int)MMG_LAYOUT_CANONICAL) p->canonical_rows = true;
        const uint64_t canon_seg = std::min<uint64_t>(p->m, 2 * ((uint64_t)p->n >> LAYOUT_BAND_SHIFT) + 4);
        const uint64_t max_seg = p->layout == (int)MMG_LAYOUT_CANONICAL ? canon_seg
                                                                          : std::max<uint64_t>(p->canonical_rows ? canon_seg : 0, std::max<uint64_t>(1024, p->m / 32));
        B_TRY(layout_segments(p->m, d_key, max_seg, seg, 0));
        B_TRY(layout_max_row_len(p->m, d_rp64, &p->max_row_len, 0));
    }
    p->idx64 = p->nnz >= 0xffffffffull || opt(MMG_OPT_FORCE_IDX64) == 1;
    if (p->idx64) {
        p->d_row_ptr = d_rp64;
        p->device_bytes += (p->m + 1) * 8;
    } else {
        B_TRY(hipMalloc(&p->d_row_ptr, (p->m + 1) * sizeof(uint32_t)));
        B_TRY(layout_narrow_row_ptr(p->m, d_rp64, (uint32_t *)p->d_row_ptr, 0));
        B_TRY(hipDeviceSynchronize());
        p->device_bytes += (p->m + 1) * 4;
    }
#undef B_TRY
    int rc = problem_build_sell(p, seg, d_key);
    if (d_key) { (void)hipFree(d_key); d_key = nullptr; }
    if (rc == MMG_OK && !p->use_sell) { p->cnt_replicas = 1; rc = problem_build_csr_tiles(p, d_rp64); }
    if (opt(MMG_OPT_CNT_REPLICAS) >= 1) p->cnt_replicas = opt(MMG_OPT_CNT_REPLICAS) > 1 ? CNT_REPLICAS : 1u;
    if (!p->idx64) (void)hipFree(d_rp64);
    return rc;
}

static int problem_create_checked(const mmg_problem_desc *d, int device, const uint64_t *tx_order, mmg_problem **out);

// what a sweep over the problem costs in the units of the tile ranges (2 per register-path tile; the CSR-tile kernel 2.8 per 64 hits)
static uint64_t modelled_sweep_cost(const mmg_problem *p) { return p->use_sell && !p->h_sell_cum.empty() ? p->h_sell_cum.back() : p->nnz * 7 / 160; }

extern "C" int mmg_problem_create(const mmg_problem_desc *d, int device, mmg_problem **out)
{
    if (!d || !out) return fail(MMG_ERR_ARG, "NULL argument");
    if (!d->row_ptr || !d->l || d->n == 0) return fail(MMG_ERR_ARG, "row_ptr/l missing or n == 0");
    if (d->row_ptr[0] != 0) return fail(MMG_ERR_ARG, "row_ptr[0] must be 0");
    if (d->layout != MMG_LAYOUT_CANONICAL && d->layout != MMG_LAYOUT_KEEP_ROWS) return fail(MMG_ERR_ARG, "unknown layout");
    if (d->layout == MMG_LAYOUT_CANONICAL && d->m >= 0xffffffffull) return fail(MMG_ERR_ARG, "the canonical layout needs fewer than 2^32 rows per device");
    const uint64_t nnz = d->row_ptr[d->m];
    if (nnz > 0 && !d->col_idx) return fail(MMG_ERR_ARG, "col_idx missing");
    for (uint64_t r = 0; r < d->m; ++r)
        if (d->row_ptr[r + 1] < d->row_ptr[r]) return fail(MMG_ERR_ARG, "row_ptr must be non-decreasing");
    {
        std::atomic<bool> bad{false};
        parallel_slices(nnz, [&](uint64_t a, uint64_t b) { for (uint64_t j = a; j < b; ++j) if (d->col_idx[j] >= d->n) { bad = true; return; } });
        if (bad) return fail(MMG_ERR_ARG, "col_idx entry out of range");
    }
    for (uint32_t t = 0; t < d->n; ++t)
        if (!(d->l[t] > 0.0)) return fail(MMG_ERR_ARG, "l[t] must be > 0 (src/mmseq.cpp:604)");
    int rc = require_device(device);
    if (rc) return rc;
    mmg_problem *p = nullptr;
    rc = problem_create_checked(d, device, d->tx_order, &p);
    if (rc) return rc;
    // No transcript order from the caller, and in the caller's numbering the rows do not fit LDS windows (first-seen numbering,
    // src/mmseq.cpp:399-408: more than a quarter above what register-path tiles alone would cost): derive an order from the hit
    // graph (order.hip) and keep the problem built on it if the model prices it at least a fifth lower.
    if (!d->tx_order && d->layout == MMG_LAYOUT_CANONICAL && p->m > 0 && d->n > 1 && opt(MMG_OPT_DERIVE_ORDER) != 0) {
        const uint64_t floor_cost = SELL_FAST_TILE_COST * (p->use_sell ? p->n_sell_tiles : (p->m + 63) / 64);
        if (modelled_sweep_cost(p) > floor_cost + floor_cost / 4 || opt(MMG_OPT_DERIVE_ORDER) == 1) {
            // The attempt must never cost the caller the problem it already has: p is valid.  The second build needs p's device memory
            // and its build's temporaries once more, plus the edge keys and their sort (16 bytes per sampled hit twice over, at most
            // 4 GB); without that much free, and on any failure below, p stays.
            size_t free_b = 0, total_b = 0;
            const size_t need = 3 * (size_t)p->device_bytes + std::min<size_t>((size_t)4 << 30, 32 * (size_t)p->nnz + ((size_t)64 << 20));
            // (What stands then is the problem in the caller's order: the chain of such a problem depends on whether the attempt could be
            // made.  mmg_problem_info.tx_renumbered carries MMG_ORDER_SKIPPED in that case; MMG_OPT_DERIVE_ORDER = 1 -- "always" -- fails.)
            const bool forced = opt(MMG_OPT_DERIVE_ORDER) == 1;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < need) {
                (void)hipGetLastError();
                if (forced) { problem_free(p); return fail(MMG_ERR_HIP, "transcript order from the hit graph: needs " + std::to_string(need >> 20) + " MiB of free device memory, " + std::to_string(free_b >> 20) + " MiB are free"); }
                p->order_skipped = true;
                *out = p;
                return MMG_OK;
            }
            std::vector<uint64_t> edges;
            hipError_t e = order_cooccurrence_edges(p->idx64, p->m, p->nnz, p->d_row_ptr, p->d_col, edges, 0); // (no tx_order: device ids are the caller's)
            if (e != hipSuccess) {
                if (forced) { problem_free(p); return fail(MMG_ERR_HIP, std::string("transcript order from the hit graph: ") + hipGetErrorString(e)); }
                (void)hipGetLastError(); edges.clear(); p->order_skipped = true;
            }
            // (a graph in which the average transcript shares rows with more than 1024 others has no band to find: hits drawn all over
            // the transcriptome -- the level structures of 10^8 edges would only cost host time before the result is discarded)
            if (!edges.empty() && edges.size() <= (uint64_t)d->n * 1024) {
                std::vector<uint32_t> pos;
                order_from_edges(d->n, edges, pos);
                std::vector<uint64_t>().swap(edges);
                std::vector<uint64_t> keys(pos.begin(), pos.end());
                mmg_problem *q = nullptr;
                rc = problem_create_checked(d, device, keys.data(), &q);
                if (rc && forced) { problem_free(p); return rc; }
                if (rc) { (void)hipGetLastError(); q = nullptr; rc = MMG_OK; p->order_skipped = true; } // (problem_create_checked frees what it built)
                if (q && modelled_sweep_cost(q) < modelled_sweep_cost(p) - modelled_sweep_cost(p) / 5) { problem_free(p); p = q; p->order_derived = true; }
                else if (q) problem_free(q);
            }
        }
    }
    // A transcript order from the caller whose keys name GROUPS (high 32 bits: the gene, src/mmseq.cpp:337-357), and rows that do not
    // fit LDS windows in the caller's order of the groups -- reads that also hit a paralogue, whose gene a name-sorted gene table puts
    // anywhere: derive an order of the groups from the group-level hit graph (spec version 7).  The transcripts of a group stay
    // together in the caller's order; groups that share no row with another keep their relative order behind the linked ones.
    if (d->tx_order && d->layout == MMG_LAYOUT_CANONICAL && p->m > 0 && d->n > 2 && opt(MMG_OPT_DERIVE_ORDER) != 0) {
        const uint64_t floor_cost = SELL_FAST_TILE_COST * (p->use_sell ? p->n_sell_tiles : (p->m + 63) / 64);
        if (modelled_sweep_cost(p) > floor_cost + floor_cost / 4 || opt(MMG_OPT_DERIVE_ORDER) == 1) {
            std::vector<uint32_t> gkeys(d->n);
            for (uint32_t t = 0; t < d->n; ++t) gkeys[t] = (uint32_t)(d->tx_order[t] >> 32);
            std::vector<uint32_t> uniq(gkeys);
            std::sort(uniq.begin(), uniq.end());
            uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
            const uint32_t nG = (uint32_t)uniq.size();
            size_t free_b = 0, total_b = 0;
            const size_t need = 3 * (size_t)p->device_bytes + std::min<size_t>((size_t)4 << 30, 32 * (size_t)p->nnz + ((size_t)64 << 20));
            const bool forced = opt(MMG_OPT_DERIVE_ORDER) == 1;
            const bool have_mem = hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b >= need;
            if (nG > 1 && nG < d->n && !have_mem) {
                (void)hipGetLastError();
                if (forced) { problem_free(p); return fail(MMG_ERR_HIP, "group order from the hit graph: needs " + std::to_string(need >> 20) + " MiB of free device memory, " + std::to_string(free_b >> 20) + " MiB are free"); }
                p->order_skipped = true;
            }
            if (nG > 1 && nG < d->n && have_mem) {
                std::vector<uint32_t> group_of_ext(d->n), label(d->n);
                for (uint32_t t = 0; t < d->n; ++t) group_of_ext[t] = (uint32_t)(std::lower_bound(uniq.begin(), uniq.end(), gkeys[t]) - uniq.begin());
                for (uint32_t i = 0; i < d->n; ++i) label[i] = group_of_ext[p->h_ext_of_int[i]];   // device column id -> group
                uint32_t *d_label = nullptr;
                std::vector<uint64_t> edges;
                hipError_t e = hipMalloc((void **)&d_label, (size_t)d->n * 4);
                if (e == hipSuccess) e = hipMemcpy(d_label, label.data(), (size_t)d->n * 4, hipMemcpyHostToDevice);
                if (e == hipSuccess) e = order_cooccurrence_edges(p->idx64, p->m, p->nnz, p->d_row_ptr, p->d_col, edges, 0, d_label);
                if (d_label) (void)hipFree(d_label);
                if (e != hipSuccess) {
                    if (forced) { problem_free(p); return fail(MMG_ERR_HIP, std::string("group order from the hit graph: ") + hipGetErrorString(e)); }
                    (void)hipGetLastError(); edges.clear(); p->order_skipped = true;
                }
                if (!edges.empty() && edges.size() <= (uint64_t)nG * 1024) {
                    std::vector<uint32_t> posg;
                    order_from_edges(nG, edges, posg, 32); // (groups: a gene linked to more than 32 others is a hub, spec version 8)
                    std::vector<uint64_t>().swap(edges);
                    std::vector<uint64_t> keys(d->n);
                    for (uint32_t t = 0; t < d->n; ++t) keys[t] = ((uint64_t)posg[group_of_ext[t]] << 32) | p->h_int_of_ext[t];
                    mmg_problem *q = nullptr;
                    rc = problem_create_checked(d, device, keys.data(), &q);
                    if (rc && forced) { problem_free(p); return rc; }
                    if (rc) { (void)hipGetLastError(); q = nullptr; rc = MMG_OK; p->order_skipped = true; }
                    if (q && modelled_sweep_cost(q) < modelled_sweep_cost(p) - modelled_sweep_cost(p) / 5) { problem_free(p); p = q; p->groups_reordered = true; }
                    else if (q) problem_free(q);
                }
            }
        }
    }
    *out = p;
    return MMG_OK;
}

// the upload and build behind mmg_problem_create, arguments checked; tx_order: the caller's, a derived one, or NULL
static int problem_create_checked(const mmg_problem_desc *d, int device, const uint64_t *tx_order, mmg_problem **out)
{
    const uint64_t nnz = d->row_ptr[d->m];
    int rc = MMG_OK;
    mmg_problem *p = new mmg_problem();
    p->device = device;
    p->m = d->m; p->n = d->n; p->nnz = nnz; p->row_id_base = d->row_id_base; p->layout = (int)d->layout;
    p->h_l.assign(d->l, d->l + d->n);
    for (uint64_t r = 0; r < d->m; ++r) {
        const uint64_t kr = d->k ? d->k[r] : 1;
        p->total_k += kr;
        if (d->row_ptr[r + 1] > d->row_ptr[r]) p->total_k_hit += kr;
    }
    uint64_t *d_rp64 = nullptr;
    auto bail = [&](int code) { if (d_rp64) (void)hipFree(d_rp64); problem_free(p); return code; };
#define C_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return bail(fail(MMG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e))); } while (0)
    // transcript renumbering: device id = rank of (tx_order[t], t).  The hits are renumbered on the device; the canonical layout then
    // sorts every row's hits (layout.hip), kept rows keep the order they came in (a stored far row: window hits first).
    const uint32_t *col_src = d->col_idx;
    std::vector<double> l_int;
    const double *l_src = d->l;
    if (tx_order) {
        std::vector<uint32_t> order(d->n);
        for (uint32_t t = 0; t < d->n; ++t) order[t] = t;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return tx_order[a] < tx_order[b]; });
        p->h_ext_of_int = order;
        p->h_int_of_ext.resize(d->n);
        for (uint32_t i = 0; i < d->n; ++i) p->h_int_of_ext[order[i]] = i;
        l_int.resize(d->n);
        for (uint32_t i = 0; i < d->n; ++i) l_int[i] = d->l[order[i]];
        l_src = l_int.data();
        C_TRY(hipMalloc((void **)&p->d_int_of_ext, d->n * 4));
        C_TRY(hipMalloc((void **)&p->d_ext_of_int, d->n * 4));
        C_TRY(hipMemcpy(p->d_int_of_ext, p->h_int_of_ext.data(), d->n * 4, hipMemcpyHostToDevice));
        C_TRY(hipMemcpy(p->d_ext_of_int, p->h_ext_of_int.data(), d->n * 4, hipMemcpyHostToDevice));
        p->device_bytes += d->n * 8;
    }
    const size_t col_bytes = (nnz + 16) * sizeof(uint32_t); // padded: vector loads may over-read
    C_TRY(hipMalloc((void **)&p->d_col, col_bytes));
    C_TRY(hipMemset(p->d_col, 0, col_bytes));
    if (nnz) C_TRY(hipMemcpy(p->d_col, col_src, nnz * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (nnz && p->d_int_of_ext) C_TRY(layout_map_cols(nnz, p->d_col, p->d_int_of_ext, 0));
    p->device_bytes += col_bytes;
    if (d->k && d->m) {
        C_TRY(hipMalloc((void **)&p->d_k, d->m * sizeof(uint32_t)));
        C_TRY(hipMemcpy(p->d_k, d->k, d->m * sizeof(uint32_t), hipMemcpyHostToDevice));
        p->device_bytes += d->m * 4;
    }
    C_TRY(hipMalloc((void **)&p->d_l, d->n * sizeof(double)));
    C_TRY(hipMemcpy(p->d_l, l_src, d->n * sizeof(double), hipMemcpyHostToDevice));
    p->device_bytes += d->n * 8;
    C_TRY(hipMalloc((void **)&d_rp64, (d->m + 1) * sizeof(uint64_t)));
    C_TRY(hipMemcpy(d_rp64, d->row_ptr, (d->m + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
#undef C_TRY
    uint64_t *rp = d_rp64;
    d_rp64 = nullptr; // consumed by problem_build
    rc = problem_build(p, rp);
    if (rc) return bail(rc);
    *out = p;
    return MMG_OK;
}

// Rows [lo, hi) of a stored problem as a problem of its own on `device` -- a read shard (row_id_base = the parent's + lo) or, with
// lo = 0 and hi = m, a replica -- cut on the parent's device and copied device to device (peer copy over xGMI): the rows, their hit
// order and the transcript numbering stay as stored, nothing passes through the host.
extern "C" int mmg_problem_shard(const mmg_problem *full, uint64_t lo, uint64_t hi, int device, mmg_problem **out)
{
    if (!full || !out) return fail(MMG_ERR_ARG, "NULL argument");
    if (lo > hi || hi > full->m) return fail(MMG_ERR_ARG, "row range out of bounds");
    int rc = require_device(device);
    if (rc) return rc;
    mmg_problem *p = new mmg_problem();
    p->device = device;
    p->m = hi - lo; p->n = full->n; p->row_id_base = full->row_id_base + lo; p->layout = (int)MMG_LAYOUT_KEEP_ROWS;
    p->h_l = full->h_l;
    p->h_int_of_ext = full->h_int_of_ext; p->h_ext_of_int = full->h_ext_of_int;
    p->canonical_rows = full->canonical_rows; p->order_derived = full->order_derived;
    uint64_t *src_rp = nullptr, *d_rp64 = nullptr;
    auto bail = [&](int code) {
        if (src_rp) { (void)hipSetDevice(full->device); (void)hipFree(src_rp); }
        if (d_rp64) { (void)hipSetDevice(device); (void)hipFree(d_rp64); }
        problem_free(p);
        return code;
    };
#define SH_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return bail(fail(MMG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e))); } while (0)
    // the shard's row offsets, rebased on the parent's device
    SH_TRY(hipSetDevice(full->device));
    SH_TRY(hipMalloc((void **)&src_rp, (p->m + 1) * sizeof(uint64_t)));
    SH_TRY(layout_rebase_row_ptr(full->idx64, full->d_row_ptr, lo, p->m, src_rp, 0));
    uint64_t nz0 = 0, nz1 = 0;
    if (full->idx64) {
        SH_TRY(hipMemcpy(&nz0, (const uint64_t *)full->d_row_ptr + lo, 8, hipMemcpyDeviceToHost));
        SH_TRY(hipMemcpy(&nz1, (const uint64_t *)full->d_row_ptr + hi, 8, hipMemcpyDeviceToHost));
    } else {
        uint32_t a = 0, b = 0;
        SH_TRY(hipMemcpy(&a, (const uint32_t *)full->d_row_ptr + lo, 4, hipMemcpyDeviceToHost));
        SH_TRY(hipMemcpy(&b, (const uint32_t *)full->d_row_ptr + hi, 4, hipMemcpyDeviceToHost));
        nz0 = a; nz1 = b;
    }
    p->nnz = nz1 - nz0;
    SH_TRY(hipSetDevice(device));
    SH_TRY(hipMalloc((void **)&d_rp64, (p->m + 1) * sizeof(uint64_t)));
    SH_TRY(hipMemcpyPeer(d_rp64, device, src_rp, full->device, (p->m + 1) * sizeof(uint64_t)));
    const size_t col_bytes = (p->nnz + 16) * sizeof(uint32_t);
    SH_TRY(hipMalloc((void **)&p->d_col, col_bytes));
    SH_TRY(hipMemset(p->d_col, 0, col_bytes));
    if (p->nnz) SH_TRY(hipMemcpyPeer(p->d_col, device, full->d_col + nz0, full->device, p->nnz * sizeof(uint32_t)));
    p->device_bytes += col_bytes;
    if (full->d_k && p->m) {
        SH_TRY(hipMalloc((void **)&p->d_k, p->m * sizeof(uint32_t)));
        SH_TRY(hipMemcpyPeer(p->d_k, device, full->d_k + lo, full->device, p->m * sizeof(uint32_t)));
        p->device_bytes += p->m * 4;
        std::vector<uint32_t> hk(p->m); // total_k of the shard (one pass over its multiplicities)
        SH_TRY(hipMemcpy(hk.data(), p->d_k, p->m * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint32_t v : hk) p->total_k += v;
        std::vector<uint64_t> hrp(p->m + 1); // ... and of its rows with a hit (the wire check of a group compares a sweep's counts with them)
        SH_TRY(hipMemcpy(hrp.data(), d_rp64, (p->m + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
        for (uint64_t r = 0; r < p->m; ++r) if (hrp[r + 1] > hrp[r]) p->total_k_hit += hk[r];
    } else {
        p->total_k = p->m;
        std::vector<uint64_t> hrp(p->m + 1);
        if (p->m) SH_TRY(hipMemcpy(hrp.data(), d_rp64, (p->m + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
        for (uint64_t r = 0; r < p->m; ++r) p->total_k_hit += hrp[r + 1] > hrp[r];
    }
    SH_TRY(hipMalloc((void **)&p->d_l, p->n * sizeof(double)));
    {
        std::vector<double> l_int;
        to_int(p, p->h_l.data(), l_int);
        SH_TRY(hipMemcpy(p->d_l, l_int.data(), p->n * sizeof(double), hipMemcpyHostToDevice));
    }
    p->device_bytes += p->n * 8;
    if (p->renumbered()) {
        SH_TRY(hipMalloc((void **)&p->d_int_of_ext, p->n * 4));
        SH_TRY(hipMalloc((void **)&p->d_ext_of_int, p->n * 4));
        SH_TRY(hipMemcpy(p->d_int_of_ext, p->h_int_of_ext.data(), p->n * 4, hipMemcpyHostToDevice));
        SH_TRY(hipMemcpy(p->d_ext_of_int, p->h_ext_of_int.data(), p->n * 4, hipMemcpyHostToDevice));
        p->device_bytes += p->n * 8;
    }
    SH_TRY(hipSetDevice(full->device));
    SH_TRY(hipFree(src_rp));
    src_rp = nullptr;
    SH_TRY(hipSetDevice(device));
#undef SH_TRY
    uint64_t *rp = d_rp64;
    d_rp64 = nullptr; // consumed by problem_build
    rc = problem_build(p, rp);
    if (rc) return bail(rc);
    *out = p;
    return MMG_OK;
}

// Contiguous ranges of the stored rows for `parts` read shards, of (nearly) equal modelled COST of a sweep (h_shard_cum): the sharded
// chain advances at the pace of its slowest device, and the canonical order puts every far row behind every near row -- equal hit
// counts (mmg_shard_bounds; the reference's schedule(static) over rows, src/mmseq.cpp:864, has the same weakness) would hand the last
// devices nothing but far tiles at 3 x the cost.  Boundaries are tile starts rounded up to even rows (a Philox block serves rows 2q and
// 2q + 1).  Problems on the CSR-tile kernel (its cost follows the hits) are cut by hits.
extern "C" int mmg_problem_shard_bounds(const mmg_problem *p, int parts, uint64_t *bounds)
{
    if (!p || !bounds || parts < 1) return fail(MMG_ERR_ARG, "bad argument");
    if (p->use_sell && !p->h_shard_cum.empty() && p->h_shard_cum.back() > 0) {
        const std::vector<uint64_t> &cum = p->h_shard_cum;
        const uint64_t nt = cum.size() - 1, total = cum[nt];
        bounds[0] = 0;
        uint64_t t = 0;
        for (int i = 1; i < parts; ++i) {
            const uint64_t target = (uint64_t)(((unsigned __int128)total * (uint64_t)i) / (uint64_t)parts);
            while (t < nt && cum[t] < target) ++t;
            // the tile boundary nearest to the target in cost (cum[t - 1] < target <= cum[t])
            const uint64_t tb = (t > 0 && target - cum[t - 1] < cum[t] - target) ? t - 1 : t;
            uint64_t r = tb < nt ? p->h_tile_row[tb] : p->m;
            r = std::min<uint64_t>(p->m, (r + ((p->row_id_base + r) & 1)));   // even random-stream id: a Philox block stays on one device
            bounds[i] = std::max<uint64_t>(r, bounds[i - 1]);
        }
        bounds[parts] = p->m;
        return MMG_OK;
    }
    HIP_TRY(hipSetDevice(p->device));
    std::vector<uint64_t> rp(p->m + 1);
    if (p->idx64) HIP_TRY(hipMemcpy(rp.data(), p->d_row_ptr, (p->m + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
    else {
        std::vector<uint32_t> rp32(p->m + 1);
        HIP_TRY(hipMemcpy(rp32.data(), p->d_row_ptr, (p->m + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i <= p->m; ++i) rp[i] = rp32[i];
    }
    return mmg_shard_bounds(rp.data(), p->m, parts, bounds);
}

// The cut by MEASURED cost.  Every candidate shard is run as what it will be -- the one-chain sample kernels over ITS interval of the
// tile lists, in ranges cut the way problem_build_sell cuts a problem of that size -- on the parent's device, with the weights mu
// (the start values or the EM optimum: which transcripts are popular decides what the count flushes cost), timed with HIP events.
// Tiles of a shard that ran long get dearer in proportion, the rows are cut again, and so on until the slowest shard is within 2 % of
// the mean (at most 6 rounds; a round is parts x 7 launches).  What the model cannot know -- what a far entry costs on this part, hit
// sets that give most of their reads to one transcript -- is in the measurement.  The chain does not depend on the cut (any cut gives
// the same bits), so a cut that differs from run to run is harmless.
extern "C" int mmg_problem_shard_bounds_timed(const mmg_problem *p, const double *mu, int parts, uint64_t *bounds)
{
    if (!p || !mu || !bounds || parts < 1) return fail(MMG_ERR_ARG, "bad argument");
    if (parts == 1 || !p->use_sell || p->h_shard_cum.empty() || p->h_shard_cum.back() == 0 || p->h_cum1.size() < 2 || p->resident1 == 0)
        return mmg_problem_shard_bounds(p, parts, bounds);
    HIP_TRY(hipSetDevice(p->device));
    const uint64_t nt = p->h_shard_cum.size() - 1;
    const bool split = p->h_cumk.size() > 1;                          // a list of multiplicity tiles with a launch of its own
    const size_t L1 = p->h_cum1.size() - 1, Lk = split ? p->h_cumk.size() - 1 : 0;
    double *d_mu = nullptr;
    int32_t *d_cnt = nullptr;
    std::vector<uint64_t *> d_hdr;                                    // range headers of the round's launches
    std::vector<hipEvent_t> ev;
    auto cleanup = [&]() {
        for (void *x : {(void *)d_mu, (void *)d_cnt}) if (x) (void)hipFree(x);
        for (uint64_t *h : d_hdr) if (h) (void)hipFree(h);
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        d_hdr.clear(); ev.clear(); d_mu = nullptr; d_cnt = nullptr;
    };
#define T_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); return fail(MMG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } } while (0)
    std::vector<double> mu_int;
    to_int(p, mu, mu_int);
    T_TRY(hipMalloc((void **)&d_mu, p->n * sizeof(double)));
    T_TRY(hipMalloc((void **)&d_cnt, (size_t)CNT_REPLICAS * p->n * sizeof(int32_t)));
    T_TRY(hipMemcpy(d_mu, mu_int.data(), p->n * sizeof(double), hipMemcpyHostToDevice));
    T_TRY(hipMemset(d_cnt, 0, (size_t)CNT_REPLICAS * p->n * sizeof(int32_t)));
    // the tile lists as the device holds them (the range headers carry the first two descriptors of a range)
    std::vector<SellTile> t1(std::max<size_t>(L1, 1)), tk(std::max<size_t>(Lk, 1));
    const SellTile *d_t1 = split ? p->d_sell_tiles_1 : p->d_sell_tiles;
    T_TRY(hipMemcpy(t1.data(), d_t1, L1 * sizeof(SellTile), hipMemcpyDeviceToHost));
    if (split) T_TRY(hipMemcpy(tk.data(), p->d_sell_tiles_k, Lk * sizeof(SellTile), hipMemcpyDeviceToHost));
    const int reps = 7;
    ev.resize((size_t)parts * reps * 2);
    for (auto &e : ev) { e = nullptr; T_TRY(hipEventCreate(&e)); }
    // entries [lo, hi) of a list as ranges: what problem_build_sell does for a problem of hi - lo tiles
    auto ranges_of = [&](const std::vector<uint64_t> &cum, uint64_t lo, uint64_t hi, bool klist, std::vector<uint64_t> &chunk) {
        const uint64_t n = hi - lo;
        std::vector<uint64_t> sub(n + 1);
        for (uint64_t i = 0; i <= n; ++i) sub[i] = cum[lo + i] - cum[lo];
        const uint64_t res = std::max<uint64_t>(1, std::min<uint64_t>(n, klist ? p->residentk : p->resident1));
        if (klist) weighted_chunks(sub, std::max<uint64_t>(1, std::min<uint64_t>(n, 4 * res)), chunk);
        else {
            const uint64_t g = (n + res * (SELL_TILES_PER_RANGE / 2)) / (res * SELL_TILES_PER_RANGE);
            const uint64_t grid = std::min<uint64_t>(n, res * std::min<uint64_t>(16, std::max<uint64_t>(1, g)));
            if (split) weighted_chunks(sub, grid, chunk); else weighted_chunks_tapered(sub, grid, res, chunk);
        }
        for (auto &c : chunk) c += lo;
    };
    auto first_entry = [&](const std::vector<uint32_t> &list, size_t L, uint64_t tile) -> uint64_t { // first list entry whose tile is >= tile
        if (list.empty()) return std::min<uint64_t>(tile, L);
        return (uint64_t)(std::lower_bound(list.begin(), list.end(), (uint32_t)std::min<uint64_t>(tile, 0xffffffffull)) - list.begin());
    };
    std::vector<double> cost(nt);
    for (uint64_t t = 0; t < nt; ++t) cost[t] = (double)(p->h_shard_cum[t + 1] - p->h_shard_cum[t]);
    std::vector<uint64_t> tb(parts + 1, 0), best_tb;
    double best_ratio = 1e300;
    const void *rp = p->d_row_ptr;
    const uint32_t *ci = p->d_col, *kk = p->d_k;
    const uint8_t *ss = p->d_sell;
    const double *mup = d_mu;
    int32_t *cnt = d_cnt;
    SampleArgs a;
    a.seed = 0x5eed; a.row_id_base = p->row_id_base; a.n = p->n; a.chain = 1u << 30; a.iter = 0;
    a.cnt_rep_stride = p->n; a.cnt_rep_mask = CNT_REPLICAS - 1u;     // (shards have many ranges per band: replicated count vectors)
    auto launch_tiles = [&](const uint64_t *hdr1, unsigned g1, const uint64_t *hdrk, unsigned gk) -> hipError_t {
        hipError_t e = hipSuccess;
        if (g1) {
            const SellTile *ts = d_t1;
            void *kargs[] = {(void *)&rp, (void *)&ci, (void *)&kk, (void *)&ts, (void *)&hdr1, (void *)&mup, (void *)&ss, (void *)&cnt, (void *)&a};
            e = hipLaunchKernel(k1_sell_kernel(p->idx64, false, p->k1_fixed_walk), dim3(g1), dim3(64), kargs, 0, 0);
        }
        if (e == hipSuccess && gk) {
            const SellTile *ts = p->d_sell_tiles_k;
            void *kargs[] = {(void *)&rp, (void *)&ci, (void *)&kk, (void *)&ts, (void *)&hdrk, (void *)&mup, (void *)&ss, (void *)&cnt, (void *)&a};
            e = hipLaunchKernel(k1_sell_kernel(p->idx64, true), dim3(gk), dim3(64), kargs, 0, 0);
        }
        return e;
    };
    // the part's rows on the conditional-binomial chain: its interval of the list (k_sample_bigk)
    auto launch_bigk = [&](uint64_t r_lo, uint64_t r_hi) -> hipError_t {
        if (!p->n_bigk) return hipSuccess;
        const uint64_t lo = (uint64_t)(std::lower_bound(p->h_bigk_list.begin(), p->h_bigk_list.end(), r_lo) - p->h_bigk_list.begin());
        const uint64_t hi = (uint64_t)(std::lower_bound(p->h_bigk_list.begin(), p->h_bigk_list.end(), r_hi) - p->h_bigk_list.begin());
        if (hi <= lo) return hipSuccess;
        const uint64_t *list = p->d_bigk_list + lo;
        uint64_t n_list = hi - lo;
        uint32_t per = bigk_piece(n_list, p->cu_count);
        void *kargs[] = {(void *)&rp, (void *)&ci, (void *)&kk, (void *)&list, (void *)&n_list, (void *)&per, (void *)&mup, (void *)&cnt, (void *)&a};
        return hipLaunchKernel(k1_bigk_kernel(p->idx64), dim3((unsigned)((n_list + per - 1) / per)), dim3(64), kargs, 0, 0);
    };
    auto row_of_tile = [&](uint64_t t) { return t < nt ? p->h_tile_row[t] : p->m; };
    for (int round = 0; round < 6; ++round) {
        // cut the tiles by the current cost
        double total = 0.0;
        for (double c : cost) total += c;
        {
            uint64_t t = 0;
            double run = 0.0;
            for (int i = 1; i < parts; ++i) {
                const double target = total * (double)i / (double)parts;
                while (t < nt && run + cost[t] <= target) run += cost[t++];
                tb[i] = std::max<uint64_t>(t, tb[i - 1]);
            }
            tb[parts] = nt;
        }
        // the launches of every part
        for (uint64_t *h : d_hdr) if (h) (void)hipFree(h);
        d_hdr.assign((size_t)parts * 2, nullptr);
        std::vector<unsigned> g1(parts, 0), gk(parts, 0);
        for (int i = 0; i < parts; ++i) {
            std::vector<uint64_t> chunk;
            const uint64_t lo1 = first_entry(p->h_list1_tile, L1, tb[i]), hi1 = first_entry(p->h_list1_tile, L1, tb[i + 1]);
            if (hi1 > lo1) { ranges_of(p->h_cum1, lo1, hi1, false, chunk); T_TRY(upload_ranges(chunk, t1, &d_hdr[2 * i])); g1[i] = (unsigned)(chunk.size() - 1); }
            if (split) {
                const uint64_t lok = first_entry(p->h_listk_tile, Lk, tb[i]), hik = first_entry(p->h_listk_tile, Lk, tb[i + 1]);
                if (hik > lok) { ranges_of(p->h_cumk, lok, hik, true, chunk); T_TRY(upload_ranges(chunk, tk, &d_hdr[2 * i + 1])); gk[i] = (unsigned)(chunk.size() - 1); }
            }
        }
        auto launch_part = [&](int i) -> hipError_t {
            const hipError_t e = launch_tiles(d_hdr[2 * i], g1[i], d_hdr[2 * i + 1], gk[i]);
            return e == hipSuccess ? launch_bigk(row_of_tile(tb[i]), row_of_tile(tb[i + 1])) : e;
        };
        if (round == 0) { // clocks up: the whole problem for about 50 ms
            for (int w = 0; w < 4; ++w) for (int i = 0; i < parts; ++i) { a.iter++; T_TRY(launch_part(i)); }
            T_TRY(hipDeviceSynchronize());
            hipEvent_t &e0 = ev[0], &e1 = ev[1];
            T_TRY(hipEventRecord(e0, 0));
            for (int i = 0; i < parts; ++i) { a.iter++; T_TRY(launch_part(i)); }
            T_TRY(hipEventRecord(e1, 0));
            T_TRY(hipEventSynchronize(e1));
            float ms = 0.f;
            T_TRY(hipEventElapsedTime(&ms, e0, e1));
            const int more = ms > 0.f ? std::min(2000, (int)(50.0f / ms)) : 0;
            for (int w = 0; w < more; ++w) for (int i = 0; i < parts; ++i) { a.iter++; T_TRY(launch_part(i)); }
        }
        for (int r = 0; r < reps; ++r)
            for (int i = 0; i < parts; ++i) {
                a.iter++;
                T_TRY(hipEventRecord(ev[((size_t)r * parts + i) * 2], 0));
                T_TRY(launch_part(i));
                T_TRY(hipEventRecord(ev[((size_t)r * parts + i) * 2 + 1], 0));
            }
        T_TRY(hipDeviceSynchronize());
        std::vector<double> T(parts);
        double mean = 0.0, worst = 0.0;
        for (int i = 0; i < parts; ++i) {
            std::vector<float> v(reps);
            for (int r = 0; r < reps; ++r) T_TRY(hipEventElapsedTime(&v[r], ev[((size_t)r * parts + i) * 2], ev[((size_t)r * parts + i) * 2 + 1]));
            std::sort(v.begin(), v.end());
            T[i] = v[reps / 2];
            mean += T[i] / parts;
            worst = std::max(worst, T[i]);
        }
        const double ratio = mean > 0.0 ? worst / mean : 1.0;
        if (ratio < best_ratio) { best_ratio = ratio; best_tb = tb; }
        if (ratio <= 1.02 || !(mean > 0.0)) break;
        // tiles of a part that ran long get dearer: the part's cost becomes proportional to its time
        for (int i = 0; i < parts; ++i) {
            double sum = 0.0;
            for (uint64_t t = tb[i]; t < tb[i + 1]; ++t) sum += cost[t];
            if (!(sum > 0.0)) continue;
            const double f = T[i] * total / (mean * parts) / sum;
            for (uint64_t t = tb[i]; t < tb[i + 1]; ++t) cost[t] *= f;
        }
    }
    cleanup();
#undef T_TRY
    bounds[0] = 0;
    for (int i = 1; i < parts; ++i) {
        uint64_t r = best_tb[i] < nt ? p->h_tile_row[best_tb[i]] : p->m;
        r = std::min<uint64_t>(p->m, (r + ((p->row_id_base + r) & 1)));
        bounds[i] = std::max<uint64_t>(r, bounds[i - 1]);
    }
    bounds[parts] = p->m;
    return MMG_OK;
}

extern "C" int mmg_problem_create_synthetic(const mmg_synth_desc *d, int device, mmg_problem **out)
{
    if (!d || !out) return fail(MMG_ERR_ARG, "NULL argument");
    if (d->n == 0 || d->rows == 0 || !(d->avg_hits >= 1.0) || !(d->far_fraction >= 0.0 && d->far_fraction <= 1.0)) return fail(MMG_ERR_ARG, "bad synthetic spec");
    if ((d->gene_size && d->uniform) || (d->far_family && !d->gene_size) || d->far_family == 1) return fail(MMG_ERR_ARG, "bad synthetic spec: gene_size excludes uniform, far_family (0 or >= 2) needs gene_size");
    if (d->sorted && d->rows >= 0xffffffffull) return fail(MMG_ERR_ARG, "the canonical layout needs fewer than 2^32 rows per device");
    int rc = require_device(device);
    if (rc) return rc;
    std::vector<double> efflen, cdf, len_cdf;
    host_synth_tables(d->seed, d->n, d->avg_hits - 1.0, efflen, cdf, len_cdf);
    if (!(cdf[d->n - 1] > 0.0)) return fail(MMG_ERR_ARG, "synthetic abundance table is all zero");
    mmg_problem *p = new mmg_problem();
    p->device = device;
    p->m = d->rows; p->n = d->n; p->row_id_base = d->row0; p->total_k = d->rows; p->total_k_hit = d->rows; // (a generated row has at least one hit)
    p->layout = d->sorted ? (int)MMG_LAYOUT_CANONICAL : (int)MMG_LAYOUT_KEEP_ROWS;
    const double N = (double)(d->mapped_reads ? d->mapped_reads : d->rows);
    p->h_l.resize(d->n);
    for (uint32_t t = 0; t < d->n; ++t) p->h_l[t] = efflen[t] * N / 1000000000.0; // src/mmseq.cpp:603
    double *d_cdf = nullptr, *d_len_cdf = nullptr;
    uint32_t *d_lens = nullptr;
    uint64_t *d_rp64 = nullptr;
    auto cleanup = [&]() { for (void *x : {(void *)d_cdf, (void *)d_len_cdf, (void *)d_lens, (void *)d_rp64}) if (x) (void)hipFree(x); };
    auto bail = [&](int code) { cleanup(); problem_free(p); return code; };
#define SYN_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return bail(fail(MMG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e))); } while (0)
    SYN_TRY(hipMalloc((void **)&d_cdf, d->n * sizeof(double)));
    SYN_TRY(hipMalloc((void **)&d_len_cdf, 99 * sizeof(double)));
    SYN_TRY(hipMalloc((void **)&d_lens, d->rows * sizeof(uint32_t)));
    SYN_TRY(hipMalloc((void **)&d_rp64, (d->rows + 1) * sizeof(uint64_t)));
    SYN_TRY(hipMemcpy(d_cdf, cdf.data(), d->n * sizeof(double), hipMemcpyHostToDevice));
    SYN_TRY(hipMemcpy(d_len_cdf, len_cdf.data(), 99 * sizeof(double), hipMemcpyHostToDevice));
    SynthArgs sa{d->seed, d->row0, d->rows, d->n, d->uniform ? 1 : 0, d_cdf, d_len_cdf, 0, 0, 0, 1, 1};
    if (d->gene_size) {
        sa.gene_size = d->gene_size; sa.far_family = d->far_family;
        sa.n_genes = (d->n + d->gene_size - 1) / d->gene_size;
        synth_family_params(d->seed, sa.n_genes, &sa.fam_a, &sa.fam_ainv);
    }
    launch_synth_len(sa, d->far_fraction, d_lens, 0);
    SYN_TRY(hipGetLastError());
    SYN_TRY(layout_scan_lens(d->rows, d_lens, d_rp64, 0));
    SYN_TRY(hipMemcpy(&p->nnz, d_rp64 + d->rows, sizeof(uint64_t), hipMemcpyDeviceToHost));
    (void)hipFree(d_lens); d_lens = nullptr;
    const size_t col_bytes = (p->nnz + 16) * sizeof(uint32_t);
    SYN_TRY(hipMalloc((void **)&p->d_col, col_bytes));
    SYN_TRY(hipMemset(p->d_col, 0, col_bytes));
    p->device_bytes += col_bytes;
    SYN_TRY(hipMalloc((void **)&p->d_l, d->n * sizeof(double)));
    SYN_TRY(hipMemcpy(p->d_l, p->h_l.data(), d->n * sizeof(double), hipMemcpyHostToDevice));
    p->device_bytes += d->n * 8;
    launch_synth_fill(sa, d->far_fraction, d_rp64, p->d_col, 0);
    SYN_TRY(hipGetLastError());
    SYN_TRY(hipDeviceSynchronize());
#undef SYN_TRY
    uint64_t *rp = d_rp64;
    d_rp64 = nullptr; // consumed by problem_build
    rc = problem_build(p, rp);
    if (rc) return bail(rc);
    cleanup();
    *out = p;
    return MMG_OK;
}

extern "C" int mmg_problem_info_get(const mmg_problem *p, mmg_problem_info *info)
{
    if (!p || !info) return fail(MMG_ERR_ARG, "NULL argument");
    info->m = p->m; info->nnz = p->nnz; info->total_k = p->total_k; info->row_id_base = p->row_id_base;
    info->n = p->n; info->max_row_len = p->max_row_len;
    info->n_tiles = p->use_sell ? p->n_sell_tiles : p->n_tiles;
    info->device_bytes = p->device_bytes; info->index_bits = p->idx64 ? 64 : 32;
    info->sample_kernel = p->use_sell ? 2 : 0;
    info->stream_bytes = p->use_sell ? p->sell_bytes : 0;
    info->fast_tiles = p->n_fast_tiles;
    info->far_tiles = p->n_far_tiles;
    info->padded_slots = p->padded_slots;
    info->layout = p->layout;
    info->tx_renumbered = (p->renumbered() ? (p->order_derived ? 2 : (p->groups_reordered ? 3 : 1)) : 0) | (p->order_skipped ? MMG_ORDER_SKIPPED : 0);
    info->sample_grid = p->use_sell ? p->grid_sell : p->grid_sample;
    info->cu_count = p->cu_count;
    return MMG_OK;
}

extern "C" int mmg_problem_download(const mmg_problem *p, uint64_t *row_ptr, uint32_t *col_idx, uint32_t *k)
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
    if (col_idx && p->nnz) {
        HIP_TRY(hipMemcpy(col_idx, p->d_col, p->nnz * sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (p->renumbered()) {
            const std::vector<uint32_t> &map = p->h_ext_of_int;
            parallel_slices(p->nnz, [&](uint64_t a, uint64_t b) { for (uint64_t j = a; j < b; ++j) col_idx[j] = map[col_idx[j]]; });
        }
    }
    if (k && p->m) {
        if (p->d_k) HIP_TRY(hipMemcpy(k, p->d_k, p->m * sizeof(uint32_t), hipMemcpyDeviceToHost));
        else for (uint64_t i = 0; i < p->m; ++i) k[i] = 1;
    }
    return MMG_OK;
}

extern "C" int mmg_problem_get_l(const mmg_problem *p, double *l)
{
    if (!p || !l) return fail(MMG_ERR_ARG, "NULL argument");
    std::memcpy(l, p->h_l.data(), p->n * sizeof(double));
    return MMG_OK;
}

extern "C" int mmg_problem_tx_perm(const mmg_problem *p, uint32_t *int_of_ext)
{
    if (!p || !int_of_ext) return fail(MMG_ERR_ARG, "NULL argument");
    for (uint32_t t = 0; t < p->n; ++t) int_of_ext[t] = p->renumbered() ? p->h_int_of_ext[t] : t;
    return MMG_OK;
}

extern "C" int mmg_problem_start_values(const mmg_problem *p, double *mu0, int32_t *unique_hits)
{
    if (!p) return fail(MMG_ERR_ARG, "NULL problem");
    HIP_TRY(hipSetDevice(p->device));
    uint64_t *d_acc = nullptr;
    int32_t *d_uh = nullptr;
    const size_t n = p->n;
    HIP_TRY(hipMalloc((void **)&d_acc, 3 * n * sizeof(uint64_t)));
    if (hipMalloc((void **)&d_uh, n * sizeof(int32_t)) != hipSuccess) { (void)hipFree(d_acc); return fail(MMG_ERR_HIP, "hipMalloc"); }
    int rc = MMG_OK;
    do {
        if (hipMemset(d_acc, 0, 3 * n * sizeof(uint64_t)) != hipSuccess || hipMemset(d_uh, 0, n * sizeof(int32_t)) != hipSuccess) { rc = fail(MMG_ERR_HIP, "hipMemset"); break; }
        launch_start_values(p->idx64, p->d_row_ptr, p->d_col, p->d_k, p->m, p->n, d_acc, d_uh, 0);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { rc = fail(MMG_ERR_HIP, "start_values launch"); break; }
        if (mu0) {
            std::vector<uint64_t> acc(3 * n);
            if (hipMemcpy(acc.data(), d_acc, 3 * n * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(MMG_ERR_HIP, "hipMemcpy mu0"); break; }
            for (uint32_t t = 0; t < p->n; ++t) {
                const uint32_t i = p->renumbered() ? p->h_int_of_ext[t] : t;
                mu0[t] = host_start_value(acc[i], acc[n + i], acc[2 * n + i], p->h_l[t]);
            }
        }
        if (unique_hits) rc = download_ext(p, d_uh, unique_hits);
    } while (0);
    (void)hipFree(d_acc);
    (void)hipFree(d_uh);
    return rc;
}

extern "C" void mmg_problem_destroy(mmg_problem *p) { problem_free(p); }

// ------------------------------------------------------------------------------ host-side keyed draws, self tests
extern "C" int mmg_host_gamma_trace(uint64_t seed, uint64_t id, double shape, double scale, int n, double *out)
{
    if (n < 0 || !out || !(shape > 0.0)) return fail(MMG_ERR_ARG, "bad argument");
    host_gamma(seed, 5 /* TAG_SIMU */, id, 1, shape, scale, n, out);
    return MMG_OK;
}

extern "C" int mmg_selftest_kernel_info(int device, int *vgprs, int *lds_bytes, int *scratch_bytes, int *resident_per_cu)
{
    int rc = require_device(device);
    if (rc) return rc;
    hipFuncAttributes fa;
    HIP_TRY(hipFuncGetAttributes(&fa, k1_sell_kernel(false, false)));
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k1_sell_kernel(false, false), 64, 0));
    if (vgprs) *vgprs = fa.numRegs;
    if (lds_bytes) *lds_bytes = (int)fa.sharedSizeBytes;
    if (scratch_bytes) *scratch_bytes = (int)fa.localSizeBytes;
    if (resident_per_cu) *resident_per_cu = per_cu;
    return MMG_OK;
}

extern "C" int mmg_selftest_math(int device, int64_t n, const double *x, double *ol, double *oe, double *os, double *orc)
{
    if (n < 0 || !x || !ol || !oe || !os || !orc) return fail(MMG_ERR_ARG, "bad argument");
    if (device < 0) { host_math(n, x, ol, oe, os, orc); return MMG_OK; }
    int rc = require_device(device);
    if (rc) return rc;
    double *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 5 * (size_t)n * sizeof(double) + 8));
    hipError_t e = hipMemcpy(d, x, n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) { launch_selftest_math(n, d, d + n, d + 2 * n, d + 3 * n, d + 4 * n, 0); e = hipDeviceSynchronize(); }
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
    if (device < 0) { host_philox(ctr, key, out); return MMG_OK; }
    int rc = require_device(device);
    if (rc) return rc;
    uint32_t *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 12 * sizeof(uint32_t)));
    hipError_t e = hipMemcpy(d, ctr, 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + 4, key, 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) { launch_selftest_philox(d, d + 4, d + 6, 0); e = hipDeviceSynchronize(); }
    if (e == hipSuccess) e = hipMemcpy(out, d + 6, 24, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(MMG_ERR_HIP, std::string("selftest_philox: ") + hipGetErrorString(e));
    return MMG_OK;
}

extern "C" int mmg_selftest_gamma(int device, uint64_t seed, double shape, double scale, int64_t n, double *out)
{
    if (n < 0 || !out || !(shape > 0.0)) return fail(MMG_ERR_ARG, "bad argument");
    if (device < 0) { host_gamma(seed, 2 /* TAG_GAMMA */, 0, 0, shape, scale, n, out); return MMG_OK; }
    int rc = require_device(device);
    if (rc) return rc;
    double *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, (size_t)n * sizeof(double) + 8));
    launch_selftest_gamma(seed, shape, scale, n, d, 0);
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, d, n * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(MMG_ERR_HIP, std::string("selftest_gamma: ") + hipGetErrorString(e));
    return MMG_OK;
}

extern "C" int mmg_selftest_binomial(int device, uint64_t seed, uint32_t nn, double p, int64_t n, uint32_t *out)
{
    if (n < 0 || !out) return fail(MMG_ERR_ARG, "bad argument");
    if (device < 0) { host_binomial(seed, nn, p, n, out); return MMG_OK; }
    int rc = require_device(device);
    if (rc) return rc;
    uint32_t *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, (size_t)n * sizeof(uint32_t) + 8));
    launch_selftest_binomial(seed, nn, p, n, d, 0);
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, d, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(MMG_ERR_HIP, std::string("selftest_binomial: ") + hipGetErrorString(e));
    return MMG_OK;
}

extern "C" int mmg_selftest_btrs_pretest(int device, uint64_t seed, int64_t n_cases, double n_lo, double n_hi, uint64_t *counts)
{
    if (n_cases < 0 || !counts || !(n_lo >= 21.0) || !(n_hi >= n_lo) || !(n_hi <= 4294967295.0)) return fail(MMG_ERR_ARG, "bad argument");
    int rc = require_device(device);
    if (rc) return rc;
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 5 * sizeof(unsigned long long)));
    hipError_t e = hipMemset(d, 0, 5 * sizeof(unsigned long long));
    if (e == hipSuccess) { launch_selftest_btrs_pretest(seed, n_cases, n_lo, n_hi, d, 0); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(counts, d, 5 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(MMG_ERR_HIP, std::string("selftest_btrs_pretest: ") + hipGetErrorString(e));
    return MMG_OK;
}

extern "C" int mmg_selftest_binv_pretest(int device, uint64_t seed, int64_t n_cases, double n_lo, double n_hi, double slack, uint64_t *counts)
{
    if (n_cases < 0 || !counts || !(n_lo >= 1.0) || !(n_hi >= n_lo) || !(n_hi <= 4294967295.0) || !(slack > 0.0)) return fail(MMG_ERR_ARG, "bad argument");
    int rc = require_device(device);
    if (rc) return rc;
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 5 * sizeof(unsigned long long)));
    hipError_t e = hipMemset(d, 0, 5 * sizeof(unsigned long long));
    if (e == hipSuccess) { launch_selftest_binv_pretest(seed, n_cases, n_lo, n_hi, (float)slack, d, 0); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(counts, d, 5 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(MMG_ERR_HIP, std::string("selftest_binv_pretest: ") + hipGetErrorString(e));
    return MMG_OK;
}
