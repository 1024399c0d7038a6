// gibbs_kernels.h -- the CDNA4 (gfx950) kernels of the Gibbs hot path.
//
//   k_sample  (K1)  src/mmseq.cpp:857-891 + :887  per-row multinomial allocation of the row's
//                   k reads over its hit set, weights mu[t], scattered into the count vector
//   k_update  (K2)  src/mmseq.cpp:896-917         Gamma redraw of mu, trace capture, moments
//
// K1 is an HBM-bound stream of the CSR (u32 row_ptr + u32 col_idx, 4 B per hit) with an
// L2-resident gather of fp64 mu and an L2 int32 atomic scatter.  It is a CSR-stream kernel:
// a workgroup owns a TILE of consecutive rows (<= TILE_NNZ hits, precomputed on the host),
// streams the tile's column indices with 16-byte coalesced non-temporal loads, gathers the
// weights and parks (col, weight) in LDS; then one lane per row walks its LDS segment
// sequentially (total, one Philox uniform, prefix walk) and issues one atomic.  The
// sequential fp64 walk is what makes the draw bit-reproducible against the CPU oracle.
#pragma once
#include "../../include/mmgibbs.h"
#include "mmg_math.h"

namespace mmg {

constexpr int K1_BLOCK = 256;
constexpr uint32_t K1_WIN_MARGIN = MMG_ROW_SPAN_HINT;       // hits are expected within this many ids above a row's first hit
constexpr uint32_t K_SMALL = 8u;              // == MMG_K_SMALL
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// K1 variants (compile-time): LDS staging elements (tile = ELEMS-8 hits), window width, unroll
enum : int { K1M_NO_PHASE2 = 1 };

// One tile of consecutive rows, built once per problem (k_tile_desc)
struct TileDesc {
    uint64_t nz0;   // first hit of the tile in col_idx
    uint64_t r0;    // first row
    uint32_t nrows;
    uint32_t nnz;   // > tile capacity <=> a single long row handled by the slow path
    uint32_t cmin;  // leading transcript of the first non-empty row
    uint32_t clast; // leading transcript of the last non-empty row
    uint32_t cmax;  // largest transcript id in the tile
    uint32_t call;  // smallest transcript id in the tile (== cmin when rows are sorted)
    uint32_t nnz4;  // hits when every row is padded to a multiple of 4 (size of the tile in the 16-bit stream)
    uint32_t pad_;
};

struct SampleArgs {
    uint64_t seed;
    uint64_t row_id_base;
    uint32_t n;
    uint32_t chain;
    uint32_t iter;
};

__device__ __forceinline__ void global_count_add(int32_t *cnt, uint32_t col, int32_t v)
{
    __hip_atomic_fetch_add(&cnt[col], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- row views: how phase 2 reads one row's hits and weights -------------------------------
// total() / pick() implement the k == 1 fast path; every variant performs the SAME sequence of
// fp64 additions in row order (adding an exact 0.0 for masked slots), so all are bit-identical.
template <int WIN, int UNR, bool INWIN>
struct RowViewWin {
    const uint32_t *cl; // row's column ids in LDS
    uint32_t L;
    uint32_t wbase;
    const double *s_mu;
    const double *gmu;
    __device__ __forceinline__ uint32_t col(uint32_t j) const { return cl[j]; }
    __device__ __forceinline__ double wc(uint32_t c) const
    {
        const uint32_t d = c - wbase;
        if (INWIN) return s_mu[d];
        return d < (uint32_t)WIN ? s_mu[d] : gmu[c];
    }
    __device__ __forceinline__ double w(uint32_t j) const { return wc(cl[j]); }
    __device__ __forceinline__ double total() const
    {
        double t = 0.0;
        if (UNR <= 1) {
            for (uint32_t j = 0; j < L; ++j) t += w(j);
            return t;
        }
        const uint32_t last = L - 1;
        for (uint32_t j = 0; j < L; j += UNR) {
            uint32_t c[UNR];
            double wv[UNR];
#pragma unroll
            for (int i = 0; i < UNR; ++i) c[i] = cl[min(j + (uint32_t)i, last)];
#pragma unroll
            for (int i = 0; i < UNR; ++i) wv[i] = wc(c[i]);
#pragma unroll
            for (int i = 0; i < UNR; ++i) t += (j + (uint32_t)i < L) ? wv[i] : 0.0;
        }
        return t;
    }
    __device__ __forceinline__ uint32_t pick(double target) const
    {
        double acc = 0.0;
        if (UNR <= 1) {
            for (uint32_t j = 0; j < L; ++j) {
                acc += w(j);
                if (target < acc) return j;
            }
            return L - 1;
        }
        const uint32_t last = L - 1;
        for (uint32_t j = 0; j < L; j += UNR) {
            uint32_t c[UNR];
            double wv[UNR];
#pragma unroll
            for (int i = 0; i < UNR; ++i) c[i] = cl[min(j + (uint32_t)i, last)];
#pragma unroll
            for (int i = 0; i < UNR; ++i) wv[i] = wc(c[i]);
            uint32_t found = 0xffffffffu;
#pragma unroll
            for (int i = 0; i < UNR; ++i) {
                acc += (j + (uint32_t)i < L) ? wv[i] : 0.0;
                if (found == 0xffffffffu && target < acc) found = min(j + (uint32_t)i, last);
            }
            if (found != 0xffffffffu) return found;
        }
        return last;
    }
};

// fast path: the tile lies inside the LDS window and s_col holds BYTE OFFSETS into s_mu
// ((col - wbase) * 8, written by commit), so a hit costs ds_read_b32 + ds_read_b64 + v_add_f64.
// Full groups of UNR hits run unmasked with the NEXT group's offsets already in flight (one exposed
// LDS latency per group); the <UNR tail is one group whose missing slots point at a 0.0 entry
// (s_mu[WIN]), so it needs no masking of the fp64 adds and no loop.  The additions happen in row
// order in every case (x + 0.0 == x exactly), so the result equals the plain sequential walk.
template <int UNR, uint32_t ZERO_OFF>
struct RowViewOff {
    const uint32_t *cl; // row's byte offsets in LDS (reading a few elements past the row is harmless)
    uint32_t L;
    const double *s_mu;
    __device__ __forceinline__ double wo(uint32_t off) const { return *(const double *)((const char *)s_mu + off); }
    __device__ __forceinline__ uint32_t col(uint32_t j) const { return cl[j]; } // a byte offset: add() understands it
    __device__ __forceinline__ double w(uint32_t j) const { return wo(cl[j]); }
    __device__ __forceinline__ double total() const
    {
        double t = 0.0;
        uint32_t j = 0;
        uint32_t o[UNR];
#pragma unroll
        for (int i = 0; i < UNR; ++i) o[i] = cl[i];
        for (; j + UNR <= L; j += UNR) {
            double wv[UNR];
#pragma unroll
            for (int i = 0; i < UNR; ++i) wv[i] = wo(o[i]);
#pragma unroll
            for (int i = 0; i < UNR; ++i) o[i] = cl[j + UNR + i]; // next group (or the tail), in flight during the adds
#pragma unroll
            for (int i = 0; i < UNR; ++i) t += wv[i];
        }
        const uint32_t rem = L - j; // 0 .. UNR-1
        double wv[UNR > 1 ? UNR - 1 : 1];
#pragma unroll
        for (int i = 0; i < UNR - 1; ++i) wv[i] = wo((uint32_t)i < rem ? o[i] : ZERO_OFF);
#pragma unroll
        for (int i = 0; i < UNR - 1; ++i) t += wv[i];
        return t;
    }
    __device__ __forceinline__ uint32_t pick(double target) const
    {
        double acc = 0.0;
        uint32_t j = 0;
        uint32_t o[UNR];
#pragma unroll
        for (int i = 0; i < UNR; ++i) o[i] = cl[i];
        for (; j + UNR <= L; j += UNR) {
            double wv[UNR], pa[UNR];
#pragma unroll
            for (int i = 0; i < UNR; ++i) wv[i] = wo(o[i]);
#pragma unroll
            for (int i = 0; i < UNR; ++i) o[i] = cl[j + UNR + i];
#pragma unroll
            for (int i = 0; i < UNR; ++i) { acc += wv[i]; pa[i] = acc; }
            if (target < acc) { // prefix sums never decrease: the first i with target < pa[i] is in this group
                uint32_t sel = UNR - 1;
#pragma unroll
                for (int i = UNR - 2; i >= 0; --i) sel = (target < pa[i]) ? (uint32_t)i : sel;
                return j + sel;
            }
        }
        const uint32_t rem = L - j;
        double wv[UNR > 1 ? UNR - 1 : 1];
#pragma unroll
        for (int i = 0; i < UNR - 1; ++i) wv[i] = wo((uint32_t)i < rem ? o[i] : ZERO_OFF);
        uint32_t sel = L - 1; // also the fallback when rounding leaves target >= total
#pragma unroll
        for (int i = UNR - 2; i >= 0; --i) {
            double p = acc;
#pragma unroll
            for (int q = 0; q <= i; ++q) p += wv[q];
            sel = ((uint32_t)i < rem && target < p) ? j + (uint32_t)i : sel;
        }
        return sel;
    }
};

// long rows: everything from global memory
struct RowViewGlobal {
    const uint32_t *cl;
    const double *gmu;
    uint32_t L;
    __device__ __forceinline__ uint32_t col(uint32_t j) const { return cl[j]; }
    __device__ __forceinline__ double w(uint32_t j) const { return gmu[cl[j]]; }
    __device__ __forceinline__ double total() const
    {
        double t = 0.0;
        for (uint32_t j = 0; j < L; ++j) t += w(j);
        return t;
    }
    __device__ __forceinline__ uint32_t pick(double target) const
    {
        double acc = 0.0;
        for (uint32_t j = 0; j < L; ++j) {
            acc += w(j);
            if (target < acc) return j;
        }
        return L - 1;
    }
};

// One row: restates src/mmseq.cpp:871-889.  add(col, v) adds v to the count of transcript col.
template <bool HAS_K, typename View, typename Add>
__device__ __forceinline__ void allocate_row(const View &v, Add add, uint32_t kk, const SampleArgs &a, uint64_t row_id)
{
    const uint32_t L = v.L;
    if (L == 0 || kk == 0) return;
    if (L == 1) { add(v.col(0), (int32_t)kk); return; }
    const double total = v.total();
    const bool degenerate = !(total > 0.0) || !(total < __builtin_huge_val());
    if (!HAS_K || kk <= K_SMALL) {
        Stream2 s(a.seed, a.chain, TAG_ROW, row_id, a.iter);
        for (uint32_t d = 0; d < kk; ++d) {
            const double u = s.next();
            uint32_t sel;
            if (degenerate) {
                sel = (uint32_t)(u * (double)L);
                if (sel >= L) sel = L - 1;
            } else {
                sel = v.pick(u * total);
            }
            add(v.col(sel), 1);
        }
        return;
    }
    // conditional-binomial chain (the published gsl_ran_multinomial scheme, src/mmseq.cpp:880)
    Stream2 q(a.seed, a.chain, TAG_ROW, row_id, a.iter);
    uint32_t remaining = kk;
    double rem_w = total;
    for (uint32_t j = 0; j + 1 < L && remaining > 0; ++j) {
        const double w = v.w(j);
        double p = degenerate ? 1.0 / (double)(L - j) : (rem_w > 0.0 ? w / rem_w : 1.0);
        if (p > 1.0) p = 1.0;
        const uint32_t x = binomial(q, remaining, p);
        if (x) add(v.col(j), (int32_t)x);
        remaining -= x;
        rem_w -= w;
    }
    if (remaining > 0) add(v.col(L - 1), (int32_t)remaining);
}

// Per-tile descriptor builder: one 64-lane workgroup per tile.
template <typename IdxT>
__global__ __launch_bounds__(64) void k_tile_desc(const IdxT *__restrict__ row_ptr, const uint32_t *__restrict__ col_idx,
                                                  const uint64_t *__restrict__ tile_row, uint64_t n_tiles, TileDesc *out)
{
    const uint64_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const uint64_t r0 = tile_row[tile], r1 = tile_row[tile + 1];
    const uint64_t nz0 = row_ptr[r0], nz1 = row_ptr[r1];
    uint32_t mx = 0, mn = 0xffffffffu, n4 = 0;
    for (uint64_t j = nz0 + threadIdx.x; j < nz1; j += 64) { const uint32_t c = col_idx[j]; mx = max(mx, c); mn = min(mn, c); }
    for (uint64_t r = r0 + threadIdx.x; r < r1; r += 64) n4 += ((uint32_t)((uint64_t)row_ptr[r + 1] - (uint64_t)row_ptr[r]) + 3u) & ~3u;
    for (int off = 32; off > 0; off >>= 1) {
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
        mn = min(mn, (uint32_t)__shfl_xor((int)mn, off));
        n4 += (uint32_t)__shfl_xor((int)n4, off);
    }
    if (threadIdx.x == 0) {
        TileDesc d;
        d.nz0 = nz0; d.r0 = r0; d.nrows = (uint32_t)(r1 - r0);
        d.nnz = (uint32_t)min((uint64_t)0xffffffffu, nz1 - nz0);
        d.cmin = 0; d.clast = 0; d.cmax = mx; d.call = mn; d.nnz4 = n4; d.pad_ = 0;
        if (nz1 > nz0) {
            d.cmin = col_idx[nz0];
            uint64_t rl = r1 - 1;
            while (rl > r0 && (uint64_t)row_ptr[rl] == nz1) --rl; // skip trailing empty rows
            d.clast = col_idx[row_ptr[rl]];
        }
        out[tile] = d;
    }
}

// K1.  A workgroup walks ONE contiguous range of tiles.  Rows arrive sorted by leading transcript
// (the order hit-set collapse produces), so consecutive tiles touch a slowly advancing band of
// transcripts: that band's mu and counts live in an LDS window [base, base+WIN); hits outside
// the window fall back to the L2 gather / global atomic, so any row order is CORRECT, sorted
// order is FAST (global int32 atomics cap at ~26 G/s on MI355X, LDS atomics at >100 G/s).
// The next tile's column ids and row offsets are prefetched into registers while the current
// tile's rows are walked, so the only dependent global access per tile is the descriptor.
template <typename IdxT, bool HAS_K, int ELEMS, int WIN, int UNR, int MODE, int BS, int RC>
__global__ __launch_bounds__(BS) void k_sample(const IdxT *__restrict__ row_ptr, const uint32_t *__restrict__ col_idx,
                                                     const uint32_t *__restrict__ kmult, const TileDesc *__restrict__ tiles,
                                                     const uint64_t *__restrict__ chunk_tile, const double *__restrict__ gmu,
                                                     int32_t *gcnt, SampleArgs a)
{
    constexpr int TILE_NNZ = ELEMS - 8;
    constexpr int ROWS_CAP = RC > 0 ? RC : ELEMS / 4; // rows per tile (host enforces the same cap)
    constexpr int NC = ELEMS / 4 / BS;        // 16-byte chunks per thread
    constexpr int NR = ROWS_CAP / BS + 1;     // row offsets per thread (nrows+1 entries)
    __shared__ __attribute__((aligned(16))) uint32_t s_col[ELEMS];
    __shared__ __attribute__((aligned(16))) double s_mu[WIN + 2]; // [WIN] stays 0.0: the tail group's padding slot
    __shared__ uint32_t s_rp[ROWS_CAP + BS];
    __shared__ uint32_t s_k[HAS_K ? ROWS_CAP + BS : 1];
    __shared__ int32_t s_cnt[WIN];
    const int tid = threadIdx.x;

    const uint64_t t_begin = chunk_tile[blockIdx.x], t_end = chunk_tile[blockIdx.x + 1];
    if (t_begin >= t_end) return;

    for (int i = tid; i < WIN; i += BS) s_cnt[i] = 0;
    if (tid < 2) s_mu[WIN + tid] = 0.0;
    uint32_t base = 0xffffffffu;
    bool win_valid = false;

    auto flush_window = [&]() {
        if (!win_valid) return;
        for (int i = tid; i < WIN; i += BS) {
            const int32_t v = s_cnt[i];
            if (v) { global_count_add(gcnt, base + (uint32_t)i, v); s_cnt[i] = 0; }
        }
    };

    // Prefetch registers: raw loaded values only (any arithmetic on them would force the
    // compiler to wait for the loads right here); all loads are unconditional with clamped
    // indices so that no branch separates them (keeps the vmcnt bookkeeping exact).
    u32x4 pc[NC];
    IdxT prp[NR];
    uint32_t pk[HAS_K ? NR : 1];
    auto issue = [&](const TileDesc &d) {
        if (d.nnz == 0 || d.nnz > (uint32_t)TILE_NNZ) return; // uniform
        const uint64_t abase = d.nz0 & ~(uint64_t)3;
        const uint32_t shift = (uint32_t)(d.nz0 - abase);
        const uint32_t nchunks = (d.nnz + shift + 3) >> 2;
        const u32x4 *__restrict__ src = (const u32x4 *)(col_idx + abase);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const uint32_t ch = min((uint32_t)tid + (uint32_t)i * BS, nchunks - 1);
            pc[i] = __builtin_nontemporal_load(src + ch);
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const uint32_t idx = min((uint32_t)tid + (uint32_t)i * BS, d.nrows);
            prp[i] = row_ptr[d.r0 + idx];
            if (HAS_K) pk[i] = kmult[d.r0 + min(idx, d.nrows - 1)];
        }
    };
    auto commit = [&](const TileDesc &d, bool inwin, uint32_t wbase) {
        const uint32_t shift = (uint32_t)(d.nz0 & 3);
        const uint32_t nchunks = (d.nnz + shift + 3) >> 2;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const uint32_t ch = (uint32_t)tid + (uint32_t)i * BS;
            // in-window tiles store byte offsets into s_mu instead of column ids (edge junk is never read)
            const u32x4 v = inwin ? (pc[i] - wbase) * 8u : pc[i];
            if (ch < nchunks) *(u32x4 *)(s_col + 4 * ch) = v;
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const uint32_t idx = (uint32_t)tid + (uint32_t)i * BS;
            if (idx <= d.nrows) s_rp[idx] = (uint32_t)((uint64_t)prp[i] - d.nz0) + shift;
            if (HAS_K && idx < d.nrows) s_k[idx] = pk[i];
        }
    };

    TileDesc d = tiles[t_begin];
    issue(d);
    for (uint64_t tile = t_begin; tile < t_end; ++tile) {
        TileDesc nd;
        nd.nnz = 0; nd.nrows = 0; nd.nz0 = 0; nd.r0 = 0; nd.cmin = nd.clast = nd.cmax = nd.call = 0; nd.nnz4 = 0; nd.pad_ = 0;
        if (tile + 1 < t_end) nd = tiles[tile + 1];
        if (d.nnz > (uint32_t)TILE_NNZ) {
            // a single row longer than a tile: one lane walks it straight from global memory
            if (tid == 0) {
                const uint64_t r0 = d.r0, r1 = r0 + d.nrows;
                const uint64_t nz0 = (uint64_t)row_ptr[r0], nz1 = (uint64_t)row_ptr[r1];
                RowViewGlobal v{col_idx + nz0, gmu, (uint32_t)(nz1 - nz0)};
                allocate_row<HAS_K>(v, [&](uint32_t c, int32_t x) { global_count_add(gcnt, c, x); }, HAS_K ? kmult[r0] : 1u, a,
                                    a.row_id_base + r0);
            }
            issue(nd);
            d = nd;
            continue;
        }
        if (d.nnz == 0) { issue(nd); d = nd; continue; } // only empty rows
        // ---- window decision (uniform, from the descriptor)
        const bool keep = win_valid && d.cmin >= base && (uint64_t)d.clast + K1_WIN_MARGIN <= (uint64_t)base + WIN;
        if (!keep) {
            flush_window();
            base = d.cmin & ~15u;
            win_valid = true;
            for (int i = tid; i < WIN; i += BS) {
                const uint32_t c = base + (uint32_t)i;
                s_mu[i] = c < a.n ? gmu[c] : 0.0;
            }
        }
        const uint32_t wbase = base;
        const bool inwin = d.call >= wbase && (uint64_t)d.cmax < (uint64_t)wbase + WIN;
        commit(d, inwin, wbase);
        issue(nd); // prefetch the next tile; the loads stay in flight across phase 2
        __syncthreads();
        // ---- phase 2: one lane per row walks its LDS segment
        if (!(MODE & K1M_NO_PHASE2)) {
            if (inwin) {
                auto add = [&](uint32_t off, int32_t x) { atomicAdd((int32_t *)((char *)s_cnt + (off >> 1)), x); };
                for (uint32_t r = tid; r < d.nrows; r += BS) {
                    const uint32_t b = s_rp[r], L = s_rp[r + 1] - b;
                    RowViewOff<UNR, (uint32_t)WIN * 8u> v{s_col + b, L, s_mu};
                    allocate_row<HAS_K>(v, add, HAS_K ? s_k[r] : 1u, a, a.row_id_base + d.r0 + r);
                }
            } else {
                auto add = [&](uint32_t c, int32_t x) {
                    const uint32_t dd = c - wbase;
                    if (dd < (uint32_t)WIN) atomicAdd(&s_cnt[dd], x);
                    else global_count_add(gcnt, c, x);
                };
                for (uint32_t r = tid; r < d.nrows; r += BS) {
                    const uint32_t b = s_rp[r], L = s_rp[r + 1] - b;
                    RowViewWin<WIN, 1, false> v{s_col + b, L, wbase, s_mu, gmu};
                    allocate_row<HAS_K>(v, add, HAS_K ? s_k[r] : 1u, a, a.row_id_base + d.r0 + r);
                }
            }
        }
        __syncthreads();
        d = nd;
    }
    flush_window();
}

// =============================================================================================
// K1 on the 16-bit tile stream (the default when the tiles qualify).
//
// Same workgroup shape and row walk as k_sample, but the tile arrives as ONE contiguous block of
// 16-bit words built once per problem (k_encode16): (nrows+1) row entries, then the hits as
// `(col - window_base) * 8`, i.e. ready-made byte offsets into the LDS window, rows padded to 4 hits
// with the offset of the window's 0.0 slot.  Half the HBM bytes, a one-instruction unpack instead of a
// compare/select/subtract/shift per hit, half the prefetch registers -- which pays for a depth-2
// prefetch (tiles i+1 and i+2 in flight while tile i is walked).  The window policy (base per tile,
// when it slides) is a pure function of the tile descriptors and the workgroup's tile range and is
// precomputed on the host into S16Tile.  Tiles that do not qualify are walked straight from the
// 32-bit CSR in global memory: correct for any input, fast for the sorted layout.
struct S16Tile {
    uint64_t s16;     // 16-byte-unit offset of the tile's block in the stream (fast tiles)
    uint64_t r0;      // first row
    uint32_t nrows;
    uint32_t nnz4;    // padded hits in the block
    uint32_t wbase;   // LDS window base in force while this tile is walked
    uint32_t flags;   // S16_*
};
enum : uint32_t { S16_FAST = 1, S16_SHIFT = 2, S16_EMPTY = 4 };
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));

// fast path view: rows are padded to 4 hits in LDS, every group is one aligned ds_read_b128.
// total() keeps the running sum at the first NGC group boundaries in registers, so pick() locates the
// group with compares and re-reads ONE group instead of walking the row again (the walk is bound by
// LDS throughput).  The prefix it resumes from is the very value the first pass produced, so the
// selected hit is the one the plain sequential walk selects.
struct RowView4 {
    static constexpr int NGC = 8;
    const uint32_t *cl; // 16-byte aligned; reading one group past the row is harmless
    uint32_t L4;        // padded length (multiple of 4)
    uint32_t L;         // true length
    const double *s_mu;
    mutable double P[NGC];
    __device__ __forceinline__ double wo(uint32_t off) const { return *(const double *)((const char *)s_mu + off); }
    __device__ __forceinline__ uint32_t col(uint32_t j) const { return cl[j]; } // a byte offset: add() understands it
    __device__ __forceinline__ double w(uint32_t j) const { return wo(cl[j]); }
    __device__ __forceinline__ double total() const
    {
        const u32x4 *g = (const u32x4 *)cl;
        const uint32_t ng = L4 >> 2;
        u32x4 o = g[0];
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < NGC; ++i) {
            if ((uint32_t)i < ng) {
                const double w0 = wo(o.x), w1 = wo(o.y), w2 = wo(o.z), w3 = wo(o.w);
                o = g[i + 1];
                t += w0; t += w1; t += w2; t += w3;
            }
            P[i] = t;
        }
        for (uint32_t i = NGC; i < ng; ++i) {
            const double w0 = wo(o.x), w1 = wo(o.y), w2 = wo(o.z), w3 = wo(o.w);
            o = g[i + 1];
            t += w0; t += w1; t += w2; t += w3;
        }
        return t;
    }
    __device__ __forceinline__ uint32_t in_group(const u32x4 *g, uint32_t i, double acc, double target, double &p3) const
    {
        const u32x4 o = g[i];
        const double p0 = acc + wo(o.x), p1 = p0 + wo(o.y), p2 = p1 + wo(o.z);
        p3 = p2 + wo(o.w);
        return target < p0 ? 0u : (target < p1 ? 1u : (target < p2 ? 2u : 3u));
    }
    __device__ __forceinline__ uint32_t pick(double target) const
    {
        const u32x4 *g = (const u32x4 *)cl;
        const uint32_t ng = L4 >> 2;
        // first cached boundary the target falls below (prefix sums never decrease)
        uint32_t gs = NGC;
#pragma unroll
        for (int i = NGC - 1; i >= 0; --i) gs = (target < P[i]) ? (uint32_t)i : gs;
        double p3;
        if (gs < NGC) {
            if (gs >= ng) return L - 1; // cannot happen (P[ng-1] is the total); keeps the index in range
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < NGC - 1; ++i) acc = (gs == (uint32_t)i + 1) ? P[i] : acc;
            return 4 * gs + in_group(g, gs, acc, target, p3);
        }
        double acc = P[NGC - 1];
        for (uint32_t i = NGC; i < ng; ++i) {
            const uint32_t sel = in_group(g, i, acc, target, p3);
            if (target < p3) return 4 * i + sel;
            acc = p3;
        }
        return L - 1; // rounding left target >= total: the last real hit
    }
};

// rows of a slow tile: column ids and row extents straight from the 32-bit device CSR
template <int WIN>
struct RowViewGlobalWin {
    const uint32_t *cl;
    uint32_t L;
    uint32_t wbase;
    const double *s_mu;
    const double *gmu;
    __device__ __forceinline__ uint32_t col(uint32_t j) const { return cl[j]; }
    __device__ __forceinline__ double w(uint32_t j) const
    {
        const uint32_t c = cl[j], d = c - wbase;
        return d < (uint32_t)WIN ? s_mu[d] : gmu[c];
    }
    __device__ __forceinline__ double total() const
    {
        double t = 0.0;
        for (uint32_t j = 0; j < L; ++j) t += w(j);
        return t;
    }
    __device__ __forceinline__ uint32_t pick(double target) const
    {
        double acc = 0.0;
        for (uint32_t j = 0; j < L; ++j) {
            acc += w(j);
            if (target < acc) return j;
        }
        return L - 1;
    }
};

// Stream builder: one 64-lane workgroup per fast tile.  Block = ceil((nrows+1)/8) chunks of u16 row
// entries (start | pad, in hits, relative to the tile, start a multiple of 4) followed by ceil(nnz4/8)
// chunks of u16 byte offsets (col - wbase) * 8, rows closed with ZERO_OFF = win * 8.
template <typename IdxT>
__global__ __launch_bounds__(64) void k_encode16(const IdxT *__restrict__ row_ptr, const uint32_t *__restrict__ col_idx,
                                                 const S16Tile *__restrict__ tiles, uint64_t n_tiles, uint32_t win, uint16_t *stream)
{
    __shared__ uint32_t s_start[1024 + 1];
    const uint64_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const S16Tile d = tiles[tile];
    if (!(d.flags & S16_FAST)) return;
    uint16_t *blk = stream + d.s16 * 8;
    if (threadIdx.x == 0) {
        uint32_t pos = 0;
        for (uint32_t r = 0; r < d.nrows; ++r) {
            s_start[r] = pos;
            pos += ((uint32_t)((uint64_t)row_ptr[d.r0 + r + 1] - (uint64_t)row_ptr[d.r0 + r]) + 3u) & ~3u;
        }
        s_start[d.nrows] = pos;
    }
    __syncthreads();
    const uint32_t rpe = ((d.nrows + 8) >> 3) << 3;
    for (uint32_t i = threadIdx.x; i < rpe; i += 64) {
        uint32_t e = 0;
        if (i < d.nrows) {
            const uint32_t L = (uint32_t)((uint64_t)row_ptr[d.r0 + i + 1] - (uint64_t)row_ptr[d.r0 + i]);
            e = s_start[i] | (((L + 3u) & ~3u) - L);
        } else if (i == d.nrows) {
            e = s_start[i];
        }
        blk[i] = (uint16_t)e;
    }
    uint16_t *cb = blk + rpe;
    const uint16_t zero_off = (uint16_t)(win * 8u);
    for (uint32_t r = 0; r < d.nrows; ++r) { // rows are short: one wave sweeps a row at a time
        const uint64_t b = row_ptr[d.r0 + r];
        const uint32_t L = (uint32_t)((uint64_t)row_ptr[d.r0 + r + 1] - b), L4 = (L + 3u) & ~3u;
        for (uint32_t j = threadIdx.x; j < L4; j += 64)
            cb[s_start[r] + j] = j < L ? (uint16_t)((col_idx[b + j] - d.wbase) * 8u) : zero_off;
    }
    const uint32_t ce = ((d.nnz4 + 7) >> 3) << 3;
    for (uint32_t j = d.nnz4 + threadIdx.x; j < ce; j += 64) cb[j] = zero_off;
}

// Fused walk of C chains over one row (k == 1): the row's offsets are read once per group and feed C
// independent accumulation chains (window c lives at s_mu + c*MU_STRIDE doubles, counts at
// s_cnt + c*WIN).  Every chain performs exactly the additions of the single-chain walk, in the same
// order, and draws from its own keyed stream, so chain c is bit-identical to a C = 1 run.
template <int C, int WIN, int MU_STRIDE>
__device__ __forceinline__ void walk_row_fused(const uint32_t *cl, uint32_t L4, uint32_t L, const double *s_mu, int32_t *s_cnt,
                                               const SampleArgs &a, uint64_t row_id)
{
    if (L == 0) return;
    if (L == 1) {
        const uint32_t off = cl[0];
#pragma unroll
        for (int c = 0; c < C; ++c) atomicAdd((int32_t *)((char *)(s_cnt + c * WIN) + (off >> 1)), 1);
        return;
    }
    const u32x4 *g = (const u32x4 *)cl;
    const uint32_t ng = L4 >> 2;
    double t[C];
#pragma unroll
    for (int c = 0; c < C; ++c) t[c] = 0.0;
    u32x4 o = g[0];
    for (uint32_t i = 0; i < ng; ++i) {
        double w[C][4];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const char *m = (const char *)(s_mu + c * MU_STRIDE);
            w[c][0] = *(const double *)(m + o.x); w[c][1] = *(const double *)(m + o.y);
            w[c][2] = *(const double *)(m + o.z); w[c][3] = *(const double *)(m + o.w);
        }
        o = g[i + 1];
#pragma unroll
        for (int c = 0; c < C; ++c) { t[c] += w[c][0]; t[c] += w[c][1]; t[c] += w[c][2]; t[c] += w[c][3]; }
    }
    double target[C];
    uint32_t sel[C];
    bool pending[C];
    bool any = false;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        Stream2 s(a.seed, a.chain + (uint32_t)c, TAG_ROW, row_id, a.iter);
        const double u = s.next();
        const bool degenerate = !(t[c] > 0.0) || !(t[c] < __builtin_huge_val());
        if (degenerate) {
            uint32_t j = (uint32_t)(u * (double)L);
            sel[c] = j < L ? j : L - 1;
            pending[c] = false;
        } else {
            target[c] = u * t[c];
            sel[c] = L - 1;
            pending[c] = true;
            any = true;
        }
    }
    if (any) {
        double acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.0;
        o = g[0];
        for (uint32_t i = 0; i < ng && any; ++i) {
            double w[C][4];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const char *m = (const char *)(s_mu + c * MU_STRIDE);
                w[c][0] = *(const double *)(m + o.x); w[c][1] = *(const double *)(m + o.y);
                w[c][2] = *(const double *)(m + o.z); w[c][3] = *(const double *)(m + o.w);
            }
            o = g[i + 1];
            any = false;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const double p0 = acc[c] + w[c][0], p1 = p0 + w[c][1], p2 = p1 + w[c][2], p3 = p2 + w[c][3];
                if (pending[c] && target[c] < p3) {
                    sel[c] = 4 * i + (target[c] < p0 ? 0u : (target[c] < p1 ? 1u : (target[c] < p2 ? 2u : 3u)));
                    pending[c] = false;
                }
                acc[c] = p3;
                any = any || pending[c];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) atomicAdd((int32_t *)((char *)(s_cnt + c * WIN) + (cl[sel[c]] >> 1)), 1);
}

template <typename IdxT, bool HAS_K, int ELEMS, int WIN, int BS, int RC, int MODE, int C = 1>
__global__ __launch_bounds__(BS) void k_sample16(const IdxT *__restrict__ row_ptr, const uint32_t *__restrict__ col_idx,
                                                 const uint32_t *__restrict__ kmult, const S16Tile *__restrict__ tiles,
                                                 const uint64_t *__restrict__ chunk_tile, const double *__restrict__ gmu,
                                                 const u32x4 *__restrict__ stream16, int32_t *gcnt, SampleArgs a)
{
    constexpr int ROWS_CAP = RC;
    constexpr int RPCH = (ROWS_CAP + 8) / 8;                         // chunks of row entries
    constexpr int NC = (RPCH + ELEMS / 8 + BS - 1) / BS;             // 16-byte chunks per thread
    constexpr int NK = (ROWS_CAP + BS - 1) / BS;
    static_assert((uint32_t)WIN * 8u + 8u < 65536u, "window offsets must fit 16 bits");
    __shared__ __attribute__((aligned(16))) uint32_t s_col[ELEMS + 8];
    __shared__ __attribute__((aligned(16))) uint32_t s_rp[RPCH * 8];
    constexpr int MU_STRIDE = WIN + 2;                               // per-chain window; [WIN] stays 0.0: what pad slots read
    __shared__ __attribute__((aligned(16))) double s_mu[C * MU_STRIDE];
    __shared__ uint32_t s_k[HAS_K ? ROWS_CAP : 1];
    __shared__ int32_t s_cnt[C * WIN];
    const int tid = threadIdx.x;

    const uint64_t t_begin = chunk_tile[blockIdx.x], t_end = chunk_tile[blockIdx.x + 1];
    if (t_begin >= t_end) return;
    const uint64_t nt = t_end - t_begin;
    const S16Tile *__restrict__ T = tiles + t_begin;

    for (int i = tid; i < C * WIN; i += BS) s_cnt[i] = 0;
    if (tid < 2 * C) s_mu[(tid >> 1) * MU_STRIDE + WIN + (tid & 1)] = 0.0;

    // chain c of this launch: mu at gmu + c*n, counts at gcnt + c*n (chain-major, as k_update expects)
    auto flush_window = [&](uint32_t base) {
        for (int i = tid; i < C * WIN; i += BS) {
            const int32_t v = s_cnt[i];
            if (v) { global_count_add(gcnt + (size_t)(i / WIN) * a.n, base + (uint32_t)(i % WIN), v); s_cnt[i] = 0; }
        }
    };
    auto load_window = [&](uint32_t base) {
        for (int i = tid; i < C * WIN; i += BS) {
            const uint32_t c = base + (uint32_t)(i % WIN);
            s_mu[(i / WIN) * MU_STRIDE + (i % WIN)] = c < a.n ? gmu[(size_t)(i / WIN) * a.n + c] : 0.0;
        }
    };

    struct Buf {
        u32x4 pc[NC];
        uint32_t pk[HAS_K ? NK : 1];
    };
    // request a tile's block: raw values only, unconditional loads with clamped indices
    auto issue = [&](const S16Tile &d, Buf &bf) {
        if (!(d.flags & S16_FAST)) return; // uniform
        const uint32_t nch = ((d.nrows + 8) >> 3) + ((d.nnz4 + 7) >> 3);
        const u32x4 *__restrict__ src = stream16 + d.s16;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const uint32_t ch = min((uint32_t)tid + (uint32_t)i * BS, nch - 1);
            bf.pc[i] = __builtin_nontemporal_load(src + ch);
        }
        if (HAS_K) {
#pragma unroll
            for (int i = 0; i < NK; ++i) bf.pk[i] = kmult[d.r0 + min((uint32_t)tid + (uint32_t)i * BS, d.nrows - 1)];
        }
    };
    // unpack the block into LDS: u16 -> u32, nothing else
    auto commit = [&](const S16Tile &d, const Buf &bf) {
        const uint32_t rpch = (d.nrows + 8) >> 3, nch = rpch + ((d.nnz4 + 7) >> 3);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const uint32_t ch = (uint32_t)tid + (uint32_t)i * BS;
            const u32x4 v = bf.pc[i];
            const u32x4 lo = {v.x & 0xffffu, v.x >> 16, v.y & 0xffffu, v.y >> 16};
            const u32x4 hi = {v.z & 0xffffu, v.z >> 16, v.w & 0xffffu, v.w >> 16};
            if (ch < rpch) {
                *(u32x4 *)(s_rp + 8 * ch) = lo;
                *(u32x4 *)(s_rp + 8 * ch + 4) = hi;
            } else if (ch < nch) {
                const uint32_t at = 8 * (ch - rpch);
                *(u32x4 *)(s_col + at) = lo;
                *(u32x4 *)(s_col + at + 4) = hi;
            }
        }
        if (HAS_K) {
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const uint32_t idx = (uint32_t)tid + (uint32_t)i * BS;
                if (idx < d.nrows) s_k[idx] = bf.pk[i];
            }
        }
    };

    // one tile: (window slide) -> commit its buffer -> refill the buffer with the tile two ahead -> walk the rows
    auto process = [&](const S16Tile &d, uint32_t prev_base, const S16Tile &refill, Buf &bf) {
        if (d.flags & S16_EMPTY) { issue(refill, bf); return; }
        if (d.flags & S16_SHIFT) { flush_window(prev_base); load_window(d.wbase); }
        if (d.flags & S16_FAST) {
            commit(d, bf);
            issue(refill, bf);
            __syncthreads();
            if (MODE & K1M_NO_PHASE2) {
                for (uint32_t r = tid; r < d.nrows; r += BS) atomicAdd(&s_cnt[(s_col[s_rp[r] & ~3u] >> 3) & (WIN - 1)], C);
            } else {
                for (uint32_t r = tid; r < d.nrows; r += BS) {
                    const uint32_t e0 = s_rp[r], e1 = s_rp[r + 1];
                    const uint32_t b = e0 & ~3u, L4 = (e1 & ~3u) - b;
                    const uint32_t kk = HAS_K ? s_k[r] : 1u;
                    if (C > 1 && kk == 1) {
                        walk_row_fused<C, WIN, MU_STRIDE>(s_col + b, L4, L4 - (e0 & 3u), s_mu, s_cnt, a, a.row_id_base + d.r0 + r);
                    } else {
#pragma unroll
                        for (int c = 0; c < C; ++c) { // one chain at a time (always the case for C == 1)
                            int32_t *cc = s_cnt + c * WIN;
                            auto add = [&](uint32_t off, int32_t x) { atomicAdd((int32_t *)((char *)cc + (off >> 1)), x); };
                            SampleArgs ac = a;
                            ac.chain = a.chain + (uint32_t)c;
                            RowView4 v{s_col + b, L4, L4 - (e0 & 3u), s_mu + c * MU_STRIDE, {}};
                            allocate_row<HAS_K>(v, add, kk, ac, a.row_id_base + d.r0 + r);
                        }
                    }
                }
            }
            __syncthreads();
            return;
        }
        // slow tile: rows straight from the 32-bit CSR (window lookups / LDS counts where possible)
        issue(refill, bf);
        __syncthreads(); // the window (re)load above must be visible
        const uint32_t wbase = d.wbase;
        for (uint32_t r = tid; r < d.nrows; r += BS) {
            const uint64_t st = (uint64_t)row_ptr[d.r0 + r];
            const uint32_t L = (uint32_t)((uint64_t)row_ptr[d.r0 + r + 1] - st);
            const uint32_t kk = HAS_K ? kmult[d.r0 + r] : 1u;
            for (int c = 0; c < C; ++c) {
                int32_t *cc = s_cnt + c * WIN;
                int32_t *gc = gcnt + (size_t)c * a.n;
                auto add = [&](uint32_t col, int32_t x) {
                    const uint32_t dd = col - wbase;
                    if (dd < (uint32_t)WIN) atomicAdd(&cc[dd], x);
                    else global_count_add(gc, col, x);
                };
                SampleArgs ac = a;
                ac.chain = a.chain + (uint32_t)c;
                RowViewGlobalWin<WIN> v{col_idx + st, L, wbase, s_mu + c * MU_STRIDE, gmu + (size_t)c * a.n};
                allocate_row<HAS_K>(v, add, kk, ac, a.row_id_base + d.r0 + r);
            }
        }
        __syncthreads();
    };

    S16Tile none;
    none.s16 = 0; none.r0 = 0; none.nrows = 0; none.nnz4 = 0; none.wbase = 0; none.flags = S16_EMPTY;
    auto tile_at = [&](uint64_t i) { return i < nt ? T[i] : none; };

    S16Tile dA = tile_at(0), dB = tile_at(1);
    Buf bufA, bufB; // A: even tiles of the range, B: odd tiles
    issue(dA, bufA);
    issue(dB, bufB);
    load_window(dA.wbase); // the first tile's window (its SHIFT flag is never set)
    uint32_t cur_base = dA.wbase; // window base in force = wbase of the tile processed last
    for (uint64_t i = 0; i < nt; i += 2) {
        const S16Tile nA = tile_at(i + 2), nB = tile_at(i + 3); // scalar loads: in flight while A and B are walked
        process(dA, cur_base, nA, bufA);
        cur_base = dA.wbase;
        if (i + 1 < nt) {
            process(dB, cur_base, nB, bufB);
            cur_base = dB.wbase;
        }
        dA = nA;
        dB = nB;
    }
    flush_window(cur_base);
}

struct UpdateArgs {
    int32_t *cnt;          // [C][n]  read, then zeroed
    int32_t *cnt_last;     // [C][n]
    const double *scale;   // n : 1/(beta + l[t])
    double *mu;            // [C][n]
    double *trace;         // [C][trace_len][n] or nullptr
    double *sum_log;       // [C][n]
    double *sum_log2;      // [C][n]
    uint64_t seed;
    double alpha;
    uint32_t n;
    uint32_t n_chains;
    uint32_t chain_base;
    uint32_t iter;
    int32_t sample_idx;    // >= 0: keep this iteration as trace sample; -1: not kept
    uint32_t trace_len;
};

__global__ __launch_bounds__(256) void k_update(UpdateArgs a)
{
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t total = (uint64_t)a.n * a.n_chains;
    if (gid >= total) return;
    const uint32_t c = (uint32_t)(gid / a.n), t = (uint32_t)(gid % a.n);
    const int32_t x = a.cnt[gid];
    a.cnt[gid] = 0;
    a.cnt_last[gid] = x;
    Stream s(a.seed, a.chain_base + c, TAG_GAMMA, (uint64_t)t, a.iter);
    const double m = gamma_unit(s, a.alpha + (double)x) * a.scale[t];
    a.mu[gid] = m;
    if (a.sample_idx >= 0) {
        if (a.trace) a.trace[((uint64_t)c * a.trace_len + (uint32_t)a.sample_idx) * a.n + t] = m;
        const double lg = dlog(m);
        a.sum_log[gid] += lg;
        a.sum_log2[gid] += lg * lg;
    }
}

// out[t*S + s] = in[s*n + t]   (sample-major device trace -> the reference's transcript-major mu_trace)
__global__ __launch_bounds__(256) void k_transpose(const double *__restrict__ in, double *__restrict__ out, uint32_t n,
                                                   uint32_t S)
{
    __shared__ double tile[32][33];
    const uint32_t t0 = blockIdx.x * 32, s0 = blockIdx.y * 32;
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
    for (uint32_t i = ty; i < 32; i += 8) {
        const uint32_t s = s0 + i, t = t0 + tx;
        if (s < S && t < n) tile[i][tx] = in[(uint64_t)s * n + t];
    }
    __syncthreads();
    for (uint32_t i = ty; i < 32; i += 8) {
        const uint32_t t = t0 + i, s = s0 + tx;
        if (s < S && t < n) out[(uint64_t)t * S + s] = tile[tx][i];
    }
}

// ---------------------------------------------------------------- start values (src/mmseq.cpp:617-638)
template <typename IdxT>
__global__ __launch_bounds__(256) void k_start_values(const IdxT *__restrict__ row_ptr, const uint32_t *__restrict__ col_idx,
                                                      const uint32_t *__restrict__ k, uint64_t m, double *acc,
                                                      int32_t *unique_hits)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    const uint64_t b = row_ptr[r], e = row_ptr[r + 1];
    const uint32_t L = (uint32_t)(e - b);
    if (L == 0) return;
    const uint32_t kk = k ? k[r] : 1u;
    const double share = (double)kk / (double)L;
    for (uint64_t j = b; j < e; ++j) unsafeAtomicAdd(&acc[col_idx[j]], share);
    if (L == 1) atomicAdd(&unique_hits[col_idx[b]], (int32_t)kk);
}

__global__ void k_div(double *acc, const double *l, uint32_t n)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) acc[t] = acc[t] / l[t];
}

// ---------------------------------------------------------------- synthetic generator
struct SynthArgs {
    uint64_t seed, row0, rows;
    uint32_t n;
    int32_t uniform;
    const double *cdf;     // n   inclusive running sum of theta*efflen
    const double *len_cdf; // 99  Poisson(avg-1) inclusive cdf
};

__host__ __device__ __forceinline__ uint32_t synth_len_from_u(const double *len_cdf, double u)
{
    uint32_t j = 0;
    while (j < 99 && !(u < len_cdf[j])) ++j;
    return 1 + j;
}
// experiment hook (not part of the generator spec): parity 1 -> odd lengths only, 2 -> even lengths only
__host__ __device__ __forceinline__ uint32_t synth_parity(uint32_t L, int mode)
{
    if (mode & 2) return L | 1u;
    if (mode & 4) return (L & 1u) ? L + 1 : L;
    return L;
}

__device__ __forceinline__ uint32_t synth_first(const SynthArgs &a, double ub)
{
    const uint32_t T = a.n;
    const double target = ub * a.cdf[T - 1];
    uint32_t lo = 0, hi = T - 1;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (target < a.cdf[mid]) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// lens[r] = row length, keys[r] = leading (smallest) transcript of generator row row0 + r
__global__ __launch_bounds__(256) void k_synth_len(SynthArgs a, uint32_t *lens, uint32_t *keys)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.rows) return;
    Stream s(a.seed, 0, TAG_SYNTH_ROW, a.row0 + r, 0);
    double ua, ub;
    s.pair(ua, ub);
    uint32_t L = synth_parity(synth_len_from_u(a.len_cdf, ua), a.uniform);
    if (L > a.n) L = a.n;
    lens[r] = L;
    if (!keys) return;
    // the leading transcript is min(t0, smallest window pick): replay the walk
    const uint32_t T = a.n, t0 = synth_first(a, ub);
    uint32_t best = t0;
    if (L > 1) {
        const uint32_t W = (a.uniform & 1) ? T : (T < 129u ? T : 129u);
        uint32_t wb = 0;
        if (!(a.uniform & 1)) {
            int64_t b = (int64_t)t0 - 64;
            if (b < 0) b = 0;
            if (b + (int64_t)W > (int64_t)T) b = (int64_t)T - (int64_t)W;
            wb = (uint32_t)b;
        }
        const uint32_t nslots = W - 1;
        uint32_t Wp = 1;
        while (Wp < nslots) Wp <<= 1;
        double uc, ud;
        s.pair(uc, ud);
        const uint32_t start = (uint32_t)(uc * (double)Wp);
        const uint32_t stride = ((uint32_t)(ud * (double)(Wp / 2 ? Wp / 2 : 1)) << 1) | 1u;
        uint32_t got = 1, pos = start & (Wp - 1);
        while (got < L) {
            if (pos < nslots) {
                uint32_t t = wb + pos;
                if (t >= t0) t += 1;
                if (t < best) best = t;
                ++got;
            }
            pos = (pos + stride) & (Wp - 1);
        }
    }
    keys[r] = best;
}

template <typename IdxT>
__global__ __launch_bounds__(256) void k_synth_fill(SynthArgs a, const IdxT *__restrict__ row_ptr, const uint32_t *__restrict__ perm,
                                                    uint32_t *col_idx)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.rows) return;
    // stored row r holds generator row perm[r] (identity when unsorted)
    Stream s(a.seed, 0, TAG_SYNTH_ROW, a.row0 + (perm ? (uint64_t)perm[r] : r), 0);
    double ua, ub;
    s.pair(ua, ub);
    uint32_t L = synth_parity(synth_len_from_u(a.len_cdf, ua), a.uniform);
    const uint32_t T = a.n;
    if (L > T) L = T;
    uint32_t *cols = col_idx + (uint64_t)row_ptr[r];
    const uint32_t t0 = synth_first(a, ub);
    cols[0] = t0;
    if (L <= 1) return;
    const uint32_t W = (a.uniform & 1) ? T : (T < 129u ? T : 129u);
    uint32_t wb = 0;
    if (!(a.uniform & 1)) {
        int64_t b = (int64_t)t0 - 64;
        if (b < 0) b = 0;
        if (b + (int64_t)W > (int64_t)T) b = (int64_t)T - (int64_t)W;
        wb = (uint32_t)b;
    }
    const uint32_t nslots = W - 1;
    uint32_t Wp = 1;
    while (Wp < nslots) Wp <<= 1;
    double uc, ud;
    s.pair(uc, ud);
    const uint32_t start = (uint32_t)(uc * (double)Wp);
    const uint32_t stride = ((uint32_t)(ud * (double)(Wp / 2 ? Wp / 2 : 1)) << 1) | 1u;
    uint32_t got = 1, pos = start & (Wp - 1);
    while (got < L) {
        if (pos < nslots) {
            uint32_t t = wb + pos;
            if (t >= t0) t += 1;
            // insertion into the sorted prefix (rows ascend, src/mmseq.cpp:412)
            uint32_t j = got;
            while (j > 0 && cols[j - 1] > t) { cols[j] = cols[j - 1]; --j; }
            cols[j] = t;
            ++got;
        }
        pos = (pos + stride) & (Wp - 1);
    }
}

// ---------------------------------------------------------------- self-test kernels
__global__ void k_selftest_math(int64_t n, const double *x, double *ol, double *oe, double *os, double *orc)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ol[i] = dlog(x[i]);
    oe[i] = dexp(x[i]);
    os[i] = dsqrt(x[i]);
    orc[i] = 1.0 / x[i];
}
__global__ void k_selftest_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out)
{
    const U4 r = philox4x32_10(U4{ctr[0], ctr[1], ctr[2], ctr[3]}, key[0], key[1]);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
    uint32_t a = ctr[0], b = ctr[1];
    philox2x32_10(a, b, key[0]);
    out[4] = a; out[5] = b;
}
__global__ void k_selftest_gamma(uint64_t seed, double shape, double scale, int64_t n, double *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Stream s(seed, 0, TAG_GAMMA, (uint64_t)i, 0);
    out[i] = gamma_unit(s, shape) * scale;
}
__global__ void k_selftest_binomial(uint64_t seed, uint32_t nn, double p, int64_t n, uint32_t *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Stream2 q(seed, 0, TAG_ROW, (uint64_t)i, 0);
    out[i] = binomial(q, nn, p);
}

} // namespace mmg
