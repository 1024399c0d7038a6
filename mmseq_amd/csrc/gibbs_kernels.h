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
#include "mmg_types.h"
#include "mmg_math.h"

namespace mmg {

constexpr uint32_t K1_WIN_MARGIN = 160;       // k_sample keeps its window while the leading transcripts stay this far below its end
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

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
// CHAIN = false: the rows on the conditional-binomial chain are left to k_sample_bigk (bigk_kernels.h; the sliced-ELL kernels)
template <bool HAS_K, bool CHAIN = true, typename View, typename Add>
__device__ __forceinline__ void allocate_row(const View &v, Add add, uint32_t kk, const SampleArgs &a, uint64_t row_id)
{
    const uint32_t L = v.L;
    if (L == 0 || kk == 0) return;
    if (L == 1) { add(v.col(0), (int32_t)kk); return; }
    const double total = v.total();
    const bool degenerate = !(total > 0.0) || !(total < __builtin_huge_val());
    if (!HAS_K || draws_categoricals(kk, L)) {
        Stream2 s(a.seed, a.chain, TAG_ROW, row_id, a.iter);
        const double ts = total * 0x1p-32, hs = ts * 0.5; // draw_target (mmg_math.h)
        for (uint32_t d = 0; d < kk; ++d) {
            const uint32_t x = s.next_word();
            uint32_t sel;
            if (degenerate) {
                sel = (uint32_t)(u32_unit(x) * (double)L);
                if (sel >= L) sel = L - 1;
            } else {
                sel = v.pick(draw_target(x, ts, hs));
            }
            add(v.col(sel), 1);
        }
        return;
    }
    if (!CHAIN) return;
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
                                                  const uint32_t *__restrict__ kmult /* may be null */,
                                                  const uint64_t *__restrict__ tile_row, uint64_t n_tiles, TileDesc *out)
{
    const uint64_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const uint64_t r0 = tile_row[tile], r1 = tile_row[tile + 1];
    const uint64_t nz0 = row_ptr[r0], nz1 = row_ptr[r1];
    uint32_t mx = 0, mn = 0xffffffffu, n4 = 0, ml = 0, kmx = 1, kn1 = 0;
    for (uint64_t j = nz0 + threadIdx.x; j < nz1; j += 64) { const uint32_t c = col_idx[j]; mx = max(mx, c); mn = min(mn, c); }
    for (uint64_t r = r0 + threadIdx.x; r < r1; r += 64) {
        const uint32_t L = (uint32_t)min((uint64_t)0xffffffffu, (uint64_t)row_ptr[r + 1] - (uint64_t)row_ptr[r]);
        n4 += (L + 3u) & ~3u;
        ml = max(ml, L);
        if (kmult) { const uint32_t kk = kmult[r]; kmx = max(kmx, kk); kn1 += kk != 1u; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
        mn = min(mn, (uint32_t)__shfl_xor((int)mn, off));
        n4 += (uint32_t)__shfl_xor((int)n4, off);
        ml = max(ml, (uint32_t)__shfl_xor((int)ml, off));
        kmx = max(kmx, (uint32_t)__shfl_xor((int)kmx, off));
        kn1 += (uint32_t)__shfl_xor((int)kn1, off);
    }
    if (threadIdx.x == 0) {
        TileDesc d;
        d.nz0 = nz0; d.r0 = r0; d.nrows = (uint32_t)(r1 - r0);
        d.nnz = (uint32_t)min((uint64_t)0xffffffffu, nz1 - nz0);
        d.cmin = 0; d.clast = 0; d.cmax = mx; d.call = mn; d.nnz4 = n4; d.maxlen = ml; d.kmax = kmx; d.knot1 = kn1;
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
template <typename IdxT, bool HAS_K, int ELEMS, int WIN, int UNR, int BS, int RC>
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
    gcnt += (size_t)(blockIdx.x & a.cnt_rep_mask) * a.cnt_rep_stride; // (mmg_types.h: CNT_REPLICAS)

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
        nd.nnz = 0; nd.nrows = 0; nd.nz0 = 0; nd.r0 = 0; nd.cmin = nd.clast = nd.cmax = nd.call = 0; nd.nnz4 = 0; nd.maxlen = 0;
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
        {
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

} // namespace mmg
