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
#include "mmg_math.h"

namespace mmg {

constexpr int K1_BLOCK = 256;
constexpr int K1_LDS_ELEMS = 4096;            // LDS column-id staging buffer (16 KiB)
constexpr int K1_TILE_NNZ = K1_LDS_ELEMS - 8; // hits per tile: leaves the alignment slack of the 16-byte stream
                                              // (16 + 16 + 8 KiB = 40 KiB LDS per workgroup -> 4 workgroups per CU)
constexpr int K1_WIN = 2048;                  // transcripts covered by the LDS window (mu 16 KiB + counts 8 KiB)
constexpr uint32_t K1_WIN_MARGIN = 160;       // hits are expected within this many ids above a row's first hit
constexpr uint32_t K_SMALL = 8u;              // == MMG_K_SMALL
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

struct SampleArgs {
    const void *row_ptr;        // IdxT[m+1]
    const uint32_t *col_idx;    // nnz (+ 32 B padding)
    const uint32_t *k;          // m or nullptr
    const uint64_t *tile_row;   // n_tiles+1
    const uint64_t *chunk_tile; // n_chunks+1 : contiguous tile ranges, one per workgroup visit
    uint64_t n_chunks;
    const double *mu;           // n
    int32_t *cnt;               // n
    uint64_t seed;
    uint64_t row_id_base;
    uint32_t n;
    uint32_t chain;
    uint32_t iter;
};

// One row.  col_at(j) / w_at(j) read the j-th hit and its weight, add(col, v) adds v to the count
// of transcript col (LDS window or global).  Restates src/mmseq.cpp:871-889 for the row.
template <bool HAS_K, typename ColAt, typename WAt, typename Add>
__device__ __forceinline__ void allocate_row(ColAt col_at, WAt w_at, Add add, uint32_t L, uint32_t kk, const SampleArgs &a,
                                             uint64_t row_id)
{
    if (L == 0 || kk == 0) return;
    if (L == 1) { add(col_at(0), (int32_t)kk); return; }
    double total = 0.0;
    for (uint32_t j = 0; j < L; ++j) total += w_at(j);
    const bool degenerate = !(total > 0.0) || !(total < __builtin_huge_val());
    if (!HAS_K || kk <= K_SMALL) {
        Stream s(a.seed, a.chain, TAG_ROW, row_id, a.iter);
        double ua = 0.0, ub = 0.0;
        for (uint32_t d = 0; d < kk; ++d) {
            if ((d & 1u) == 0) s.pair(ua, ub);
            const double u = (d & 1u) ? ub : ua;
            uint32_t sel;
            if (degenerate) {
                sel = (uint32_t)(u * (double)L);
                if (sel >= L) sel = L - 1;
            } else {
                const double target = u * total;
                double acc = 0.0;
                sel = L - 1;
                for (uint32_t j = 0; j < L; ++j) {
                    acc += w_at(j);
                    if (target < acc) { sel = j; break; }
                }
            }
            add(col_at(sel), 1);
        }
        return;
    }
    // conditional-binomial chain (the published gsl_ran_multinomial scheme, src/mmseq.cpp:880)
    SeqStream q(Stream(a.seed, a.chain, TAG_ROW, row_id, a.iter));
    uint32_t remaining = kk;
    double rem_w = total;
    for (uint32_t j = 0; j + 1 < L && remaining > 0; ++j) {
        const double w = w_at(j);
        double p = degenerate ? 1.0 / (double)(L - j) : (rem_w > 0.0 ? w / rem_w : 1.0);
        if (p > 1.0) p = 1.0;
        const uint32_t x = binomial(q, remaining, p);
        if (x) add(col_at(j), (int32_t)x);
        remaining -= x;
        rem_w -= w;
    }
    if (remaining > 0) add(col_at(L - 1), (int32_t)remaining);
}

__device__ __forceinline__ void global_count_add(int32_t *cnt, uint32_t col, int32_t v)
{
    __hip_atomic_fetch_add(&cnt[col], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// K1.  A workgroup walks a contiguous chunk of tiles.  Rows arrive sorted by leading transcript
// (the order hit-set collapse produces), so consecutive tiles touch a slowly advancing band of
// transcripts: that band's mu and counts live in an LDS window [base, base+K1_WIN); hits outside
// the window fall back to the L2 gather / global atomic, so any row order is CORRECT, sorted
// order is FAST (global int32 atomics cap at ~26 G/s on MI355X, LDS atomics at >100 G/s).
template <typename IdxT, bool HAS_K>
__global__ __launch_bounds__(K1_BLOCK) void k_sample(SampleArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_col[K1_LDS_ELEMS];
    __shared__ __attribute__((aligned(16))) double s_mu[K1_WIN];
    __shared__ int32_t s_cnt[K1_WIN];
    const IdxT *__restrict__ row_ptr = (const IdxT *)a.row_ptr;
    const int tid = threadIdx.x;

    for (int i = tid; i < K1_WIN; i += K1_BLOCK) s_cnt[i] = 0;
    uint32_t base = 0xffffffffu; // no window yet (every lookup misses)
    bool win_valid = false;

    auto flush_window = [&]() {
        if (!win_valid) return;
        for (int i = tid; i < K1_WIN; i += K1_BLOCK) {
            const int32_t v = s_cnt[i];
            if (v) { global_count_add(a.cnt, base + (uint32_t)i, v); s_cnt[i] = 0; }
        }
    };

    for (uint64_t chunk = blockIdx.x; chunk < a.n_chunks; chunk += gridDim.x) {
        const uint64_t tile_end = a.chunk_tile[chunk + 1];
        for (uint64_t tile = a.chunk_tile[chunk]; tile < tile_end; ++tile) {
            const uint64_t r0 = a.tile_row[tile], r1 = a.tile_row[tile + 1];
            const uint64_t nz0 = (uint64_t)row_ptr[r0], nz1 = (uint64_t)row_ptr[r1];
            const uint64_t nt = nz1 - nz0;
            if (nt > (uint64_t)K1_TILE_NNZ) {
                // a single row longer than a tile: one lane walks it straight from global memory
                if (tid == 0) {
                    const uint32_t *cols = a.col_idx + nz0;
                    const double *mu = a.mu;
                    int32_t *cnt = a.cnt;
                    allocate_row<HAS_K>([&](uint32_t j) { return cols[j]; }, [&](uint32_t j) { return mu[cols[j]]; },
                                        [&](uint32_t c, int32_t v) { global_count_add(cnt, c, v); }, (uint32_t)nt,
                                        HAS_K ? a.k[r0] : 1u, a, a.row_id_base + r0);
                }
                continue;
            }
            if (nt == 0) continue; // only empty rows
            // ---- phase 1: stream the tile's column ids into LDS (16-byte aligned, non-temporal)
            const uint64_t abase = nz0 & ~(uint64_t)3;
            const uint32_t shift = (uint32_t)(nz0 - abase);
            const uint32_t nchunks = (uint32_t)((nz1 - abase + 3) >> 2);
            const u32x4 *__restrict__ src = (const u32x4 *)(a.col_idx + abase);
            for (uint32_t ch = tid; ch < nchunks; ch += K1_BLOCK)
                *(u32x4 *)(s_col + 4 * ch) = __builtin_nontemporal_load(src + ch);
            __syncthreads();
            // ---- window decision (uniform): first hit of the first / last non-empty row
            const uint32_t cmin = s_col[shift];
            uint64_t rl = r1 - 1;
            while (rl > r0 && (uint64_t)row_ptr[rl] == nz1) --rl; // skip trailing empty rows
            const uint32_t clast = s_col[(uint32_t)((uint64_t)row_ptr[rl] - nz0) + shift];
            const bool keep = win_valid && cmin >= base && (uint64_t)clast + K1_WIN_MARGIN <= (uint64_t)base + K1_WIN;
            if (!keep) {
                flush_window();
                base = cmin & ~15u;
                win_valid = true;
                for (int i = tid; i < K1_WIN; i += K1_BLOCK) {
                    const uint32_t c = base + (uint32_t)i;
                    s_mu[i] = c < a.n ? a.mu[c] : 0.0;
                }
                __syncthreads();
            }
            // ---- phase 2: one lane per row walks its LDS segment
            const uint32_t wbase = base;
            const double *__restrict__ gmu = a.mu;
            int32_t *gcnt = a.cnt;
            for (uint64_t r = r0 + tid; r < r1; r += K1_BLOCK) {
                const uint32_t b = (uint32_t)((uint64_t)row_ptr[r] - nz0) + shift;
                const uint32_t L = (uint32_t)((uint64_t)row_ptr[r + 1] - (uint64_t)row_ptr[r]);
                const uint32_t *cl = s_col + b;
                allocate_row<HAS_K>(
                    [&](uint32_t j) { return cl[j]; },
                    [&](uint32_t j) {
                        const uint32_t c = cl[j], d = c - wbase;
                        return d < (uint32_t)K1_WIN ? s_mu[d] : gmu[c];
                    },
                    [&](uint32_t c, int32_t v) {
                        const uint32_t d = c - wbase;
                        if (d < (uint32_t)K1_WIN) atomicAdd(&s_cnt[d], v);
                        else global_count_add(gcnt, c, v);
                    },
                    L, HAS_K ? a.k[r] : 1u, a, a.row_id_base + r);
            }
            __syncthreads();
        }
    }
    flush_window();
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

// ---------------------------------------------------------------- EM (src/mmseq.cpp:741-811)
// d_i = sum_{t in row i} mu_t ; loglik part sum_i k_i log d_i ; acc_t += k_i / d_i
template <typename IdxT>
__global__ __launch_bounds__(256) void k_em_rows(const IdxT *__restrict__ row_ptr, const uint32_t *__restrict__ col_idx,
                                                 const uint32_t *__restrict__ k, uint64_t m, const double *__restrict__ mu,
                                                 double *acc, double *loglik)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double ll = 0.0;
    if (r < m) {
        const uint64_t b = row_ptr[r], e = row_ptr[r + 1];
        if (e > b) {
            double d = 0.0;
            for (uint64_t j = b; j < e; ++j) d += mu[col_idx[j]];
            const double kk = k ? (double)k[r] : 1.0;
            ll = kk * log(d);
            if (acc) {
                const double q = kk / d;
                for (uint64_t j = b; j < e; ++j) unsafeAtomicAdd(&acc[col_idx[j]], q);
            }
        }
    }
    // block reduction of the log-likelihood part
    __shared__ double red[256];
    red[threadIdx.x] = ll;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) unsafeAtomicAdd(loglik, red[0]);
}

// mu_t <- mu_t * acc_t / l_t ; loglik -= mu_t l_t (new mu) ; acc zeroed for the next sweep
__global__ __launch_bounds__(256) void k_em_cols(double *mu, double *acc, const double *__restrict__ l, uint32_t n,
                                                 double *loglik, int apply)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    double pen = 0.0;
    if (t < n) {
        double m = mu[t];
        if (apply) {
            m = m * acc[t] / l[t];
            mu[t] = m;
            acc[t] = 0.0;
        }
        pen = m * l[t];
    }
    __shared__ double red[256];
    red[threadIdx.x] = pen;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) unsafeAtomicAdd(loglik, -red[0]);
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
    uint32_t L = synth_len_from_u(a.len_cdf, ua);
    if (L > a.n) L = a.n;
    lens[r] = L;
    if (!keys) return;
    // the leading transcript is min(t0, smallest window pick): replay the walk
    const uint32_t T = a.n, t0 = synth_first(a, ub);
    uint32_t best = t0;
    if (L > 1) {
        const uint32_t W = a.uniform ? T : (T < 129u ? T : 129u);
        uint32_t wb = 0;
        if (!a.uniform) {
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
    uint32_t L = synth_len_from_u(a.len_cdf, ua);
    const uint32_t T = a.n;
    if (L > T) L = T;
    uint32_t *cols = col_idx + (uint64_t)row_ptr[r];
    const uint32_t t0 = synth_first(a, ub);
    cols[0] = t0;
    if (L <= 1) return;
    const uint32_t W = a.uniform ? T : (T < 129u ? T : 129u);
    uint32_t wb = 0;
    if (!a.uniform) {
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
    SeqStream q(Stream(seed, 0, TAG_ROW, (uint64_t)i, 0));
    out[i] = binomial(q, nn, p);
}

} // namespace mmg
