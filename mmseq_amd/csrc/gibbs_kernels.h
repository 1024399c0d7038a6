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
constexpr int K1_TILE_NNZ = 4096;            // hits staged per tile (LDS: 4096 * 12 B = 48 KiB)
constexpr int K1_LDS_ELEMS = K1_TILE_NNZ + 8; // + alignment slack of the 16-byte stream
constexpr uint32_t K_SMALL = 8u;              // == MMG_K_SMALL
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

struct SampleArgs {
    const void *row_ptr;      // IdxT[m+1]
    const uint32_t *col_idx;  // nnz (+ 16 B padding)
    const uint32_t *k;        // m or nullptr
    const uint64_t *tile_row; // n_tiles+1
    uint64_t n_tiles;
    const double *mu;         // n
    int32_t *cnt;             // n
    uint64_t seed;
    uint64_t row_id_base;
    uint32_t chain;
    uint32_t iter;
};

// One row: cols/w point at the row's segment (LDS or global-gathered), counts added atomically.
template <bool HAS_K, typename ColAt, typename WAt>
__device__ __forceinline__ void allocate_row(ColAt col_at, WAt w_at, uint32_t L, uint32_t kk, const SampleArgs &a,
                                             uint64_t row_id)
{
    if (L == 0 || kk == 0) return;
    if (L == 1) {
        __hip_atomic_fetch_add(&a.cnt[col_at(0)], (int32_t)kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
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
            __hip_atomic_fetch_add(&a.cnt[col_at(sel)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        if (x) __hip_atomic_fetch_add(&a.cnt[col_at(j)], (int32_t)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        remaining -= x;
        rem_w -= w;
    }
    if (remaining > 0)
        __hip_atomic_fetch_add(&a.cnt[col_at(L - 1)], (int32_t)remaining, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename IdxT, bool HAS_K>
__global__ __launch_bounds__(K1_BLOCK) void k_sample(SampleArgs a)
{
    __shared__ __attribute__((aligned(16))) double s_w[K1_LDS_ELEMS];
    __shared__ __attribute__((aligned(16))) uint32_t s_col[K1_LDS_ELEMS];
    const IdxT *__restrict__ row_ptr = (const IdxT *)a.row_ptr;
    const int tid = threadIdx.x;

    for (uint64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const uint64_t r0 = a.tile_row[tile], r1 = a.tile_row[tile + 1];
        const uint64_t nz0 = (uint64_t)row_ptr[r0], nz1 = (uint64_t)row_ptr[r1];
        const uint64_t nt = nz1 - nz0;
        if (nt > (uint64_t)K1_TILE_NNZ) {
            // a single row longer than a tile: one lane walks it straight from global memory
            if (tid == 0) {
                const uint32_t *cols = a.col_idx + nz0;
                const double *mu = a.mu;
                allocate_row<HAS_K>([&](uint32_t j) { return cols[j]; }, [&](uint32_t j) { return mu[cols[j]]; },
                                    (uint32_t)nt, HAS_K ? a.k[r0] : 1u, a, a.row_id_base + r0);
            }
            continue;
        }
        // ---- phase 1: stream the tile's hits (16-byte aligned chunks), gather weights -> LDS
        const uint64_t abase = nz0 & ~(uint64_t)3;
        const uint32_t shift = (uint32_t)(nz0 - abase);
        const uint32_t nchunks = (uint32_t)((nz1 - abase + 3) >> 2);
        const u32x4 *__restrict__ src = (const u32x4 *)(a.col_idx + abase);
        for (uint32_t ch = tid; ch < nchunks; ch += K1_BLOCK) {
            const u32x4 c = __builtin_nontemporal_load(src + ch);
            const double w0 = a.mu[c.x], w1 = a.mu[c.y], w2 = a.mu[c.z], w3 = a.mu[c.w];
            *(u32x4 *)(s_col + 4 * ch) = c;
            *(f64x2 *)(s_w + 4 * ch) = f64x2{w0, w1};
            *(f64x2 *)(s_w + 4 * ch + 2) = f64x2{w2, w3};
        }
        __syncthreads();
        // ---- phase 2: one lane per row walks its LDS segment
        for (uint64_t r = r0 + tid; r < r1; r += K1_BLOCK) {
            const uint32_t b = (uint32_t)((uint64_t)row_ptr[r] - nz0) + shift;
            const uint32_t L = (uint32_t)((uint64_t)row_ptr[r + 1] - (uint64_t)row_ptr[r]);
            const double *w = s_w + b;
            const uint32_t *cl = s_col + b;
            allocate_row<HAS_K>([&](uint32_t j) { return cl[j]; }, [&](uint32_t j) { return w[j]; }, L,
                                HAS_K ? a.k[r] : 1u, a, a.row_id_base + r);
        }
        __syncthreads();
    }
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

__global__ __launch_bounds__(256) void k_synth_len(SynthArgs a, uint32_t *lens)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.rows) return;
    Stream s(a.seed, 0, TAG_SYNTH_ROW, a.row0 + r, 0);
    double ua, ub;
    s.pair(ua, ub);
    uint32_t L = synth_len_from_u(a.len_cdf, ua);
    if (L > a.n) L = a.n;
    lens[r] = L;
}

template <typename IdxT>
__global__ __launch_bounds__(256) void k_synth_fill(SynthArgs a, const IdxT *__restrict__ row_ptr, uint32_t *col_idx)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.rows) return;
    Stream s(a.seed, 0, TAG_SYNTH_ROW, a.row0 + r, 0);
    double ua, ub;
    s.pair(ua, ub);
    uint32_t L = synth_len_from_u(a.len_cdf, ua);
    const uint32_t T = a.n;
    if (L > T) L = T;
    uint32_t *cols = col_idx + (uint64_t)row_ptr[r];
    const double target = ub * a.cdf[T - 1];
    uint32_t lo = 0, hi = T - 1;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (target < a.cdf[mid]) hi = mid; else lo = mid + 1;
    }
    const uint32_t t0 = lo;
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
