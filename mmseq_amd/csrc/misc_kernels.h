// misc_kernels.h -- K2 (Gamma redraw + trace capture, src/mmseq.cpp:896-917), trace read-out, start values
// (src/mmseq.cpp:617-638), the synthetic generator of the benchmark inputs, and the self-test kernels.
#pragma once
#include "mmg_types.h"
#include "mmg_math.h"

namespace mmg {

// K2: one lane per (chain, transcript).  HBM: 28 bytes per lane on kept iterations.
__global__ __launch_bounds__(256) void k_update(UpdateArgs a)
{
    const uint32_t c = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x; // grid.y = chains: no 64-bit division per lane
    if (t >= a.n) return;
    const uint64_t gid = (uint64_t)c * a.n + t;
    // the counts of the iteration: the sum over the replicas K1's workgroups added into (mmg_types.h: CNT_REPLICAS); all loads go out together
    int32_t x = a.cnt[gid];
    if (a.cnt_replicas > 1) { // uniform
        int32_t xr[CNT_REPLICAS - 1];
#pragma unroll
        for (uint32_t r = 1; r < CNT_REPLICAS; ++r) xr[r - 1] = a.cnt[(uint64_t)r * a.cnt_rep_stride + gid];
#pragma unroll
        for (uint32_t r = 1; r < CNT_REPLICAS; ++r) x += xr[r - 1];
    }
    // the moments are read-modify-write: their loads go out with the count's, not after the draw (a wave of this kernel is one long
    // dependent chain -- at config 2 there is one wave per SIMD and nothing to hide a second memory round trip behind)
    double sl = 0.0, sl2 = 0.0;
    if (a.sample_idx >= 0) { sl = a.sum_log[gid]; sl2 = a.sum_log2[gid]; }
    a.cnt[gid] = 0;
    if (a.cnt_replicas > 1) {
#pragma unroll
        for (uint32_t r = 1; r < CNT_REPLICAS; ++r) a.cnt[(uint64_t)r * a.cnt_rep_stride + gid] = 0;
    }
    a.cnt_last[gid] = x;
    // the Gamma stream is keyed by the CALLER's transcript id: the chain does not depend on the device numbering
    Stream s(a.seed, a.chain_base + c, TAG_GAMMA, (uint64_t)(a.ext_of_int ? a.ext_of_int[t] : t), a.iter);
    const double m = gamma_unit(s, a.alpha + (double)x) * a.scale[t];
    a.mu[gid] = m;
    if (a.sample_idx >= 0) {
        if (a.trace) a.trace[((uint64_t)c * a.trace_len + (uint32_t)a.sample_idx) * a.n + t] = m;
        const double lg = dlog(m);
        a.sum_log[gid] = sl + lg;
        a.sum_log2[gid] = sl2 + lg * lg;
    }
}

// out[t*S + s] = in[s*n + perm(t)]   (sample-major device trace -> the reference's transcript-major mu_trace, :914)
__global__ __launch_bounds__(256) void k_transpose(const double *__restrict__ in, double *__restrict__ out, uint32_t n,
                                                   uint32_t S, const uint32_t *__restrict__ int_of_ext)
{
    __shared__ double tile[32][33];
    const uint32_t t0 = blockIdx.x * 32, s0 = blockIdx.y * 32;
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
    const uint32_t tt = t0 + tx;
    const uint32_t src = tt < n ? (int_of_ext ? int_of_ext[tt] : tt) : 0;
    for (uint32_t i = ty; i < 32; i += 8) {
        const uint32_t s = s0 + i;
        if (s < S && tt < n) tile[i][tx] = in[(uint64_t)s * n + src];
    }
    __syncthreads();
    for (uint32_t i = ty; i < 32; i += 8) {
        const uint32_t t = t0 + i, s = s0 + tx;
        if (s < S && t < n) out[(uint64_t)t * S + s] = tile[tx][i];
    }
}

// out[r*n + t] = in[r*n + perm(t)]
template <typename T>
__global__ __launch_bounds__(256) void k_gather_rows(const T *__restrict__ in, T *__restrict__ out, uint32_t n, uint32_t rows,
                                                     const uint32_t *__restrict__ int_of_ext)
{
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)n * rows) return;
    const uint32_t r = (uint32_t)(gid / n), t = (uint32_t)(gid % n);
    out[gid] = in[(uint64_t)r * n + (int_of_ext ? int_of_ext[t] : t)];
}

// ---------------------------------------------------------------- start values (src/mmseq.cpp:617-638)
// mu0[t] = (sum over rows i containing t of k_i / |row i|) / l[t].  The reference adds the shares in row order; here every
// share q = (double)k / (double)L is turned into the integer floor(q * 2^52) (< 2^85) and added as three 32-bit limbs with
// integer atomics, so the sum is exact, independent of the order of addition, and equal to the oracle's bit for bit.
__host__ __device__ __forceinline__ void start_share_limbs(uint32_t kk, uint32_t L, uint64_t &a0, uint64_t &a1, uint64_t &a2)
{
    const double q = (double)kk / (double)L;           // 2^-32 <= q < 2^32
    const uint64_t b = bits_of(q);
    const int e = (int)((b >> 52) & 0x7ff) - 1023;     // ilogb(q) in [-32, 31]; q is normal
    const uint64_t m = (b & 0xfffffffffffffull) | (1ull << 52);
    uint64_t lo, hi;                                   // floor(q * 2^52) = m * 2^e
    if (e >= 0) { lo = m << e; hi = e ? m >> (64 - e) : 0; }
    else { lo = m >> (-e); hi = 0; }
    a0 = lo & 0xffffffffull; a1 = lo >> 32; a2 = hi;
}

// A workgroup takes SV_ROWS consecutive rows, 256 at a time, and keeps the limbs of a SV_WIN-wide window of transcripts in LDS: the stored
// rows come band by band (canonical order), so nearly every hit of a stretch of rows falls into one window, and what reaches the global
// accumulators is one atomic per touched transcript and limb when the window moves on -- not two or three per HIT (one pass of global
// 64-bit atomics over 1.0 G hits took 97.7 ms at config 3: as much as 400 sweeps of K1).  Hits outside the window go to the global
// accumulators directly: rows in any order are summed correctly, rows in canonical order quickly.  Integer sums: any order, same bits.
constexpr uint32_t SV_WIN = 512, SV_ROWS = 4096;
template <typename IdxT>
__global__ __launch_bounds__(256) void k_start_values(const IdxT *__restrict__ row_ptr, const uint32_t *__restrict__ col_idx,
                                                      const uint32_t *__restrict__ k, uint64_t m, uint32_t n, uint64_t *acc /* [3][n] */,
                                                      int32_t *unique_hits)
{
    __shared__ unsigned long long s_acc[3][SV_WIN];
    __shared__ int32_t s_uh[SV_WIN];
    __shared__ uint32_t s_min;
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < SV_WIN; i += 256) { s_acc[0][i] = 0; s_acc[1][i] = 0; s_acc[2][i] = 0; s_uh[i] = 0; }
    uint32_t wbase = 0;
    bool have = false;
    auto flush = [&]() {
        for (uint32_t i = tid; i < SV_WIN; i += 256) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const unsigned long long v = s_acc[q][i];
                if (v) { atomicAdd((unsigned long long *)&acc[(size_t)q * n + wbase + i], v); s_acc[q][i] = 0; }
            }
            const int32_t u = s_uh[i];
            if (u) { atomicAdd(&unique_hits[wbase + i], u); s_uh[i] = 0; }
        }
    };
    const uint64_t r_begin = (uint64_t)blockIdx.x * SV_ROWS, r_end = r_begin + SV_ROWS < m ? r_begin + SV_ROWS : m;
    for (uint64_t r0 = r_begin; r0 < r_end; r0 += 256) {
        const uint64_t r = r0 + tid;
        uint64_t b = 0, e = 0;
        uint32_t kk = 0;
        if (r < r_end) { b = row_ptr[r]; e = row_ptr[r + 1]; kk = k ? k[r] : 1u; }
        const uint32_t L = (uint32_t)(e - b);
        // the window of this stretch: from the band of the smallest leading transcript among its rows
        if (tid == 0) s_min = 0xffffffffu;
        __syncthreads();
        if (L && kk) atomicMin(&s_min, col_idx[b]);
        __syncthreads();
        const uint32_t lead = s_min;
        __syncthreads(); // (every thread has read it before thread 0 resets it for the next stretch: the branches below are uniform)
        if (lead != 0xffffffffu) {
            const uint32_t nb = lead & ~63u;
            if (!have || nb < wbase || nb + 256u > wbase + SV_WIN) { // (uniform)
                if (have) { __syncthreads(); flush(); }
                wbase = nb;
                have = true;
                __syncthreads();
            }
        }
        if (L && kk) {
            uint64_t a0, a1, a2;
            start_share_limbs(kk, L, a0, a1, a2);
            for (uint64_t j = b; j < e; ++j) {
                const uint32_t t = col_idx[j], d = t - wbase;
                if (d < SV_WIN) {
                    if (a0) atomicAdd(&s_acc[0][d], (unsigned long long)a0);
                    if (a1) atomicAdd(&s_acc[1][d], (unsigned long long)a1);
                    if (a2) atomicAdd(&s_acc[2][d], (unsigned long long)a2);
                } else {
                    if (a0) atomicAdd((unsigned long long *)&acc[t], (unsigned long long)a0);
                    if (a1) atomicAdd((unsigned long long *)&acc[(size_t)n + t], (unsigned long long)a1);
                    if (a2) atomicAdd((unsigned long long *)&acc[2 * (size_t)n + t], (unsigned long long)a2);
                }
            }
            if (L == 1) {
                const uint32_t t = col_idx[b], d = t - wbase;
                if (d < SV_WIN) atomicAdd(&s_uh[d], (int32_t)kk);
                else atomicAdd(&unique_hits[t], (int32_t)kk);
            }
        }
    }
    __syncthreads();
    if (have) flush();
}

// ---------------------------------------------------------------- synthetic generator
// Row r is a pure function of (seed, row0 + r): length 1 + Poisson(avg - 1) clipped to [1, 100]; the first transcript follows
// the abundance table, the others are distinct members of a 129-wide index window around it (uniform: of all transcripts).
// far_fraction > 0: a row of >= 2 hits is "far" with that probability; its last drawn hit is then a transcript anywhere in
// [0, n) outside the window (a read that also hits a paralogue).
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

__global__ __launch_bounds__(256) void k_synth_len(SynthArgs a, uint32_t *lens)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.rows) return;
    Stream s(a.seed, 0, TAG_SYNTH_ROW, a.row0 + r, 0);
    double ua, ub;
    s.pair(ua, ub);
    uint32_t L = synth_len_from_u(a.len_cdf, ua);
    if (L > a.n) L = a.n;
    if (a.gene_size) { // gene-block mode: a read's hits are isoforms of its gene
        const uint32_t g0 = synth_first(a, ub) / a.gene_size, W = min(a.gene_size, a.n - g0 * a.gene_size);
        if (L > W) L = W;
    }
    lens[r] = L;
}

__global__ __launch_bounds__(256) void k_synth_fill(SynthArgs a, double far_fraction, const uint64_t *__restrict__ row_ptr,
                                                    uint32_t *col_idx)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.rows) return;
    Stream s(a.seed, 0, TAG_SYNTH_ROW, a.row0 + r, 0);
    double ua, ub;
    s.pair(ua, ub);
    uint32_t L = synth_len_from_u(a.len_cdf, ua);
    const uint32_t T = a.n;
    if (L > T) L = T;
    uint32_t *cols = col_idx + row_ptr[r];
    const uint32_t t0 = synth_first(a, ub);
    cols[0] = t0;
    if (L <= 1) return;
    uint32_t W = a.uniform ? T : (T < 129u ? T : 129u);
    uint32_t wb = 0;
    if (a.gene_size) {
        wb = (t0 / a.gene_size) * a.gene_size;
        W = min(a.gene_size, T - wb);
        if (L > W) L = W;
        if (L <= 1) return;
    } else if (!a.uniform) {
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
    bool far = false;
    uint32_t tfar = 0;
    if (far_fraction > 0.0 && a.gene_size && a.far_family >= 2u && a.n_genes >= 2u * a.far_family) {
        // paralogue families: the far hit is an isoform of ANOTHER gene of the read's family (the genes whose images under
        // g -> fam_a * g mod n_genes share a block of far_family consecutive values: scattered over the transcriptome, fixed per gene)
        double ue, uf;
        s.pair(ue, uf);
        const uint32_t F = a.far_family, g0 = t0 / a.gene_size;
        const uint32_t pg = (uint32_t)(((uint64_t)a.fam_a * g0) % a.n_genes), fam = pg / F, idx = pg % F;
        const uint32_t fsize = min(F, a.n_genes - fam * F);
        if (fsize >= 2u) {
            far = ue < far_fraction;
            uint32_t j = 1u + (uint32_t)(uf * (double)(fsize - 1u));
            if (j > fsize - 1u) j = fsize - 1u;
            const uint32_t pm = fam * F + (idx + j) % fsize;
            const uint32_t gm = (uint32_t)(((uint64_t)a.fam_ainv * pm) % a.n_genes);
            double ug, uh;
            s.pair(ug, uh);
            const uint32_t Wm = min(a.gene_size, T - gm * a.gene_size);
            uint32_t iso = (uint32_t)(ug * (double)Wm);
            if (iso >= Wm) iso = Wm - 1u;
            tfar = gm * a.gene_size + iso;            // another gene: outside the read's own, hence distinct from every other hit
        }
    } else if (far_fraction > 0.0 && !a.uniform && T > 2u * W) {
        double ue, uf;
        s.pair(ue, uf);
        far = ue < far_fraction;
        tfar = (uint32_t)(uf * (double)T);
        if (tfar >= T) tfar = T - 1;
        if (tfar >= wb && tfar < wb + W) tfar = (tfar + W) % T; // outside the window, hence distinct from every other hit
    }
    uint32_t got = 1, pos = start & (Wp - 1);
    while (got < L) {
        if (pos < nslots) {
            uint32_t t = wb + pos;
            if (t >= t0) t += 1;
            if (far && got == L - 1) t = tfar;
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
    // the branch-free variant wherever it is defined (positive, normal, finite): the test holds both to the oracle's log
    ol[i] = (x[i] >= 0x1p-1022 && x[i] < __builtin_huge_val()) ? dlog_pn(x[i]) : dlog(x[i]);
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

// BTRS candidates over (n, p) drawn from the case index: n log-uniform in [n_lo, n_hi], p log-uniform in [10 / n, 1/2] (where binomial() takes
// BTRS), one attempt each from the case's keyed stream.  counts[0] attempts that reach the exact test, [1] of them decided by btrs_pretest,
// [2] decided AND different from the fp64 test (must stay 0), [3] accepted by the fp64 test, [4] max |estimate - fp64 difference| / bound, in millionths.
__global__ __launch_bounds__(256) void k_selftest_btrs_pretest(uint64_t seed, int64_t n_cases, double n_lo, double n_hi, unsigned long long *counts)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool slow = false, decided = false, wrong = false, acc = false;
    if (i < n_cases) {
        Stream s(seed, 0, TAG_SYNTH_ROW, (uint64_t)i, 0);
        double ua, ub;
        s.pair(ua, ub);
        const double dn = dfloor(dexp(dlog(n_lo) + ua * (dlog(n_hi) - dlog(n_lo))));
        const double p_lo = 10.0 / dn;
        if (dn >= 21.0 && p_lo < 0.5) {
            const double p = dexp(dlog(p_lo) + ub * (dlog(0.5) - dlog(p_lo)));
            if (dn * p >= 10.0 && p <= 0.5) {
                const double qq = 1.0 - p, spq = dsqrt(dn * p * qq);
                const double b = 1.15 + 2.53 * spq, a = -0.0873 + 0.0248 * b + 0.01 * p, c = dn * p + 0.5, vr = 0.92 - 4.2 / b;
                Stream2 q(seed, 1, TAG_ROW, (uint64_t)i, 0);
                const double u = q.next() - 0.5, v = q.next();
                const double us = 0.5 - dabs(u);
                const double kf = dfloor((2.0 * a / us + b) * u + c);
                if (!(kf < 0.0 || kf > dn) && !(us >= 0.07 && v <= vr)) {
                    slow = true;
                    double diff = 0.0;
                    acc = btrs_exact_test(dn, p, kf, us, v, a, b, spq, &diff);
                    float d32, e32;
                    btrs_estimate(dn, p, kf, us, v, a, b, spq, d32, e32);
                    const int pre = d32 > e32 ? 1 : (d32 < -e32 ? -1 : 0);
                    decided = pre != 0;
                    wrong = decided && ((pre > 0) != acc);
                    // the estimate's actual error as a share of its bound, in millionths (counts[4]: the largest seen)
                    const double share = dabs((double)d32 - diff) / (double)e32 * 1e6;
                    if (share == share) atomicMax(&counts[4], (unsigned long long)(share < 1.8e19 ? share : 1.8e19));
                }
            }
        }
    }
    const unsigned long long c0 = __popcll(__ballot(slow)), c1 = __popcll(__ballot(decided)), c2 = __popcll(__ballot(wrong)), c3 = __popcll(__ballot(acc));
    if ((threadIdx.x & 63) == 0) {
        if (c0) atomicAdd(&counts[0], c0);
        if (c1) atomicAdd(&counts[1], c1);
        if (c2) atomicAdd(&counts[2], c2);
        if (c3) atomicAdd(&counts[3], c3);
    }
}

// mmg_selftest_binv_pretest: binv_pretest (mmg_math.h) against the fp64 search over the inversion's whole range -- n log-uniform in [n_lo, n_hi],
// n p log-uniform in [1e-6, 10), p <= 1/2, the uniform of the sampler's row stream.  counts: [0] cases, [1] decided by the fp32 search, [2] decided
// AND different from the fp64 search (must stay 0), [3] cases whose fp64 search fell off the end (binomial() draws again), [4] the sum of the outcomes.
__global__ __launch_bounds__(256) void k_selftest_binv_pretest(uint64_t seed, int64_t n_cases, double n_lo, double n_hi, float slack, unsigned long long *counts)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool valid = false, decided = false, wrong = false, fell = false;
    uint32_t xs = 0;
    if (i < n_cases) {
        Stream s(seed, 0, TAG_SYNTH_ROW, (uint64_t)i, 0);
        double ua, ub;
        s.pair(ua, ub);
        const double dn = dfloor(dexp(dlog(n_lo) + ua * (dlog(n_hi) - dlog(n_lo))));
        const double np = dexp(dlog(1e-6) + ub * (dlog(10.0) - dlog(1e-6)));
        const double p = np / dn;
        if (dn >= 1.0 && p > 0.0 && p <= 0.5 && dn * p < 10.0) {
            valid = true;
            Stream2 q(seed, 1, TAG_ROW, (uint64_t)i, 0);
            const double u = q.next();
            uint32_t x64 = 0;
            const bool ok = binv_exact(dn, p, u, (uint32_t)dn, x64);
            fell = !ok;
            const int pre = binv_pretest(dn, p, u, slack);
            decided = pre >= 0;
            wrong = decided && (!ok || (uint32_t)pre != x64);
            xs = ok ? x64 : 0;
        }
    }
    const unsigned long long c0 = __popcll(__ballot(valid)), c1 = __popcll(__ballot(decided)), c2 = __popcll(__ballot(wrong)), c3 = __popcll(__ballot(fell));
    if (xs) atomicAdd(&counts[4], (unsigned long long)xs);
    if ((threadIdx.x & 63) == 0) {
        if (c0) atomicAdd(&counts[0], c0);
        if (c1) atomicAdd(&counts[1], c1);
        if (c2) atomicAdd(&counts[2], c2);
        if (c3) atomicAdd(&counts[3], c3);
    }
}

} // namespace mmg
