// misc.hip -- device TU: K2, trace read-out, start values, synthetic generator, self tests (kernels in misc_kernels.h)
#include "misc_kernels.h"
#include "mmg_launch.h"
#include <hip/hip_ext.h>

namespace mmg {

void launch_update(const UpdateArgs &a, hipStream_t s, hipEvent_t start, hipEvent_t stop)
{
    if (start || stop) hipExtLaunchKernelGGL(k_update, dim3((a.n + 255u) / 256u, a.n_chains), dim3(256), 0, s, start, stop, 0, a);
    else hipLaunchKernelGGL(k_update, dim3((a.n + 255u) / 256u, a.n_chains), dim3(256), 0, s, a);
}

void launch_transpose(const double *in, double *out, uint32_t n, uint32_t S, const uint32_t *int_of_ext, hipStream_t s)
{
    const dim3 g((n + 31) / 32, (S + 31) / 32);
    hipLaunchKernelGGL(k_transpose, g, dim3(256), 0, s, in, out, n, S, int_of_ext);
}

void launch_gather_rows(const void *in, void *out, uint32_t n, uint32_t rows, int elem_bytes, const uint32_t *int_of_ext, hipStream_t s)
{
    const unsigned g = (unsigned)(((uint64_t)n * rows + 255) / 256);
    if (elem_bytes == 8) hipLaunchKernelGGL(k_gather_rows<uint64_t>, dim3(g), dim3(256), 0, s, (const uint64_t *)in, (uint64_t *)out, n, rows, int_of_ext);
    else hipLaunchKernelGGL(k_gather_rows<uint32_t>, dim3(g), dim3(256), 0, s, (const uint32_t *)in, (uint32_t *)out, n, rows, int_of_ext);
}

// cnt[i] += sum over the replicas r >= 1 of cnt[r * stride + i], which are zeroed: after mmg_sampler_sample the public count vector
// (replica 0: mmg_sampler_counts_devptr, the buffer a read-shard all-reduce works on) holds the device's column sums
__global__ __launch_bounds__(256) void k_fold_counts(int32_t *__restrict__ cnt, uint64_t stride, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t x = cnt[i];
#pragma unroll
    for (uint32_t r = 1; r < CNT_REPLICAS; ++r) { x += cnt[(uint64_t)r * stride + i]; cnt[(uint64_t)r * stride + i] = 0; }
    cnt[i] = x;
}
void launch_fold_counts(int32_t *cnt, uint64_t stride, size_t n, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_fold_counts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, cnt, stride, n);
}

// dst[i] += src[i]: the count exchange of the Gibbs self test that runs read shards side by side on ONE device
__global__ __launch_bounds__(256) void k_add_i32(int32_t *__restrict__ dst, const int32_t *__restrict__ src, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}
void launch_add_i32(int32_t *dst, const int32_t *src, size_t n, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_add_i32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, src, n);
}

void launch_start_values(bool idx64, const void *row_ptr, const uint32_t *col, const uint32_t *k, uint64_t m, uint32_t n,
                         uint64_t *acc3, int32_t *unique_hits, hipStream_t s)
{
    if (!m) return;
    const unsigned g = (unsigned)((m + SV_ROWS - 1) / SV_ROWS);
    if (idx64) hipLaunchKernelGGL(k_start_values<uint64_t>, dim3(g), dim3(256), 0, s, (const uint64_t *)row_ptr, col, k, m, n, acc3, unique_hits);
    else hipLaunchKernelGGL(k_start_values<uint32_t>, dim3(g), dim3(256), 0, s, (const uint32_t *)row_ptr, col, k, m, n, acc3, unique_hits);
}

void launch_synth_len(const SynthArgs &a, double, uint32_t *lens, hipStream_t s)
{
    hipLaunchKernelGGL(k_synth_len, dim3((unsigned)((a.rows + 255) / 256)), dim3(256), 0, s, a, lens);
}

void launch_synth_fill(const SynthArgs &a, double far_fraction, const uint64_t *row_ptr, uint32_t *col, hipStream_t s)
{
    hipLaunchKernelGGL(k_synth_fill, dim3((unsigned)((a.rows + 255) / 256)), dim3(256), 0, s, a, far_fraction, row_ptr, col);
}

void launch_selftest_math(int64_t n, const double *x, double *ol, double *oe, double *os, double *orc, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_selftest_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, x, ol, oe, os, orc);
}
void launch_selftest_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_selftest_philox, dim3(1), dim3(1), 0, s, ctr, key, out);
}
void launch_selftest_gamma(uint64_t seed, double shape, double scale, int64_t n, double *out, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_selftest_gamma, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed, shape, scale, n, out);
}
void launch_selftest_binomial(uint64_t seed, uint32_t nn, double p, int64_t n, uint32_t *out, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_selftest_binomial, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed, nn, p, n, out);
}

// ---- host instantiations of the same inline code
void host_math(int64_t n, const double *x, double *ol, double *oe, double *os, double *orc)
{
    for (int64_t i = 0; i < n; ++i) {
        ol[i] = (x[i] >= 0x1p-1022 && x[i] < __builtin_huge_val()) ? dlog_pn(x[i]) : dlog(x[i]); // as the device self test
        oe[i] = dexp(x[i]); os[i] = dsqrt(x[i]); orc[i] = 1.0 / x[i];
    }
}
void host_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out)
{
    const U4 r = philox4x32_10(U4{ctr[0], ctr[1], ctr[2], ctr[3]}, key[0], key[1]);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
    uint32_t a = ctr[0], b = ctr[1];
    philox2x32_10(a, b, key[0]);
    out[4] = a; out[5] = b;
}
// by_iter == 0: out[i] from stream (seed, tag, id0 + i, iteration 0); by_iter != 0: from stream (seed, tag, id0, iteration i)
void host_gamma(uint64_t seed, uint32_t tag, uint64_t id0, int by_iter, double shape, double scale, int64_t n, double *out)
{
    for (int64_t i = 0; i < n; ++i) {
        Stream s(seed, 0, tag, by_iter ? id0 : id0 + (uint64_t)i, by_iter ? (uint32_t)i : 0u);
        out[i] = gamma_unit(s, shape) * scale;
    }
}
void host_binomial(uint64_t seed, uint32_t nn, double p, int64_t n, uint32_t *out)
{
    for (int64_t i = 0; i < n; ++i) { Stream2 q(seed, 0, TAG_ROW, (uint64_t)i, 0); out[i] = binomial(q, nn, p); }
}

// transcript tables of the synthetic generator
void host_synth_tables(uint64_t seed, uint32_t T, double lambda, std::vector<double> &efflen, std::vector<double> &cdf,
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

// (a2 * 2^64 + a1 * 2^32 + a0) * 2^-52 / l, the integer converted to double with round-to-nearest-even
double host_start_value(uint64_t a0, uint64_t a1, uint64_t a2, double l)
{
    unsigned __int128 v = ((unsigned __int128)a2 << 64) + ((unsigned __int128)a1 << 32) + (unsigned __int128)a0;
    if (v == 0) return 0.0 / l;
    int top = 127;
    while (!((v >> top) & 1)) --top;
    double d;
    if (top <= 52) {
        d = (double)(uint64_t)v; // exact
    } else {
        const int sh = top - 52;
        uint64_t mant = (uint64_t)(v >> sh);                       // 53 bits
        const unsigned __int128 rest = v & (((unsigned __int128)1 << sh) - 1), half = (unsigned __int128)1 << (sh - 1);
        if (rest > half || (rest == half && (mant & 1))) ++mant;    // may become 2^53: still exact as a double
        d = dscalbn((double)mant, sh);
    }
    return d * 0x1p-52 / l;
}

void launch_selftest_btrs_pretest(uint64_t seed, int64_t n_cases, double n_lo, double n_hi, unsigned long long *counts, hipStream_t s)
{
    if (n_cases <= 0) return;
    hipLaunchKernelGGL(k_selftest_btrs_pretest, dim3((unsigned)((n_cases + 255) / 256)), dim3(256), 0, s, seed, n_cases, n_lo, n_hi, counts);
}

void launch_selftest_binv_pretest(uint64_t seed, int64_t n_cases, double n_lo, double n_hi, float slack, unsigned long long *counts, hipStream_t s)
{
    if (n_cases <= 0) return;
    hipLaunchKernelGGL(k_selftest_binv_pretest, dim3((unsigned)((n_cases + 255) / 256)), dim3(256), 0, s, seed, n_cases, n_lo, n_hi, slack, counts);
}

} // namespace mmg
