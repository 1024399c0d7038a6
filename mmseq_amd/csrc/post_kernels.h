// post_kernels.h -- the posterior summary of the Gibbs trace on the device (src/mmseq.cpp:927-1008 trace aggregation,
// :1110-1192 percentiles, :1195-1227 log means, :1235-1305 proportion summaries, :1307-1363 Sokal per feature; src/sokal.cc:33-87).
//
// The trace stays where K2 wrote it (sample-major [S][n] in device numbering).  Derived traces (simulated isoforms, sums over
// identical sets and genes, proportions) are built sample-major as well -- the layout the trace writers print, row by row -- and
// transposed once to series-major for the per-series summary kernel: one workgroup per series sorts the S samples (percentiles),
// logs them, and runs Sokal's estimator with two radix-2 FFTs in LDS.  Sums run in the reference's order (members of a group in
// the given order, samples ascending), the FFT uses the butterfly order and twiddle factors of the host implementation
// (mmseq_amd/csrc/host/numerics.hpp; the table is computed on the host and uploaded), so device and host agree to the rounding of
// log() alone.
#pragma once
#include "mmg_types.h"
#include "mmg_math.h"

namespace mmg {

// simulated traces of isoforms without hits (:971-978): V[s * nv + v] = Gamma(alpha) * scale[v], keyed (seed, TAG_SIMU, id[v], s)
__global__ __launch_bounds__(256) void k_virtual_traces(uint64_t seed, double alpha, uint32_t nv, uint32_t S, const uint64_t *__restrict__ id,
                                                        const double *__restrict__ scale, double *__restrict__ V)
{
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)nv * S) return;
    const uint32_t s = (uint32_t)(gid / nv), v = (uint32_t)(gid % nv);
    Stream st(seed, 0, TAG_SIMU, id[v], s);
    V[gid] = gamma_unit(st, alpha) * scale[v];
}

// G[s * ng + g] = sum over the members of group g, in the given order, of their trace at sample s (:927-1008).
// member < n: the caller's transcript (device column int_of_ext[member]); member >= n: virtual transcript member - n.
__global__ __launch_bounds__(256) void k_group_sums(uint32_t ng, uint32_t S, uint32_t n, uint32_t nv, const uint64_t *__restrict__ ptr,
                                                    const uint32_t *__restrict__ member, const uint32_t *__restrict__ int_of_ext,
                                                    const double *__restrict__ trace, const double *__restrict__ V, double *__restrict__ G)
{
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)ng * S) return;
    const uint32_t s = (uint32_t)(gid / ng), g = (uint32_t)(gid % ng);
    double acc = 0.0;
    for (uint64_t j = ptr[g]; j < ptr[g + 1]; ++j) {
        const uint32_t m = member[j];
        acc += m < n ? trace[(uint64_t)s * n + (int_of_ext ? int_of_ext[m] : m)] : V[(uint64_t)s * nv + (m - n)];
    }
    G[gid] = acc;
}

// proportions of gene expression (:1014-1031): P[s * cnt + i] = x_i(s) / G[s * ng + gene_of[i]], i over the caller's transcripts
// (x from the trace) or over the virtual ones (x from V); gene_of == 0xffffffff: NaN (a transcript outside every gene)
__global__ __launch_bounds__(256) void k_proportions(uint32_t cnt, uint32_t S, uint32_t stride, const double *__restrict__ X,
                                                     const uint32_t *__restrict__ col_of, const uint32_t *__restrict__ gene_of, uint32_t ng,
                                                     const double *__restrict__ G, double *__restrict__ P)
{
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)cnt * S) return;
    const uint32_t s = (uint32_t)(gid / cnt), i = (uint32_t)(gid % cnt);
    const uint32_t g = gene_of[i];
    const double x = X[(uint64_t)s * stride + (col_of ? col_of[i] : i)];
    P[gid] = g == 0xffffffffu ? __builtin_nan("") : x / G[(uint64_t)s * ng + g];
}

// inverse standard normal CDF: Wichura (1988) AS 241 PPND16, the algorithm of host/numerics.hpp:probit (gsl_cdf_ugaussian_Pinv
// at src/mmseq.cpp:1250, :1286), on the library's own log / sqrt
__device__ __forceinline__ double dprobit(double p)
{
    const double q = p - 0.5;
    if (dabs(q) <= 0.425) {
        const double r = 0.180625 - q * q;
        const double num = (((((((2.5090809287301226727e3 * r + 3.3430575583588128105e4) * r + 6.7265770927008700853e4) * r +
                                4.5921953931549871457e4) * r + 1.3731693765509461125e4) * r + 1.9715909503065514427e3) * r +
                             1.3314166789178437745e2) * r + 3.3871328727963666080e0);
        const double den = (((((((5.2264952788528545610e3 * r + 2.8729085735721942674e4) * r + 3.9307895800092710610e4) * r +
                                2.1213794301586595867e4) * r + 5.3941960214247511077e3) * r + 6.8718700749205790830e2) * r +
                             4.2313330701600911252e1) * r + 1.0);
        return q * num / den;
    }
    double r = q < 0 ? p : 1.0 - p;
    if (r <= 0.0) return q < 0 ? -__builtin_huge_val() : __builtin_huge_val();
    r = dsqrt(-dlog(r));
    double val;
    if (r <= 5.0) {
        r -= 1.6;
        const double num = (((((((7.74545014278341407640e-4 * r + 2.27238449892691845833e-2) * r + 2.41780725177450611770e-1) * r +
                                1.27045825245236838258e0) * r + 3.64784832476320460504e0) * r + 5.76949722146069140550e0) * r +
                             4.63033784615654529590e0) * r + 1.42343711074968357734e0);
        const double den = (((((((1.05075007164441684324e-9 * r + 5.47593808499534494600e-4) * r + 1.51986665636164571966e-2) * r +
                                1.48103976427480074590e-1) * r + 6.89767334985100004550e-1) * r + 1.67638483018380384940e0) * r +
                             2.05319162663775882187e0) * r + 1.0);
        val = num / den;
    } else {
        r -= 5.0;
        const double num = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r + 1.24266094738807843860e-3) * r +
                                2.65321895265761230930e-2) * r + 2.96560571828504891230e-1) * r + 1.78482653991729133580e0) * r +
                             5.46378491116411436990e0) * r + 6.65790464350110377720e0);
        const double den = (((((((2.04426310338993978564e-15 * r + 1.42151175831644588870e-7) * r + 1.84631831751005468180e-5) * r +
                                7.86869131145613259100e-4) * r + 1.48753612908506148525e-2) * r + 1.36929880922735805310e-1) * r +
                             5.99832206555887937690e-1) * r + 1.0);
        val = num / den;
    }
    return q < 0 ? -val : val;
}

// order-preserving map of doubles onto unsigned integers (NaNs sort last): the sort below compares these
__device__ __forceinline__ uint64_t sort_key(double x)
{
    const uint64_t b = bits_of(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double sort_unkey(uint64_t k)
{
    return double_of((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k);
}

struct SeriesOut {
    double *log_mean, *var, *tau;   // [count]      (log mode)
    int32_t *rc;                    // [count]      Sokal return code (src/sokal.cc:36-39)
    double *pct;                    // [count][np]  percentiles of the series itself (both modes)
    double *mean, *probit_mean, *probit_sd; // [count]  (proportion mode)
};

// One workgroup per series of S samples at a time (series-major input X[series * S + s]); workgroup b takes the series b, b + gridDim.x, ...
//   LOG_MODE:  percentiles of x (:1110-1192); y = log x; mean of y (:1195-1227); Sokal var / tau of y (:1307-1363)
//   otherwise: percentiles of x; mean of x; mean and sd of probit(clamp(x)) when multi[series] (:1235-1305)
// SMAX > 0: the series is sorted and transformed in LDS (S <= SMAX; the sort keys and the real parts share their storage: the sort
// is over before the transform starts).  SMAX == 0: any S, in the workgroup's slice of ws (3 * SP * 8 bytes per workgroup, SP = S
// rounded up to a power of two) -- the same steps on global memory, for traces longer than LDS holds.
// S a power of two in [4, 2^21] for the Sokal part, else rc = 201 / 200 / 100 (src/sokal.cc:36-39).
template <int SMAX, bool LOG_MODE>
__global__ __launch_bounds__(256) void k_series_summary(uint32_t count, uint32_t S, const double *__restrict__ X, uint32_t np,
                                                        const int32_t *__restrict__ pind, const uint8_t *__restrict__ multi,
                                                        const double *__restrict__ tw /* [S] (cos, sin) pairs at tw[2 * (half + j)] */,
                                                        SeriesOut o, uint64_t *__restrict__ ws)
{
    constexpr bool IN_LDS = SMAX > 0;
    __shared__ uint64_t l_key[IN_LDS ? SMAX : 1];
    __shared__ double l_im[(IN_LDS && LOG_MODE) ? SMAX : 1];
    const uint32_t tid = threadIdx.x;
    uint32_t SP = 1;
    while (SP < S) SP <<= 1;
    uint64_t *s_key;
    double *s_re, *s_im, *s_pw;
    uint32_t cap;
    if constexpr (IN_LDS) {
        s_key = l_key; s_re = reinterpret_cast<double *>(l_key); s_im = l_im; s_pw = nullptr; cap = SMAX;
    } else {
        s_key = ws + (uint64_t)blockIdx.x * 3 * SP; s_re = reinterpret_cast<double *>(s_key + SP); s_im = s_re + SP;
        s_pw = reinterpret_cast<double *>(s_key); cap = SP;
    }
    uint32_t lg = 0;
    while ((1u << lg) < S) ++lg;
    for (uint32_t ser = blockIdx.x; ser < count; ser += gridDim.x) {
        __syncthreads();   // the previous series of this workgroup is done with the buffers
        const double *x = X + (uint64_t)ser * S;
        for (uint32_t i = tid; i < S; i += 256) s_key[i] = sort_key(x[i]);
        for (uint32_t i = S + tid; i < cap; i += 256) s_key[i] = ~0ull; // padding sorts last
        __syncthreads();
        // bitonic sort of SP = next power of two >= S keys
        for (uint32_t k = 2; k <= SP; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = tid; i < SP; i += 256) {
                    const uint32_t l = i ^ j;
                    if (l > i) {
                        const uint64_t a = s_key[i], b = s_key[l];
                        const bool up = (i & k) == 0;
                        if ((a > b) == up) { s_key[i] = b; s_key[l] = a; }
                    }
                }
                __syncthreads();
            }
        for (uint32_t q = tid; q < np; q += 256) {
            const int32_t idx = pind[q];
            o.pct[(uint64_t)ser * np + q] = (idx >= 0 && (uint32_t)idx < S) ? sort_unkey(s_key[idx]) : __builtin_nan("");
        }
        if constexpr (!LOG_MODE) {
            // sequential sums in sample order, as the reference (:1237-1262)
            if (tid == 0) {
                double sp = 0.0;
                for (uint32_t i = 0; i < S; ++i) sp += x[i];
                o.mean[ser] = sp / (double)S;
            }
            if (tid == 64) {
                const bool mm = multi[ser] != 0;
                double s1 = 0.0, s2 = 0.0;
                for (uint32_t i = 0; i < S; ++i) {
                    double z = __builtin_huge_val();
                    if (mm) {
                        double p = x[i];
                        p = p < 0.000000001 ? 0.000000001 : p;   // std::min(std::max(p, 1e-9), 1 - 1e-9) (:1250); a NaN stays a NaN
                        p = 0.999999999 < p ? 0.999999999 : p;
                        z = dprobit(p);
                    }
                    s1 += z;
                    s2 += z * z;
                }
                o.probit_mean[ser] = s1 / (double)S;
                o.probit_sd[ser] = dsqrt((s2 - s1 * s1 / (double)S) / ((double)S - 1.0));
            }
        } else {
            // ---- log mode
            __syncthreads();   // the percentiles are out: the keys' storage becomes the real parts
            int rc = 0;
            if (S > (2u << 20)) rc = 100;
            else if (S < 4) rc = 200;
            else if (S & (S - 1)) rc = 201;
            for (uint32_t i = tid; i < S; i += 256) s_im[i] = dlog(x[i]); // natural order, for the mean
            __syncthreads();
            if (tid == 0) {
                double acc = 0.0;
                for (uint32_t i = 0; i < S; ++i) acc += s_im[i];
                o.log_mean[ser] = acc / (double)S;
            }
            if (rc != 0) {
                if (tid == 0) { o.rc[ser] = rc; o.var[ser] = 0.0; o.tau[ser] = 0.0; }
                continue;
            }
            // y into the transform buffers in bit-reversed order (the host permutes after loading; same values)
            for (uint32_t i = tid; i < S; i += 256) s_re[__brev(i) >> (32 - lg)] = s_im[i];
            __syncthreads();
            for (uint32_t i = tid; i < S; i += 256) s_im[i] = 0.0;
            __syncthreads();
            auto fft = [&]() { // in-place radix-2 DIT on bit-reversed input: the butterflies of host/numerics.hpp:fft_pow2
                for (uint32_t len = 2; len <= S; len <<= 1) {
                    const uint32_t half = len >> 1;
                    for (uint32_t b = tid; b < (S >> 1); b += 256) {
                        const uint32_t j = b & (half - 1), i = ((b / half) * len) + j, q = i + half;
                        const double wr = tw[2 * (half + j)], wi = tw[2 * (half + j) + 1];
                        const double xr = s_re[q] * wr - s_im[q] * wi, xi = s_re[q] * wi + s_im[q] * wr;
                        const double ar = s_re[i], ai = s_im[i];
                        s_re[q] = ar - xr; s_im[q] = ai - xi;
                        s_re[i] = ar + xr; s_im[i] = ai + xi;
                    }
                    __syncthreads();
                }
            };
            fft();
            // power spectrum, mean removed, back into bit-reversed order for the second transform
            if constexpr (IN_LDS) {
                double pw[(SMAX + 255) / 256];
                for (uint32_t i = tid, c = 0; i < S; i += 256, ++c) pw[c] = i == 0 ? 0.0 : s_re[i] * s_re[i] + s_im[i] * s_im[i];
                __syncthreads();
                for (uint32_t i = tid, c = 0; i < S; i += 256, ++c) { s_re[__brev(i) >> (32 - lg)] = pw[c]; s_im[i] = 0.0; }
            } else {
                for (uint32_t i = tid; i < S; i += 256) s_pw[i] = i == 0 ? 0.0 : s_re[i] * s_re[i] + s_im[i] * s_im[i];
                __syncthreads();
                for (uint32_t i = tid; i < S; i += 256) { s_re[__brev(i) >> (32 - lg)] = s_pw[i]; s_im[i] = 0.0; }
            }
            __syncthreads();
            fft();
            if (tid == 0) {
                const double n = (double)S;
                const double r0 = s_re[0];
                o.var[ser] = r0 / (n * (n - 1.0));
                const double c = 1.0 / r0;
                double sum = -0.333333333333333333333;
                int m = (int)S + 1;
                for (uint32_t i = 0; i < S; ++i) {
                    sum += s_re[i] * c - 0.166666666666666666666;
                    if (sum < 0) { m = (int)i + 1; break; }
                }
                o.tau[ser] = 2 * (sum + ((double)m - 1.0) / 6.0);
                o.rc[ser] = 0;
            }
        }
    }
}

} // namespace mmg
