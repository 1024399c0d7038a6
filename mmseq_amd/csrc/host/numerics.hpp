// numerics.hpp -- host-side numerics of the posterior summary (the reference reaches these through
// GSL / its bundled sokal.cc):
//   sokal_iact()   Sokal's adaptive-window integrated autocorrelation time via two FFTs,
//                  same outputs and return codes as src/sokal.cc:33-87 (used at src/mmseq.cpp:1316)
//   digamma(), trigamma()   gsl_sf_psi / gsl_sf_psi_n(1,.)   (src/mmseq.cpp:1372-1373)
//   probit()                gsl_cdf_ugaussian_Pinv            (src/mmseq.cpp:1250, :1286)
#pragma once
#include <cmath>
#include <vector>

namespace mmnum {

// in-place iterative radix-2 complex FFT (forward, e^{-i...}); n must be a power of two
inline void fft_pow2(std::vector<double> &re, std::vector<double> &im)
{
    const size_t n = re.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * M_PI / (double)len;
        const size_t half = len / 2;
        for (size_t j = 0; j < half; ++j) {
            const double wr = std::cos(ang * (double)j), wi = std::sin(ang * (double)j);
            for (size_t i = j; i < n; i += len) {
                const size_t b = i + half;
                const double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                re[b] = re[i] - xr; im[b] = im[i] - xi;
                re[i] += xr; im[i] += xi;
            }
        }
    }
}

// Returns 0 on success; 100 = too long, 200 = n < 4, 201 = n not a power of two (src/sokal.cc:36-39, :108, :119-126).
// var = circular autocovariance at lag 0 / (n (n-1)); tau = 2 (sum + (m-1)/6) with the window m chosen where
// the running sum of (rho_i - 1/6), started at -1/3, first goes negative (src/sokal.cc:63-84).
inline int sokal_iact(const double *x, int n, double *var, double *tau, int *m)
{
    if (n > (2 << 20)) return 100;
    if (n < 4) return 200;
    for (int t = n; t > 1; t >>= 1)
        if (t & 1) return 201;
    std::vector<double> re(x, x + n), im((size_t)n, 0.0);
    fft_pow2(re, im);
    for (int i = 0; i < n; ++i) { re[i] = re[i] * re[i] + im[i] * im[i]; im[i] = 0.0; }
    re[0] = 0.0; // removes the mean
    fft_pow2(re, im);
    *var = re[0] / ((double)n * (double)(n - 1));
    const double c = 1.0 / re[0];
    double sum = -0.333333333333333333333;
    *m = n + 1;
    for (int i = 0; i < n; ++i) {
        sum += re[i] * c - 0.166666666666666666666;
        if (sum < 0) { *m = i + 1; break; }
    }
    *tau = 2 * (sum + (*m - 1.0) / 6.0);
    return 0;
}

// digamma: recurrence up to x >= 12, then the asymptotic series
inline double digamma(double x)
{
    double r = 0.0;
    while (x < 12.0) { r -= 1.0 / x; x += 1.0; }
    const double f = 1.0 / (x * x);
    return r + std::log(x) - 0.5 / x -
           f * (1.0 / 12.0 - f * (1.0 / 120.0 - f * (1.0 / 252.0 - f * (1.0 / 240.0 - f * (1.0 / 132.0 - f * (691.0 / 32760.0 - f / 12.0))))));
}

// trigamma: recurrence up to x >= 12, then the asymptotic series
inline double trigamma(double x)
{
    double r = 0.0;
    while (x < 12.0) { r += 1.0 / (x * x); x += 1.0; }
    const double f = 1.0 / (x * x);
    return r + 1.0 / x + 0.5 * f +
           (1.0 / x) * f * (1.0 / 6.0 - f * (1.0 / 30.0 - f * (1.0 / 42.0 - f * (1.0 / 30.0 - f * (5.0 / 66.0 - f * (691.0 / 2730.0 - f * 7.0 / 6.0))))));
}

// inverse standard normal CDF: Wichura (1988) algorithm AS 241, PPND16 (about 1e-16 relative)
inline double probit(double p)
{
    const double q = p - 0.5;
    if (std::fabs(q) <= 0.425) {
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
    if (r <= 0.0) return q < 0 ? -HUGE_VAL : HUGE_VAL;
    r = std::sqrt(-std::log(r));
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

} // namespace mmnum
