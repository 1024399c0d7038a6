// mmg_math.h -- counter-based randomness and reproducible fp64 elementary functions
// shared by every kernel of libmmgibbs (and by the host-side table builders).
//
// Everything here is built from IEEE-754 +,-,*,/,sqrt and integer bit operations, each
// rounded once (the library is compiled with -ffp-contract=off), so the host and the
// gfx950 instantiations produce the same bits and a chain is a pure function of
// (seed, chain, iteration, row | transcript) -- independent of launch geometry, thread
// count or number of GPUs.  That replaces the reference's thread-count-dependent
// "one MT19937 per OpenMP thread" (src/mmseq.cpp:834-838).
//
// Samplers restate the published algorithms the reference reaches through GSL:
// gsl_ran_gamma = Marsaglia & Tsang (2000) (src/mmseq.cpp:907), gsl_ran_multinomial =
// conditional binomials (src/mmseq.cpp:880).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MMG_HD __host__ __device__ __forceinline__

namespace mmg {

// ------------------------------------------------------------------ Philox4x32-10
struct U4 { uint32_t x, y, z, w; };

MMG_HD uint32_t mulhi32(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

MMG_HD U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
#if defined(__HIP_DEVICE_COMPILE__)
        // one v_mad_u64_u32 yields both halves of a product (32-bit multiplies are quarter-rate: the Gamma redraw is mostly these)
        uint64_t p0, p1;
        asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p0) : "v"(c.x), "s"(0xD2511F53u) : "vcc");
        asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p1) : "v"(c.z), "s"(0xCD9E8D57u) : "vcc");
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
#else
        const uint32_t hi0 = mulhi32(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = mulhi32(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
#endif
        U4 n;
        n.x = hi1 ^ c.y ^ k0;
        n.y = lo1;
        n.z = hi0 ^ c.w ^ k1;
        n.w = lo0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// Philox2x32-10: same family, 64-bit output, half the multiplies.  One call = one uniform; used for
// the per-row allocation draw (one 52-bit uniform per read), the dominant integer work of a sweep.
MMG_HD void philox2x32_10(uint32_t &c0, uint32_t &c1, uint32_t k)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
#if defined(__HIP_DEVICE_COMPILE__)
        // one v_mad_u64_u32 yields both halves: 32-bit multiplies are quarter-rate, and this is the hot integer path
        uint64_t pr;
        asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(pr) : "v"(c0), "s"(0xD256D193u) : "vcc");
        const uint32_t hi = (uint32_t)(pr >> 32), lo = (uint32_t)pr;
        c0 = __builtin_amdgcn_bitop3_b32(hi, c1, k, 0x96); // hi ^ c1 ^ k as ONE instruction (gfx950): a VALU slot per round less
#else
        const uint32_t hi = mulhi32(0xD256D193u, c0), lo = 0xD256D193u * c0;
        c0 = hi ^ k ^ c1;
#endif
        c1 = lo;
        k += 0x9E3779B9u;
    }
}

// 52 random bits -> uniform strictly inside (0,1); every step exact
MMG_HD double double_of(uint64_t u);
MMG_HD double u52(uint32_t a, uint32_t b)
{
    const uint64_t v = ((uint64_t)(a >> 6) << 26) | (uint64_t)(b >> 6);
#if defined(__HIP_DEVICE_COMPILE__)
    // same value without the 64-bit integer -> double conversion: v as the mantissa of a double in [1,2) is 1 + v 2^-52;
    // subtracting 1 is exact, and v 2^-52 + 2^-53 = (2v+1) 2^-53 is representable, so both additions are exact
    return (double_of(0x3ff0000000000000ull | v) - 1.0) + 0x1p-53;
#else
    return ((double)v + 0.5) * 0x1p-52;
#endif
}

enum : uint32_t { TAG_ROW = 1, TAG_GAMMA = 2, TAG_SYNTH_ROW = 3, TAG_SYNTH_TX = 4, TAG_SIMU = 5 };

// A stream = (key from seed/chain/tag, counter words id_lo,id_hi,iter) + running block index.
struct Stream {
    uint32_t k0, k1, c0, c1, c2, c3;
    MMG_HD Stream(uint64_t seed, uint32_t chain, uint32_t tag, uint64_t id, uint32_t iter)
        : k0((uint32_t)seed), k1((uint32_t)(seed >> 32) ^ (chain & 0x00FFFFFFu) ^ (tag << 24)),
          c0((uint32_t)id), c1((uint32_t)(id >> 32)), c2(iter), c3(0) {}
    // one Philox block = one pair of uniforms
    MMG_HD void pair(double &ua, double &ub)
    {
        const U4 r = philox4x32_10(U4{c0, c1, c2, c3}, k0, k1);
        ++c3;
        ua = u52(r.x, r.y);
        ub = u52(r.z, r.w);
    }
};

// 32 random bits -> uniform strictly inside (0,1): (x + 1/2) 2^-32, exact.  The reference draws its allocations from MT19937,
// i.e. with the same 32-bit resolution (gsl_rng_uniform of gsl_rng_mt19937, src/mmseq.cpp:836, :880).
MMG_HD double u32_unit(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // x as the top mantissa bits of a double in [1,2) is 1 + x 2^-32; subtracting 1 and adding 2^-33 are both exact
    return (double_of(0x3ff0000000000000ull | ((uint64_t)x << 20)) - 1.0) + 0x1p-33;
#else
    return ((double)x + 0.5) * 0x1p-32;
#endif
}

// The target of a categorical draw over weights of total t from a 32-bit random word x: (x + 1/2) 2^-32 * t, as ONE fused
// multiply-add of the exact product terms ts = t 2^-32 and hs = t 2^-33 (both exact unless they underflow): fma(x, ts, hs) rounds
// the real number (x + 1/2) 2^-32 t once, like u32_unit(x) * t does -- in two instructions (conversion, fma) instead of five.
// Host (libm fma), gfx950 (v_fma_f64) and the oracle agree bit for bit.
MMG_HD double draw_target(uint32_t x, double ts, double hs)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __fma_rn((double)x, ts, hs);
#else
    return __builtin_fma((double)x, ts, hs);
#endif
}

// Row stream.  One Philox2x32-10 block serves the TWO rows 2q and 2q+1: key from (seed, chain, tag, q >> 32), counter
// (q & 0xffffffff, iteration); row id r = 2q + h takes output word h; the b-th uniform of a row uses key + b * 0xBB67AE85.
// The per-row allocation draw is the bulk of a sweep's integer work (10 quarter-rate multiplies per block): a wave of
// k_sample_sell computes one block per lane for TWO tiles of 64 rows and hands the words out with ds_bpermute.
MMG_HD uint32_t stream2_key(uint64_t seed, uint32_t chain, uint32_t tag, uint32_t q_hi)
{
    return (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B1u) ^ (chain * 0x85EBCA6Bu) ^ (tag << 28) ^ (q_hi * 0xC2B2AE35u);
}
struct Stream2 {
    uint32_t key, c0, c1, blk, half;
    MMG_HD Stream2(uint64_t seed, uint32_t chain, uint32_t tag, uint64_t id, uint32_t iter)
        : key(stream2_key(seed, chain, tag, (uint32_t)(id >> 33))), c0((uint32_t)(id >> 1)), c1(iter), blk(0), half((uint32_t)id & 1u) {}
    MMG_HD uint32_t next_word()
    {
        uint32_t a = c0, b = c1;
        philox2x32_10(a, b, key + blk * 0xBB67AE85u);
        ++blk;
        return half ? b : a;
    }
    MMG_HD double next() { return u32_unit(next_word()); }
};

// one-uniform-at-a-time view of a stream (first of each pair, then second)
struct SeqStream {
    Stream s;
    double spare;
    bool have;
    MMG_HD SeqStream(const Stream &st) : s(st), spare(0.0), have(false) {}
    MMG_HD double next()
    {
        if (have) { have = false; return spare; }
        double ua, ub;
        s.pair(ua, ub);
        spare = ub; have = true;
        return ua;
    }
};

// ------------------------------------------------------------------ bit casts
MMG_HD uint64_t bits_of(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t)__double_as_longlong(x);
#else
    uint64_t u; __builtin_memcpy(&u, &x, 8); return u;
#endif
}
MMG_HD double double_of(uint64_t u)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)u);
#else
    double x; __builtin_memcpy(&x, &u, 8); return x;
#endif
}

MMG_HD double dsqrt(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __dsqrt_rn(x);
#else
    return __builtin_sqrt(x);
#endif
}

// ------------------------------------------------------------------ log / exp
// Classic fdlibm scheme: x = 2^k (1+f), log(1+f) = f - f^2/2 + s*(f^2/2 + R(s^2)), s = f/(2+f).
MMG_HD double dlog(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t ix = bits_of(x);
    uint32_t hx = (uint32_t)(ix >> 32);
    int k = 0;
    if (hx < 0x00100000u || (hx >> 31)) {
        if ((ix << 1) == 0) return -__builtin_huge_val();
        if (hx >> 31) return __builtin_nan("");
        k -= 54;
        x *= 0x1p54;
        ix = bits_of(x);
        hx = (uint32_t)(ix >> 32);
    } else if (hx >= 0x7ff00000u) {
        return x;
    } else if (hx == 0x3ff00000u && (ix << 32) == 0) {
        return 0.0;
    }
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    x = double_of(((uint64_t)hx << 32) | (ix & 0xffffffffull));
    const double f = x - 1.0;
    const double hfsq = 0.5 * f * f;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double dk = (double)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

// dlog for a positive, normal, finite argument: the same arithmetic without the special cases (for x == 1.0 every term is +0.0, like
// the early return) -- straight-line code the scheduler can interleave with independent work
MMG_HD double dlog_pn(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    const uint64_t ix = bits_of(x);
    uint32_t hx = (uint32_t)(ix >> 32);
    hx += 0x3ff00000u - 0x3fe6a09eu;
    const int k = (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    x = double_of(((uint64_t)hx << 32) | (ix & 0xffffffffull));
    const double f = x - 1.0;
    const double hfsq = 0.5 * f * f;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double dk = (double)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

MMG_HD double dscalbn(double y, int n)
{
    if (n > 1023) {
        y *= 0x1p1023; n -= 1023;
        if (n > 1023) { y *= 0x1p1023; n -= 1023; if (n > 1023) n = 1023; }
    } else if (n < -1022) {
        y *= 0x1p-1022 * 0x1p53; n += 1022 - 53;
        if (n < -1022) { y *= 0x1p-1022 * 0x1p53; n += 1022 - 53; if (n < -1022) n = -1022; }
    }
    return y * double_of((uint64_t)(0x3ff + n) << 52);
}

MMG_HD double dexp(double x)
{
    const double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                 P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                 P5 = 4.13813679705723846039e-08;
    uint32_t hx = (uint32_t)(bits_of(x) >> 32);
    const int sign = (int)(hx >> 31);
    hx &= 0x7fffffffu;
    double hi, lo;
    int k;
    if (hx >= 0x4086232bu) {
        if (x != x) return x;
        if (x > 709.782712893383973096) return __builtin_huge_val();
        if (x < -745.13321910194110842) return 0.0;
    }
    if (hx > 0x3fd62e42u) {
        if (hx >= 0x3ff0a2b2u) k = (int)(invln2 * x + (sign ? -0.5 : 0.5));
        else k = 1 - sign - sign;
        hi = x - (double)k * ln2hi;
        lo = (double)k * ln2lo;
        x = hi - lo;
    } else if (hx > 0x3e300000u) {
        k = 0; hi = x; lo = 0.0;
    } else {
        return 1.0 + x;
    }
    const double xx = x * x;
    const double c = x - xx * (P1 + xx * (P2 + xx * (P3 + xx * (P4 + xx * P5))));
    const double y = 1.0 + (x * c / (2.0 - c) - lo + hi);
    if (k == 0) return y;
    return dscalbn(y, k);
}

MMG_HD double dfloor(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return floor(x);
#else
    return __builtin_floor(x);
#endif
}
// ilogb for x > 0 finite (subnormals included); host and device agree by construction
MMG_HD int dilogb(double x)
{
    const uint64_t b = bits_of(x);
    const int e = (int)((b >> 52) & 0x7ff);
    if (e) return e - 1023;
    return 63 - __builtin_clzll(b & 0xfffffffffffffull) - 1074;
}
MMG_HD double dabs(double x) { return double_of(bits_of(x) & 0x7fffffffffffffffull); }

// ------------------------------------------------------------------ samplers
// N(0,1), Marsaglia polar method; one Philox block per attempt
MMG_HD double normal(Stream &s)
{
    for (;;) {
        double ua, ub;
        s.pair(ua, ub);
        const double v1 = 2.0 * ua - 1.0, v2 = 2.0 * ub - 1.0;
        const double r2 = v1 * v1 + v2 * v2;
        if (r2 >= 1.0 || r2 == 0.0) continue;
        return v1 * dsqrt(-2.0 * dlog(r2) / r2);
    }
}

// unit-scale Gamma(a), a > 0: Marsaglia-Tsang; a < 1 via Gamma(a+1) * U^(1/a)
MMG_HD double gamma_unit(Stream &s, double a_in)
{
    const double a = (a_in < 1.0) ? a_in + 1.0 : a_in;
    const double d = a - 1.0 / 3.0;
    const double c = (1.0 / 3.0) / dsqrt(d);
    double v, x, ua;
#if defined(__HIP_DEVICE_COMPILE__)
    // The same draws, four Philox blocks at a time.  The stream is counter-based: pair j of this lane is block c3 + j whatever
    // was drawn before, so the blocks a sequential attempt MAY consume are computed side by side (four independent chains of 10
    // rounds instead of one after the other -- a wave of K2 is one long dependent chain) and the attempt is then resolved from
    // them: the first pair the polar method accepts (among the first three), the pair behind it for the acceptance test.  A wave
    // used to run the polar loop until its unluckiest lane had a normal (about four passes), then the test, then everything
    // again for the one lane in twenty the test rejects.
    for (;;) {
        const U4 r0 = philox4x32_10(U4{s.c0, s.c1, s.c2, s.c3}, s.k0, s.k1), r1 = philox4x32_10(U4{s.c0, s.c1, s.c2, s.c3 + 1u}, s.k0, s.k1),
                 r2 = philox4x32_10(U4{s.c0, s.c1, s.c2, s.c3 + 2u}, s.k0, s.k1), r3 = philox4x32_10(U4{s.c0, s.c1, s.c2, s.c3 + 3u}, s.k0, s.k1);
        const double a0 = 2.0 * u52(r0.x, r0.y) - 1.0, b0 = 2.0 * u52(r0.z, r0.w) - 1.0, q0 = a0 * a0 + b0 * b0;
        const double a1 = 2.0 * u52(r1.x, r1.y) - 1.0, b1 = 2.0 * u52(r1.z, r1.w) - 1.0, q1 = a1 * a1 + b1 * b1;
        const double a2 = 2.0 * u52(r2.x, r2.y) - 1.0, b2 = 2.0 * u52(r2.z, r2.w) - 1.0, q2 = a2 * a2 + b2 * b2;
        const bool ok0 = !(q0 >= 1.0 || q0 == 0.0), ok1 = !(q1 >= 1.0 || q1 == 0.0), ok2 = !(q2 >= 1.0 || q2 == 0.0);
        if (!(ok0 || ok1 || ok2)) { s.c3 += 3u; continue; } // (block 3 only ever serves as the pair behind block 2)
        const uint32_t j1 = ok0 ? 0u : (ok1 ? 1u : 2u);
        const double v1 = ok0 ? a0 : (ok1 ? a1 : a2), q = ok0 ? q0 : (ok1 ? q1 : q2);
        ua = ok0 ? u52(r1.x, r1.y) : (ok1 ? u52(r2.x, r2.y) : u52(r3.x, r3.y)); // the first uniform of the pair behind it
        x = v1 * dsqrt(-2.0 * dlog_pn(q) / q); // 0 < q < 1, normal (a sum of squares of multiples of 2^-52)
        v = 1.0 + c * x;
        if (v <= 0.0) { s.c3 += j1 + 1u; continue; } // the polar method goes on with the next pair
        v = v * v * v;
        s.c3 += j1 + 2u;
        const double x2 = x * x;
        if (ua < 1.0 - 0.0331 * x2 * x2) break;
        if (dlog_pn(ua) < 0.5 * x2 + d * (1.0 - v + dlog_pn(v))) break; // ua >= 2^-53; v = (1 + c x)^3 >= 2^-159
    }
#else
    for (;;) {
        do {
            x = normal(s);
            v = 1.0 + c * x;
        } while (v <= 0.0);
        v = v * v * v;
        double ub;
        s.pair(ua, ub);
        const double x2 = x * x;
        if (ua < 1.0 - 0.0331 * x2 * x2) break;
        if (dlog(ua) < 0.5 * x2 + d * (1.0 - v + dlog(v))) break;
    }
#endif
    double g = d * v;
    if (a_in < 1.0) {
        double ub;
        s.pair(ua, ub);
#if defined(__HIP_DEVICE_COMPILE__)
        g = g * dexp(dlog_pn(ua) / a_in); // ua >= 2^-53
#else
        g = g * dexp(dlog(ua) / a_in);
#endif
    }
    return g;
}

// log(k!) Stirling remainder used by the binomial rejection sampler
MMG_HD double stirling_tail(double k)
{
    // k <= 9: tabulated.  A chain of selects, not a switch: the lanes of a wave ask for different k, and a switch is one branch
    // target per case, run one after the other.
    const int i = k <= 9.0 ? (int)k : 9;
    double t = 0.0810614667953272;
    t = i >= 1 ? 0.0413406959554092 : t;
    t = i >= 2 ? 0.0276779256849983 : t;
    t = i >= 3 ? 0.02079067210376509 : t;
    t = i >= 4 ? 0.0166446911898211 : t;
    t = i >= 5 ? 0.0138761288230707 : t;
    t = i >= 6 ? 0.0118967099458917 : t;
    t = i >= 7 ? 0.0104112652619720 : t;
    t = i >= 8 ? 0.00925546218271273 : t;
    t = i >= 9 ? 0.00833056343336287 : t;
    const double kp1sq = (k + 1.0) * (k + 1.0);
    const double f = (1.0 / 12.0 - (1.0 / 360.0 - 1.0 / 1260.0 / kp1sq) / kp1sq) / (k + 1.0);
    return k <= 9.0 ? t : f;
}

// Binomial(n, p): sequential-search inversion below n*min(p,1-p) < 10, Hormann's BTRS above
template <typename Src>
MMG_HD uint32_t binomial(Src &q, uint32_t n, double p)
{
    if (n == 0 || !(p > 0.0)) return 0;
    if (p >= 1.0) return n;
    bool flip = false;
    if (p > 0.5) { p = 1.0 - p; flip = true; }
    const double dn = (double)n;
    uint32_t res;
    if (dn * p < 10.0) {
        const double qq = 1.0 - p, s = p / qq, a = (dn + 1.0) * s;
        for (;;) {
            double r = dexp(dn * dlog_pn(qq)); // 0.5 <= qq < 1
            double u = q.next();
            uint32_t x = 0;
            bool ok = true;
            while (u > r) {
                u -= r;
                x++;
                if (x > n) { ok = false; break; }
                r *= (a / (double)x - s);
            }
            if (ok) { res = x; break; }
        }
    } else {
        const double qq = 1.0 - p, spq = dsqrt(dn * p * qq);
        const double b = 1.15 + 2.53 * spq;
        const double a = -0.0873 + 0.0248 * b + 0.01 * p;
        const double c = dn * p + 0.5;
        const double vr = 0.92 - 4.2 / b;
        const double r = p / qq;
        const double alpha = (2.83 + 5.1 / b) * spq;
        const double m = dfloor((dn + 1.0) * p);
        for (;;) {
            const double u = q.next() - 0.5;
            double v = q.next();
            const double us = 0.5 - dabs(u);
            const double kf = dfloor((2.0 * a / us + b) * u + c);
            if (kf < 0.0 || kf > dn) continue;
            if (us >= 0.07 && v <= vr) { res = (uint32_t)kf; break; }
            // (every argument below is a ratio of positive finite numbers far from the subnormal range: dlog_pn)
            v = dlog_pn(v * alpha / (a / (us * us) + b));
            const double ub = (m + 0.5) * dlog_pn((m + 1.0) / (r * (dn - m + 1.0))) +
                              (dn + 1.0) * dlog_pn((dn - m + 1.0) / (dn - kf + 1.0)) +
                              (kf + 0.5) * dlog_pn(r * (dn - kf + 1.0) / (kf + 1.0)) +
                              stirling_tail(m) + stirling_tail(dn - m) - stirling_tail(kf) - stirling_tail(dn - kf);
            if (v <= ub) { res = (uint32_t)kf; break; }
        }
    }
    return flip ? n - res : res;
}

// ------------------------------------------------------------------ BTRS: the exact test, decided in fp32 where that is safe
// binomial()'s exact acceptance test of a candidate k that missed the squeeze is
//     T0 <= T1 + T2 + T3 + S,   T0 = log(v alpha / (a / us^2 + b)),
//     T1 = (m + 1/2) log((m + 1) / (r (n - m + 1))),  T2 = (n + 1) log((n - m + 1) / (n - k + 1)),  T3 = (k + 1/2) log(r (n - k + 1) / (k + 1)),
//     S  = st(m) + st(n - m) - st(k) - st(n - k)                                       (r = p / q, st = stirling_tail)
// -- four fp64 logarithms, a dozen fp64 divisions: 430 vector instructions, a quarter of k_sample_bigk's.  The test is a COMPARISON: its
// outcome is known without fp64 whenever a cheap estimate D of (T1 + T2 + T3 + S - T0) is farther from zero than a bound E on the
// estimate's own error.  btrs_pretest returns +1 (accept) / -1 (reject) in that case, 0 (undecided: run the fp64 test) otherwise, so
// the draw is the one binomial() makes, uniform for uniform.
// The estimate: the three logarithms have arguments near 1 (m is the mode), written as log1p(x_i) with the differences in the
// numerators formed in fp64 -- x1 = ((m + 1) q - p (n - m + 1)) / (p (n - m + 1)), x2 = (k - m) / (n - k + 1), x3 = (p (n - k + 1) - q (k + 1)) /
// (q (k + 1)): no cancellation is left to fp32, whose divisions (v_rcp_f32, 1 ulp), series (|x| < 1/8) and native logarithm (elsewhere:
// 1 + x loses 5e-7 of the result at most there) each err by a few 2^-24 RELATIVE to their result.  S from the asymptotic series of
// stirling_tail (m and n - m are 10 at least; k and n - k take the table's entries below 3).  First-order error of T_i: c_i |x_i| /
// min(1, 1 + x_i) times 6e-7 (c_i the coefficient; the fp64 numerators contribute n 2^-52, i.e. 2e-7 relative at n = 2^32); of T0 (native
// log2 of a ratio of fp32 values, not amplified): 1e-6 (1 + |T0|); of S: 1e-7.  E takes six times that:
//     E = sum_i c_i |x_i| / min(1, 1 + x_i) 2^-18 + (1 + |T0|) 2^-16 + 2^-14.
// mmg_selftest_btrs_pretest runs candidates over the whole range of (n, p) through both and counts decided cases that disagree
// with the fp64 test: none in 2.2 10^10 tests out of 6 10^10 attempts, the largest error 11 % of E (profiles/r06_btrs_pretest.txt;
// tests/test_gpu_parity.py::test_btrs_pretest_never_contradicts_the_exact_test).  A NaN anywhere
// compares false twice: undecided.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float mmg_rcpf(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float mmg_log1pf(float x)
{
    // |x| < 1/8: the series x (1 - x/2 + x^2/3 - ... - x^7/8) (truncation 7e-9 relative); elsewhere ln(1 + x) from the native log2
    // (1 + x rounds within 6e-8 of a result of at least 0.117 in magnitude: 5e-7 relative)
    float sacc = -1.0f / 8.0f;
    sacc = sacc * x + 1.0f / 7.0f; sacc = sacc * x - 1.0f / 6.0f; sacc = sacc * x + 1.0f / 5.0f; sacc = sacc * x - 1.0f / 4.0f;
    sacc = sacc * x + 1.0f / 3.0f; sacc = sacc * x - 1.0f / 2.0f; sacc = sacc * x + 1.0f;
    const float series = sacc * x;
    const float native = __builtin_amdgcn_logf(1.0f + x) * 0.693147180559945309f;
    return __builtin_fabsf(x) < 0.125f ? series : native;
}
// stirling_tail(k) from inv = 1 / (k + 1): the series of mmg_math.h's fp64 function (k >= 10 there), good to 4e-8 from k = 3 on ...
__device__ __forceinline__ float mmg_stirling_seriesf(float inv)
{
    const float inv2 = inv * inv;
    return (1.0f / 12.0f - (1.0f / 360.0f - (1.0f / 1260.0f) * inv2) * inv2) * inv;
}
// ... and the table's first three entries below (k1 = k + 1)
__device__ __forceinline__ float mmg_stirling_tailf(float k1, float inv)
{
    const float t = k1 < 1.5f ? 0.0810614667953272f : (k1 < 2.5f ? 0.0413406959554092f : 0.0276779256849983f);
    return k1 < 3.5f ? t : mmg_stirling_seriesf(inv);
}
// d: the estimate of T1 + T2 + T3 + S - T0, e: the bound on its error
__device__ __forceinline__ void btrs_estimate(double dn, double p, double kf, double us, double vv, double a, double b, double spq, float &d_out, float &e_out)
{
    const double q = 1.0 - p, m1 = dfloor((dn + 1.0) * p) + 1.0, kf1 = kf + 1.0;
    const double nm1 = dn + 2.0 - m1, nk1 = dn - kf + 1.0;                                  // n - m + 1, n - k + 1: exact
    const double pnm = p * nm1, qk = q * kf1;
    const double num1 = m1 * q - pnm, num3 = p * nk1 - qk, num2 = kf1 - m1;
    const float fm1 = (float)m1, fk1 = (float)kf1, fnm1 = (float)nm1, fnk1 = (float)nk1;   // m + 1 >= 11 and n - m + 1 >= 11 (n p >= 10, p <= 1/2)
    const float ink1 = mmg_rcpf(fnk1);
    const float x1 = (float)num1 * mmg_rcpf((float)pnm), x2 = (float)num2 * ink1, x3 = (float)num3 * mmg_rcpf((float)qk);
    const float c1 = fm1 - 0.5f, c2 = (float)dn + 1.0f, c3 = fk1 - 0.5f;
    const float t1 = c1 * mmg_log1pf(x1), t2 = c2 * mmg_log1pf(x2), t3 = c3 * mmg_log1pf(x3);
    const float fb = (float)b, fus = (float)us, fspq = (float)spq;
    const float alpha = (2.83f + 5.1f * mmg_rcpf(fb)) * fspq;
    const float arg = (float)vv * alpha * mmg_rcpf((float)a * mmg_rcpf(fus * fus) + fb);
    const float t0 = __builtin_amdgcn_logf(arg) * 0.693147180559945309f;
    const float sfix = mmg_stirling_seriesf(mmg_rcpf(fm1)) + mmg_stirling_seriesf(mmg_rcpf(fnm1)) - mmg_stirling_tailf(fk1, mmg_rcpf(fk1)) - mmg_stirling_tailf(fnk1, ink1);
    const float d = ((t1 + t2) + t3) + sfix - t0;
    auto amp = [](float c, float x) { return c * __builtin_fabsf(x) * mmg_rcpf(__builtin_fminf(1.0f, 1.0f + x)); };
    const float e = (amp(c1, x1) + amp(c2, x2) + amp(c3, x3)) * 0x1p-18f + (1.0f + __builtin_fabsf(t0)) * 0x1p-16f + 0x1p-14f;
    d_out = d; e_out = e;
}
__device__ __forceinline__ int btrs_pretest(double dn, double p, double kf, double us, double vv, double a, double b, double spq)
{
    float d, e;
    btrs_estimate(dn, p, kf, us, vv, a, b, spq, d, e);
    return d > e ? 1 : (d < -e ? -1 : 0);
}
#else
// (the host pass of a device translation unit only needs the name: the estimate runs on the device, the oracle and the host instantiation
// of binomial() always take the fp64 test)
__device__ __forceinline__ int btrs_pretest(double, double, double, double, double, double, double, double) { return 0; }
__device__ __forceinline__ void btrs_estimate(double, double, double, double, double, double, double, double, float &d, float &e) { d = 0.0f; e = 0.0f; }
#endif
// the fp64 test itself (binomial()'s expressions): true = accept; diff (optional) = ub - v as computed
MMG_HD bool btrs_exact_test(double dn, double p, double kf, double us, double vv, double a, double b, double spq, double *diff = nullptr)
{
    const double r = p / (1.0 - p), alpha = (2.83 + 5.1 / b) * spq, m = dfloor((dn + 1.0) * p);
    const double v = dlog_pn(vv * alpha / (a / (us * us) + b));
    const double ub = (m + 0.5) * dlog_pn((m + 1.0) / (r * (dn - m + 1.0))) +
                      (dn + 1.0) * dlog_pn((dn - m + 1.0) / (dn - kf + 1.0)) +
                      (kf + 0.5) * dlog_pn(r * (dn - kf + 1.0) / (kf + 1.0)) +
                      stirling_tail(m) + stirling_tail(dn - m) - stirling_tail(kf) - stirling_tail(dn - kf);
    if (diff) *diff = ub - v;
    return v <= ub;
}

// ------------------------------------------------------------------ inversion: the search decided in fp32 where that is safe
// binomial()'s inversion finds the first x whose cumulative probability reaches the uniform: r0 = exp(n log q) (an fp64 logarithm and
// exponential), then r_x = r_(x-1) (a / x - s) with an fp64 division per term.  Again the OUTCOME is a comparison: with the terms r~ and
// their running sum c~ in fp32 and a bound e_x on the sum's error, x is known whenever  c~_(x-1) + e_(x-1) < u < c~_x - e_x.  binv_pretest
// returns that x, or -1 (undecided: a boundary closer than the bound, more than 64 terms) -- the fp64 search runs then, on the same uniform.
// Errors: r0 -- the exponent n log1p(-p) is below 20 in magnitude, fp32 product and log1pf err by 5e-7 of it, v_exp_f32 by an ulp: 1e-5
// relative; every further term multiplies by s (n + 1 - x) / x with two v_rcp_f32 and four roundings: 4e-7 more per term; the sum adds an ulp of
// itself per term; the uniform rounds to fp32 within 2^-25.  The bound takes six times that:  e_x = c~_x (2^-14 + x 2^-19) + 2^-22.
// mmg_selftest_binv_pretest counts decided cases that differ from the fp64 search: none in 4 10^10, none with e_x at a sixteenth
// (profiles/r06_binv_pretest.txt; tests/test_gpu_parity.py::test_binv_pretest_never_contradicts_the_fp64_search).
#if defined(__HIP_DEVICE_COMPILE__)
// (slack: the selftest's knob -- the bound scaled down until decided cases start to differ shows how much room it has; 1 in the sampler)
__device__ __forceinline__ int binv_pretest(double dn, double p, double u, float slack = 1.0f)
{
    const float fp = (float)p, fq = (float)(1.0 - p), fn = (float)dn, fu = (float)u;
    float r = __builtin_amdgcn_exp2f(fn * mmg_log1pf(-fp) * 1.44269504088896341f);
    const float s = fp * mmg_rcpf(fq);
    float c = r, k = 0.0f, nk = fn, rel = 0x1p-14f;
    for (int it = 0; it < 64; ++it) {
        const float e = (c * rel + 0x1p-22f) * slack;
        if (!(fu > c + e)) return fu < c - e ? (int)k : -1;
        k += 1.0f;
        r *= s * nk * mmg_rcpf(k);                                           // s (n + 1 - k) / k
        nk -= 1.0f;
        c += r;
        rel += 0x1p-19f;
    }
    return -1;
}
#else
__device__ __forceinline__ int binv_pretest(double, double, double, float = 1.0f) { return -1; }
#endif
// the fp64 search itself on one uniform (binomial()'s loop): false = the sum of the terms fell short of the uniform (binomial() starts over)
MMG_HD bool binv_exact(double dn, double p, double u, uint32_t n, uint32_t &x_out)
{
    const double qq = 1.0 - p, s = p / qq, a = (dn + 1.0) * s;
    double r = dexp(dn * dlog_pn(qq)); // 0.5 <= qq < 1
    uint32_t x = 0;
    while (u > r) {
        u -= r;
        x++;
        if (x > n) return false;
        r *= (a / (double)x - s);
    }
    x_out = x;
    return true;
}

} // namespace mmg
