/*
 * oracle/mmseq_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, libm + OpenMP only) of the Gibbs hot path of
 * eturro/mmseq, written from the reference's behaviour, used ONLY by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker /
 * timed CPU baseline.  Nothing under mmseq_amd/ includes, links or calls it.
 *
 * Reference lines restated (all relative to /root/reference):
 *   src/mmseq.cpp:617-638   start values mu0 and the unique-hit histogram
 *   src/mmseq.cpp:741-811   EM fixed point (Gibbs start, log_mu_em)
 *   src/mmseq.cpp:851-918   the Gibbs loop: per-row multinomial allocation (:865-891),
 *                           column sums (:887,:896-899), Gamma redraw (:905-908),
 *                           thinned trace capture (:911-917)
 *   src/mmseq.cpp:834-838   one MT19937 per OpenMP thread seeded seed+tid
 *   src/uh.cpp:3-26         unique hits to transcript groups
 *   src/sokal.cc:33-87      Sokal IACT estimator (checked against the compiled
 *                           reference object oracle/_ref/libsokal_ref.so)
 *
 * PARITY STATUS: the reference ships no tests/golden vectors for this path and its
 * arithmetic lives in GSL (not in the tree, not in this image; version unpinned:
 * src/Makefile:16 links distro -lgsl), so everything except sokal() is
 * "parity unpinned" against the reference binary.  What pins this file instead:
 * Random123 / MT19937 known-answer vectors, analytic posteriors, scipy
 * distributions, and the compiled reference sokal.cc (tests/test_oracle_*.py).
 *
 * Two Gibbs engines live here:
 *  (1) orc_gibbs_ref   -- reference-STRUCTURED: static row partition over OpenMP
 *      threads, one MT19937 per thread seeded seed+tid, multinomial by conditional
 *      binomials (the published gsl_ran_multinomial algorithm), per-thread count
 *      slabs zeroed and reduced every iteration, Marsaglia-Tsang gamma (the
 *      published gsl_ran_gamma algorithm).  This is the timed CPU baseline
 *      ("port") and the statistical oracle.
 *  (2) orc_gibbs_keyed -- the same Gibbs update, but every random draw comes from a
 *      counter-based Philox4x32-10 stream keyed by (seed, chain, iteration, row or
 *      transcript), and every transcendental is the fdlibm-style routine below, so
 *      the result does not depend on thread count / geometry and a GPU
 *      implementation of the same spec must match it BIT FOR BIT (integer counts
 *      and fp64 mu alike).  This is the parity oracle for the HIP kernels.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_K_SMALL 64u /* no row draws more categoricals than this */
#define ORC_K_DRAWS_PER_HIT 16u /* ... nor more than this * (hits - 1); above either: binomial chain (spec version 8: the smaller of the two) */
static inline int orc_draws_categoricals(uint32_t k, uint32_t L) /* mmg_types.h: draws_categoricals */
{
    const uint64_t per_hit = (uint64_t)ORC_K_DRAWS_PER_HIT * (L - 1u);
    return k <= 1u || (uint64_t)k <= (per_hit < ORC_K_SMALL ? per_hit : (uint64_t)ORC_K_SMALL);
}

/* ------------------------------------------------------------------------- */
/* Philox4x32-10 (Salmon et al., SC'11; Random123 reference constants)        */
/* ------------------------------------------------------------------------- */
typedef struct { uint32_t v[4]; } orc_u4;

static inline orc_u4 philox4x32_10(orc_u4 c, uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c.v[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c.v[2];
        orc_u4 n;
        n.v[0] = (uint32_t)(p1 >> 32) ^ c.v[1] ^ k0;
        n.v[1] = (uint32_t)p1;
        n.v[2] = (uint32_t)(p0 >> 32) ^ c.v[3] ^ k1;
        n.v[3] = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    orc_u4 c = {{ctr[0], ctr[1], ctr[2], ctr[3]}};
    c = philox4x32_10(c, key[0], key[1]);
    memcpy(out, c.v, sizeof c.v);
}

/* Philox2x32-10 (same family, 64-bit output): one call = one uniform.  Used for the per-row allocation
 * draws, which need a single 52-bit uniform per read and dominate the sweep's integer work. */
static inline void philox2x32_10(uint32_t *c0, uint32_t *c1, uint32_t k)
{
    uint32_t a = *c0, b = *c1;
    for (int r = 0; r < 10; ++r) {
        uint64_t p = (uint64_t)0xD256D193u * a;
        uint32_t na = (uint32_t)(p >> 32) ^ k ^ b;
        b = (uint32_t)p;
        a = na;
        k += 0x9E3779B9u;
    }
    *c0 = a; *c1 = b;
}

void orc_philox2x32(const uint32_t ctr[2], uint32_t key, uint32_t out[2])
{
    uint32_t a = ctr[0], b = ctr[1];
    philox2x32_10(&a, &b, key);
    out[0] = a; out[1] = b;
}

/* 52-bit uniform strictly inside (0,1): (x + 1/2) * 2^-52, every step exact */
static inline double u52(uint32_t a, uint32_t b)
{
    uint64_t x = ((uint64_t)(a >> 6) << 26) | (uint64_t)(b >> 6);
    return ((double)x + 0.5) * 0x1p-52;
}

/* stream tags (high byte of key word 1) */
enum { ORC_TAG_ROW = 1, ORC_TAG_GAMMA = 2, ORC_TAG_SYNTH_ROW = 3, ORC_TAG_SYNTH_TX = 4, ORC_TAG_SIMU = 5 };

typedef struct {
    uint32_t k0, k1;
    uint32_t c0, c1, c2, c3; /* c3 = running block index */
} orc_stream;

static inline orc_stream stream_make(uint64_t seed, uint32_t chain, uint32_t tag, uint64_t id, uint32_t iter)
{
    orc_stream s;
    s.k0 = (uint32_t)seed;
    s.k1 = (uint32_t)(seed >> 32) ^ (chain & 0x00FFFFFFu) ^ (tag << 24);
    s.c0 = (uint32_t)id;
    s.c1 = (uint32_t)(id >> 32);
    s.c2 = iter;
    s.c3 = 0;
    return s;
}

/* Row stream (mmseq_amd/csrc/mmg_math.h:Stream2).  One Philox2x32-10 block serves the two rows 2q and 2q+1: key from
 * (seed, chain, tag, q >> 32), counter (q & 0xffffffff, iteration); row id r = 2q + h takes output word h as the 32-bit
 * uniform (x + 1/2) 2^-32 -- the resolution of the reference's gsl_rng_mt19937 draws; the b-th uniform of a row uses
 * key + b * 0xBB67AE85. */
typedef struct { uint32_t key, c0, c1, blk, half; } orc_stream2;

static inline orc_stream2 stream2_make(uint64_t seed, uint32_t chain, uint32_t tag, uint64_t id, uint32_t iter)
{
    orc_stream2 s;
    s.key = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B1u) ^ (chain * 0x85EBCA6Bu) ^ (tag << 28) ^
            ((uint32_t)(id >> 33) * 0xC2B2AE35u);
    s.c0 = (uint32_t)(id >> 1);
    s.c1 = iter;
    s.blk = 0;
    s.half = (uint32_t)id & 1u;
    return s;
}

static inline uint32_t stream2_next_word(orc_stream2 *s)
{
    uint32_t a = s->c0, b = s->c1;
    philox2x32_10(&a, &b, s->key + s->blk * 0xBB67AE85u);
    s->blk++;
    return s->half ? b : a;
}
static inline double u32_unit(uint32_t x) { return ((double)x + 0.5) * 0x1p-32; }
static inline double stream2_next(orc_stream2 *s) { return u32_unit(stream2_next_word(s)); }

/* one Philox block = one pair of uniforms */
static inline void stream_pair(orc_stream *s, double *ua, double *ub)
{
    orc_u4 c = {{s->c0, s->c1, s->c2, s->c3}};
    c = philox4x32_10(c, s->k0, s->k1);
    s->c3++;
    *ua = u52(c.v[0], c.v[1]);
    *ub = u52(c.v[2], c.v[3]);
}

/* ------------------------------------------------------------------------- */
/* fdlibm-style log / exp (Sun Microsystems' published algorithm: argument     */
/* reduction + fixed minimax polynomial).  Only +,-,*,/ and integer bit ops,   */
/* each rounded once (compile with -ffp-contract=off), so any IEEE-754 machine */
/* reproduces the same bits.                                                   */
/* ------------------------------------------------------------------------- */
typedef union { double f; uint64_t i; } orc_bits;

double orc_log(double x)
{
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                        Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                        Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                        Lg7 = 1.479819860511658591e-01;
    orc_bits u; u.f = x;
    uint32_t hx = (uint32_t)(u.i >> 32);
    int k = 0;
    if (hx < 0x00100000u || (hx >> 31)) {
        if ((u.i << 1) == 0) return -INFINITY;
        if (hx >> 31) return NAN;
        k -= 54; x *= 0x1p54; u.f = x; hx = (uint32_t)(u.i >> 32);
    } else if (hx >= 0x7ff00000u) {
        return x;
    } else if (hx == 0x3ff00000u && (u.i << 32) == 0) {
        return 0.0;
    }
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    u.i = ((uint64_t)hx << 32) | (u.i & 0xffffffffu);
    x = u.f;
    double f = x - 1.0;
    double hfsq = 0.5 * f * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double dk = (double)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

static inline double orc_scalbn(double y, int n)
{
    orc_bits u;
    if (n > 1023) {
        y *= 0x1p1023; n -= 1023;
        if (n > 1023) { y *= 0x1p1023; n -= 1023; if (n > 1023) n = 1023; }
    } else if (n < -1022) {
        y *= 0x1p-1022 * 0x1p53; n += 1022 - 53;
        if (n < -1022) { y *= 0x1p-1022 * 0x1p53; n += 1022 - 53; if (n < -1022) n = -1022; }
    }
    u.i = (uint64_t)(0x3ff + n) << 52;
    return y * u.f;
}

double orc_exp(double x)
{
    static const double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10,
                        invln2 = 1.44269504088896338700e+00,
                        P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                        P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                        P5 = 4.13813679705723846039e-08;
    orc_bits u; u.f = x;
    uint32_t hx = (uint32_t)(u.i >> 32);
    int sign = (int)(hx >> 31);
    hx &= 0x7fffffffu;
    double hi, lo;
    int k;
    if (hx >= 0x4086232bu) { /* |x| >= 708.39 or NaN */
        if (x != x) return x;
        if (x > 709.782712893383973096) return INFINITY;
        if (x < -745.13321910194110842) return 0.0;
    }
    if (hx > 0x3fd62e42u) { /* |x| > 0.5 ln2 */
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
    double xx = x * x;
    double c = x - xx * (P1 + xx * (P2 + xx * (P3 + xx * (P4 + xx * P5))));
    double y = 1.0 + (x * c / (2.0 - c) - lo + hi);
    if (k == 0) return y;
    return orc_scalbn(y, k);
}

void orc_log_v(int64_t n, const double *x, double *out) { for (int64_t i = 0; i < n; ++i) out[i] = orc_log(x[i]); }
void orc_exp_v(int64_t n, const double *x, double *out) { for (int64_t i = 0; i < n; ++i) out[i] = orc_exp(x[i]); }

/* ------------------------------------------------------------------------- */
/* Keyed samplers: normal (Marsaglia polar), gamma (Marsaglia-Tsang 2000, the  */
/* algorithm gsl_ran_gamma documents), binomial (inversion / Hormann BTRS).    */
/* ------------------------------------------------------------------------- */
static inline double keyed_normal(orc_stream *s)
{
    for (;;) {
        double ua, ub;
        stream_pair(s, &ua, &ub);
        double v1 = 2.0 * ua - 1.0, v2 = 2.0 * ub - 1.0;
        double r2 = v1 * v1 + v2 * v2;
        if (r2 >= 1.0 || r2 == 0.0) continue;
        return v1 * sqrt(-2.0 * orc_log(r2) / r2);
    }
}

/* unit-scale Gamma(a), a > 0 */
static double keyed_gamma_unit(orc_stream *s, double a)
{
    double boost_a = a;
    if (a < 1.0) a = a + 1.0;
    double d = a - 1.0 / 3.0;
    double c = (1.0 / 3.0) / sqrt(d);
    double v, x, ua, ub;
    for (;;) {
        do {
            x = keyed_normal(s);
            v = 1.0 + c * x;
        } while (v <= 0.0);
        v = v * v * v;
        stream_pair(s, &ua, &ub);
        double x2 = x * x;
        if (ua < 1.0 - 0.0331 * x2 * x2) break;
        if (orc_log(ua) < 0.5 * x2 + d * (1.0 - v + orc_log(v))) break;
    }
    double g = d * v;
    if (boost_a < 1.0) {
        stream_pair(s, &ua, &ub);
        g = g * orc_exp(orc_log(ua) / boost_a);
    }
    return g;
}

double orc_gamma_draw(uint64_t seed, uint32_t chain, uint32_t iter, uint64_t t, double shape, double scale)
{
    orc_stream s = stream_make(seed, chain, ORC_TAG_GAMMA, t, iter);
    return keyed_gamma_unit(&s, shape) * scale;
}

/* simulated trace of an isoform without hits (src/mmseq.cpp:971-978), keyed by (seed, SIMU, id, sample) */
void orc_simu_gamma_trace(uint64_t seed, uint64_t id, double shape, double scale, int n, double *out)
{
    for (int i = 0; i < n; ++i) {
        orc_stream s = stream_make(seed, 0, ORC_TAG_SIMU, id, (uint32_t)i);
        out[i] = keyed_gamma_unit(&s, shape) * scale;
    }
}

/* Stirling-series tail log(k!) - [ (k+1/2)log(k+1) - (k+1) + log(2pi)/2 ] */
static inline double stirling_tail(double k)
{
    static const double tab[10] = {0.0810614667953272, 0.0413406959554092, 0.0276779256849983,
                                   0.02079067210376509, 0.0166446911898211, 0.0138761288230707,
                                   0.0118967099458917, 0.0104112652619720, 0.00925546218271273,
                                   0.00833056343336287};
    if (k <= 9.0) return tab[(int)k];
    double kp1sq = (k + 1.0) * (k + 1.0);
    return (1.0 / 12.0 - (1.0 / 360.0 - 1.0 / 1260.0 / kp1sq) / kp1sq) / (k + 1.0);
}

/* uniform source abstraction so the same binomial serves Philox streams and MT19937 */
typedef double (*orc_unif_fn)(void *);

static uint32_t binomial_draw(orc_unif_fn U, void *st, uint32_t n, double p,
                              double (*LOG)(double), double (*EXP)(double))
{
    if (n == 0 || !(p > 0.0)) return 0;
    if (p >= 1.0) return n;
    int flip = 0;
    if (p > 0.5) { p = 1.0 - p; flip = 1; }
    double dn = (double)n;
    uint32_t res;
    if (dn * p < 10.0) {
        /* inversion by sequential search (Kachitvichyanukul & Schmeiser BINV) */
        double q = 1.0 - p, s = p / q, a = (dn + 1.0) * s;
        for (;;) {
            double r = EXP(dn * LOG(q));
            double u = U(st);
            uint32_t x = 0;
            int ok = 1;
            while (u > r) {
                u -= r;
                x++;
                if (x > n) { ok = 0; break; }
                r *= (a / (double)x - s);
            }
            if (ok) { res = x; break; }
        }
    } else {
        /* Hormann (1993) transformed rejection with squeeze, BTRS */
        double q = 1.0 - p, spq = sqrt(dn * p * q);
        double b = 1.15 + 2.53 * spq;
        double a = -0.0873 + 0.0248 * b + 0.01 * p;
        double c = dn * p + 0.5;
        double vr = 0.92 - 4.2 / b;
        double r = p / q;
        double alpha = (2.83 + 5.1 / b) * spq;
        double m = floor((dn + 1.0) * p);
        for (;;) {
            double u = U(st) - 0.5;
            double v = U(st);
            double us = 0.5 - fabs(u);
            double kf = floor((2.0 * a / us + b) * u + c);
            if (kf < 0.0 || kf > dn) continue;
            if (us >= 0.07 && v <= vr) { res = (uint32_t)kf; break; }
            v = LOG(v * alpha / (a / (us * us) + b));
            double ub = (m + 0.5) * LOG((m + 1.0) / (r * (dn - m + 1.0))) +
                        (dn + 1.0) * LOG((dn - m + 1.0) / (dn - kf + 1.0)) +
                        (kf + 0.5) * LOG(r * (dn - kf + 1.0) / (kf + 1.0)) +
                        stirling_tail(m) + stirling_tail(dn - m) - stirling_tail(kf) - stirling_tail(dn - kf);
            if (v <= ub) { res = (uint32_t)kf; break; }
        }
    }
    return flip ? n - res : res;
}

static double seq2_unif(void *p) { return stream2_next((orc_stream2 *)p); }

uint32_t orc_binomial_keyed(uint64_t seed, uint64_t id, uint32_t n, double p)
{
    orc_stream2 q = stream2_make(seed, 0, ORC_TAG_ROW, id, 0);
    return binomial_draw(seq2_unif, &q, n, p, orc_log, orc_exp);
}

/* ------------------------------------------------------------------------- */
/* Keyed Gibbs kernels (the spec the HIP kernels implement)                    */
/* ------------------------------------------------------------------------- */
/* One categorical draw from the 32-bit word x (mmg_math.h: draw_target): the target is (x + 1/2) 2^-32 * total rounded ONCE,
 * computed as fma(x, total 2^-32, total 2^-33); then the first hit whose running sum (stored order) exceeds it. */
static inline uint32_t pick_index(const uint32_t *cols, uint32_t L, const double *mu, double total, uint32_t x)
{
    if (!(total > 0.0) || !(total < INFINITY)) { /* degenerate weights: uniform over the hits */
        uint32_t j = (uint32_t)(u32_unit(x) * (double)L);
        return j < L ? j : L - 1;
    }
    const double ts = total * 0x1p-32, hs = ts * 0.5;
    double target = fma((double)x, ts, hs), acc = 0.0;
    for (uint32_t j = 0; j < L; ++j) {
        acc += mu[cols[j]];
        if (target < acc) return j;
    }
    return L - 1;
}

/* restates src/mmseq.cpp:865-891 for one row; counts are ADDED into cnt */
static void keyed_row_allocate(const uint32_t *cols, uint32_t L, uint32_t k, const double *mu,
                               uint64_t seed, uint32_t chain, uint32_t iter, uint64_t row_id, int32_t *cnt)
{
    if (L == 0 || k == 0) return;
    if (L == 1) { cnt[cols[0]] += (int32_t)k; return; }
    double total = 0.0;
    for (uint32_t j = 0; j < L; ++j) total += mu[cols[j]];
    if (orc_draws_categoricals(k, L)) {
        orc_stream2 s = stream2_make(seed, chain, ORC_TAG_ROW, row_id, iter);
        for (uint32_t d = 0; d < k; ++d) {
            uint32_t x = stream2_next_word(&s);
            cnt[cols[pick_index(cols, L, mu, total, x)]] += 1;
        }
        return;
    }
    /* multinomial by conditional binomials (published gsl_ran_multinomial algorithm) */
    orc_stream2 q = stream2_make(seed, chain, ORC_TAG_ROW, row_id, iter);
    uint32_t remaining = k;
    double rem_w = total;
    int degenerate = !(total > 0.0) || !(total < INFINITY);
    for (uint32_t j = 0; j + 1 < L && remaining > 0; ++j) {
        double w = mu[cols[j]];
        double p = degenerate ? 1.0 / (double)(L - j) : (rem_w > 0.0 ? w / rem_w : 1.0);
        if (p > 1.0) p = 1.0;
        uint32_t x = binomial_draw(seq2_unif, &q, remaining, p, orc_log, orc_exp);
        cnt[cols[j]] += (int32_t)x;
        remaining -= x;
        rem_w -= w;
    }
    if (remaining > 0) cnt[cols[L - 1]] += (int32_t)remaining;
}

/* K1 equivalent: cnt[t] = sum over rows of the allocation (cnt is overwritten). */
void orc_sample_counts(uint64_t m, uint32_t n, const uint64_t *row_ptr, const uint32_t *col_idx,
                       const uint32_t *k, const double *mu, uint64_t seed, uint32_t chain, uint32_t iter,
                       uint64_t row_id_base, int32_t *cnt)
{
    memset(cnt, 0, (size_t)n * sizeof(int32_t));
    for (uint64_t i = 0; i < m; ++i) {
        uint64_t b = row_ptr[i];
        uint32_t L = (uint32_t)(row_ptr[i + 1] - b);
        keyed_row_allocate(col_idx + b, L, k ? k[i] : 1u, mu, seed, chain, iter, row_id_base + i, cnt);
    }
}

/* K2 equivalent: restates src/mmseq.cpp:905-908 with the keyed gamma stream. */
void orc_gamma_update(uint32_t n, const int32_t *cnt, const double *l, double alpha, double beta,
                      uint64_t seed, uint32_t chain, uint32_t iter, double *mu)
{
    for (uint32_t t = 0; t < n; ++t) {
        orc_stream s = stream_make(seed, chain, ORC_TAG_GAMMA, t, iter);
        mu[t] = keyed_gamma_unit(&s, alpha + (double)cnt[t]) * (1.0 / (beta + l[t]));
    }
}

/* Full keyed chain.  trace is transcript-major like src/mmseq.cpp:914
 * (trace[t*trace_len + s]); gibbs_ss = n_iter/trace_len (:284); sample s is taken
 * after iteration iter when iter % ss == 0 (:911).  cnt_last (optional) returns the
 * last iteration's counts; sum_log / sum_log2 (optional) the per-transcript
 * moments of orc_log(mu) over the kept samples.  Threads only split rows /
 * transcripts: results are independent of the thread count. */
int orc_gibbs_keyed(uint64_t m, uint32_t n, const uint64_t *row_ptr, const uint32_t *col_idx,
                    const uint32_t *k, const double *l, const double *mu0, double alpha, double beta,
                    uint64_t seed, uint32_t chain, int n_iter, int trace_len, uint64_t row_id_base,
                    double *trace, int32_t *cnt_last, double *sum_log, double *sum_log2, double *mu_last)
{
    if (n_iter <= 0 || trace_len <= 0 || n_iter % trace_len != 0) return 1;
    int ss = n_iter / trace_len;
    double *mu = (double *)malloc((size_t)n * sizeof(double));
    int nth = 1;
#ifdef _OPENMP
    /* threads only split the work (results do not depend on them): keep >= 4096 rows per thread, at most 32 */
    nth = omp_get_max_threads();
    if (nth > 32) nth = 32;
    if ((uint64_t)nth > m / 4096 + 1) nth = (int)(m / 4096 + 1);
#endif
    int32_t *slabs = (int32_t *)malloc((size_t)n * nth * sizeof(int32_t));
    int32_t *cnt = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    memcpy(mu, mu0, (size_t)n * sizeof(double));
    if (sum_log) memset(sum_log, 0, (size_t)n * sizeof(double));
    if (sum_log2) memset(sum_log2, 0, (size_t)n * sizeof(double));
    for (int iter = 0; iter < n_iter; ++iter) {
        memset(slabs, 0, (size_t)n * nth * sizeof(int32_t));
#pragma omp parallel num_threads(nth)
        {
            int tid = 0;
#ifdef _OPENMP
            tid = omp_get_thread_num();
#endif
            int32_t *my = slabs + (size_t)tid * n;
#pragma omp for schedule(static)
            for (int64_t i = 0; i < (int64_t)m; ++i) {
                uint64_t b = row_ptr[i];
                uint32_t L = (uint32_t)(row_ptr[i + 1] - b);
                keyed_row_allocate(col_idx + b, L, k ? k[i] : 1u, mu, seed, chain, (uint32_t)iter,
                                   row_id_base + (uint64_t)i, my);
            }
#pragma omp for schedule(static)
            for (int64_t t = 0; t < (int64_t)n; ++t) {
                int32_t c = 0;
                for (int j = 0; j < nth; ++j) c += slabs[(size_t)j * n + t];
                cnt[t] = c;
                orc_stream s = stream_make(seed, chain, ORC_TAG_GAMMA, (uint64_t)t, (uint32_t)iter);
                mu[t] = keyed_gamma_unit(&s, alpha + (double)c) * (1.0 / (beta + l[t]));
            }
        }
        if (iter % ss == 0) {
            int sidx = iter / ss;
            for (uint32_t t = 0; t < n; ++t) {
                if (trace) trace[(size_t)t * trace_len + sidx] = mu[t];
                if (sum_log || sum_log2) {
                    double lg = orc_log(mu[t]);
                    if (sum_log) sum_log[t] += lg;
                    if (sum_log2) sum_log2[t] += lg * lg;
                }
            }
        }
    }
    if (cnt_last) memcpy(cnt_last, cnt, (size_t)n * sizeof(int32_t));
    if (mu_last) memcpy(mu_last, mu, (size_t)n * sizeof(double));
    free(mu); free(slabs); free(cnt);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Reference-structured engine: MT19937 per thread (src/mmseq.cpp:834-838)     */
/* ------------------------------------------------------------------------- */
typedef struct { uint32_t mt[624]; int idx; } orc_mt;

static void mt_seed(orc_mt *g, uint32_t s)
{
    if (s == 0) s = 4357; /* GSL maps seed 0 to 4357 */
    g->mt[0] = s;
    for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

static inline uint32_t mt_next(orc_mt *g)
{
    if (g->idx >= 624) {
        uint32_t *mt = g->mt;
        int kk;
        for (kk = 0; kk < 624 - 397; ++kk) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; kk < 623; ++kk) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

void orc_mt19937(uint32_t seed, int64_t n, uint32_t *out)
{
    orc_mt g; mt_seed(&g, seed);
    for (int64_t i = 0; i < n; ++i) out[i] = mt_next(&g);
}

/* gsl_rng_uniform: get()/2^32 ; gsl_rng_uniform_pos rejects 0 */
static double mt_unif(void *p) { return mt_next((orc_mt *)p) / 4294967296.0; }
static double mt_unif_pos(void *p) { double x; do x = mt_unif(p); while (x == 0.0); return x; }

static double mt_normal(orc_mt *g)
{
    for (;;) {
        double v1 = 2.0 * mt_unif(g) - 1.0, v2 = 2.0 * mt_unif(g) - 1.0;
        double r2 = v1 * v1 + v2 * v2;
        if (r2 >= 1.0 || r2 == 0.0) continue;
        return v1 * sqrt(-2.0 * log(r2) / r2);
    }
}

static double mt_gamma(orc_mt *g, double a, double b)
{
    if (a < 1.0) {
        double u = mt_unif_pos(g);
        return mt_gamma(g, 1.0 + a, b) * pow(u, 1.0 / a);
    }
    double d = a - 1.0 / 3.0, c = (1.0 / 3.0) / sqrt(d), x, v, u;
    for (;;) {
        do { x = mt_normal(g); v = 1.0 + c * x; } while (v <= 0.0);
        v = v * v * v;
        u = mt_unif_pos(g);
        if (u < 1.0 - 0.0331 * x * x * x * x) break;
        if (log(u) < 0.5 * x * x + d * (1.0 - v + log(v))) break;
    }
    return b * d * v;
}

/* gsl_ran_multinomial: n[j] = Binomial(p[j]/(norm - sum_p), N - sum_n) */
static void mt_multinomial(orc_mt *g, uint32_t K, uint32_t N, const double *p, uint32_t *x)
{
    double norm = 0.0, sum_p = 0.0;
    uint32_t sum_n = 0;
    for (uint32_t j = 0; j < K; ++j) norm += p[j];
    for (uint32_t j = 0; j < K; ++j) {
        if (p[j] > 0.0) x[j] = binomial_draw(mt_unif, g, N - sum_n, p[j] / (norm - sum_p), log, exp);
        else x[j] = 0;
        sum_p += p[j];
        sum_n += x[j];
    }
}

/* Reference-structured chain (src/mmseq.cpp:833-918).  Returns wall seconds spent in
 * the iteration loop (trace capture into memory included, no file I/O) via *seconds. */
int orc_gibbs_ref(uint64_t m, uint32_t n, const uint64_t *row_ptr, const uint32_t *col_idx,
                  const uint32_t *k, const double *l, const double *mu0, double alpha, double beta,
                  uint32_t seed, int n_iter, int trace_len, int n_threads, double *trace,
                  int32_t *cnt_last, double *mu_last, double *seconds)
{
    if (n_iter <= 0 || trace_len <= 0) return 1;
    int ss = n_iter / trace_len;
    if (ss < 1) ss = 1;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    orc_mt *rg = (orc_mt *)malloc(sizeof(orc_mt) * (size_t)n_threads);
    for (int i = 0; i < n_threads; ++i) mt_seed(&rg[i], seed + (uint32_t)i);
    double *mu = (double *)malloc((size_t)n * sizeof(double));
    memcpy(mu, mu0, (size_t)n * sizeof(double));
    int32_t *Xcolsum = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    int32_t *Xcolsums = (int32_t *)malloc((size_t)n * n_threads * sizeof(int32_t));
    double t0 = 0, t1 = 0;
#ifdef _OPENMP
    t0 = omp_get_wtime();
#endif
    for (int iter = 0; iter < n_iter; ++iter) {
        memset(Xcolsum, 0, (size_t)n * sizeof(int32_t));
        memset(Xcolsums, 0, (size_t)n * n_threads * sizeof(int32_t));
#pragma omp parallel num_threads(n_threads)
        {
            int tid = 0;
#ifdef _OPENMP
            tid = omp_get_thread_num();
#endif
            double p[128];
            uint32_t x[128];
            double *pp = p; uint32_t *xx = x; uint32_t cap = 128;
#pragma omp for schedule(static)
            for (int64_t i = 0; i < (int64_t)m; ++i) {
                uint64_t b = row_ptr[i];
                uint32_t L = (uint32_t)(row_ptr[i + 1] - b);
                if (L > cap) {
                    if (pp != p) { free(pp); free(xx); }
                    cap = L; pp = (double *)malloc(sizeof(double) * cap); xx = (uint32_t *)malloc(sizeof(uint32_t) * cap);
                }
                for (uint32_t j = 0; j < L; ++j) pp[j] = mu[col_idx[b + j]];
                mt_multinomial(&rg[tid], L, k ? k[i] : 1u, pp, xx);
                for (uint32_t j = 0; j < L; ++j) Xcolsums[(size_t)col_idx[b + j] + (size_t)n * tid] += (int32_t)xx[j];
            }
            if (pp != p) { free(pp); free(xx); }
#pragma omp for schedule(static)
            for (int64_t t = 0; t < (int64_t)n; ++t)
                for (int j = 0; j < n_threads; ++j) Xcolsum[t] += Xcolsums[t + (size_t)j * n];
#pragma omp for schedule(static)
            for (int64_t t = 0; t < (int64_t)n; ++t)
                mu[t] = mt_gamma(&rg[tid], alpha + Xcolsum[t], 1.0 / (beta + l[t]));
        }
        if (iter % ss == 0 && trace && iter / ss < trace_len)
            for (uint32_t t = 0; t < n; ++t) trace[(size_t)t * trace_len + iter / ss] = mu[t];
    }
#ifdef _OPENMP
    t1 = omp_get_wtime();
#endif
    if (seconds) *seconds = t1 - t0;
    if (cnt_last) memcpy(cnt_last, Xcolsum, (size_t)n * sizeof(int32_t));
    if (mu_last) memcpy(mu_last, mu, (size_t)n * sizeof(double));
    free(rg); free(mu); free(Xcolsum); free(Xcolsums);
    return 0;
}

/* statistical helpers for the sampler tests */
void orc_mt_gamma_v(uint32_t seed, double a, double b, int64_t n, double *out)
{
    orc_mt g; mt_seed(&g, seed);
    for (int64_t i = 0; i < n; ++i) out[i] = mt_gamma(&g, a, b);
}
void orc_mt_binomial_v(uint32_t seed, uint32_t nn, double p, int64_t n, uint32_t *out)
{
    orc_mt g; mt_seed(&g, seed);
    for (int64_t i = 0; i < n; ++i) out[i] = binomial_draw(mt_unif, &g, nn, p, log, exp);
}
void orc_keyed_gamma_v(uint64_t seed, double a, double b, int64_t n, double *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = orc_gamma_draw(seed, 0, 0, (uint64_t)i, a, b);
}
void orc_keyed_binomial_v(uint64_t seed, uint32_t nn, double p, int64_t n, uint32_t *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = orc_binomial_keyed(seed, (uint64_t)i, nn, p);
}
void orc_keyed_normal_v(uint64_t seed, int64_t n, double *out)
{
    for (int64_t i = 0; i < n; ++i) {
        orc_stream s = stream_make(seed, 0, ORC_TAG_GAMMA, (uint64_t)i, 0);
        out[i] = keyed_normal(&s);
    }
}

/* ------------------------------------------------------------------------- */
/* Start values, EM, unique hits (deterministic host pieces)                   */
/* ------------------------------------------------------------------------- */
/* src/mmseq.cpp:617-638: mu0[t] = (sum_{rows i containing t} k_i/|row i|) / l[t];
 * unique_hits[t] = sum of k_i over rows {t} (histogram bin 0, :633 / :1500). */
void orc_start_values(uint64_t m, uint32_t n, const uint64_t *row_ptr, const uint32_t *col_idx,
                      const uint32_t *k, const double *l, double *mu0, int32_t *unique_hits)
{
    for (uint32_t t = 0; t < n; ++t) { mu0[t] = 0.0; if (unique_hits) unique_hits[t] = 0; }
    for (uint64_t i = 0; i < m; ++i) {
        uint64_t b = row_ptr[i], e = row_ptr[i + 1];
        uint32_t L = (uint32_t)(e - b), ki = k ? k[i] : 1u;
        for (uint64_t j = b; j < e; ++j) mu0[col_idx[j]] += (double)ki / (double)L;
        if (L == 1 && unique_hits) unique_hits[col_idx[b]] += (int32_t)ki;
    }
    for (uint32_t t = 0; t < n; ++t) mu0[t] /= l[t];
}

/* The device's order-independent form of the same start values (mmseq_amd/csrc/misc_kernels.h:k_start_values): every share
 * q = (double)k / (double)L enters as the integer floor(q * 2^52); the exact integer sum is converted to double with
 * round-to-nearest-even, scaled by 2^-52 and divided by l[t].  Differs from orc_start_values by rounding only. */
void orc_start_values_exact(uint64_t m, uint32_t n, const uint64_t *row_ptr, const uint32_t *col_idx,
                            const uint32_t *k, const double *l, double *mu0)
{
    unsigned __int128 *acc = (unsigned __int128 *)calloc(n ? n : 1, sizeof(unsigned __int128));
    for (uint64_t i = 0; i < m; ++i) {
        uint64_t b = row_ptr[i], e = row_ptr[i + 1];
        uint32_t L = (uint32_t)(e - b), ki = k ? k[i] : 1u;
        if (L == 0 || ki == 0) continue;
        double q = (double)ki / (double)L;
        uint64_t bits;
        memcpy(&bits, &q, 8);
        int ex = (int)((bits >> 52) & 0x7ff) - 1023;
        unsigned __int128 mant = (bits & 0xfffffffffffffull) | (1ull << 52);
        unsigned __int128 T = ex >= 0 ? mant << ex : mant >> (-ex);
        for (uint64_t j = b; j < e; ++j) acc[col_idx[j]] += T;
    }
    for (uint32_t t = 0; t < n; ++t) {
        unsigned __int128 v = acc[t];
        double d;
        if (v == 0) d = 0.0;
        else {
            int top = 127;
            while (!((v >> top) & 1)) --top;
            if (top <= 52) d = (double)(uint64_t)v;
            else {
                int sh = top - 52;
                uint64_t mant = (uint64_t)(v >> sh);
                unsigned __int128 rest = v & (((unsigned __int128)1 << sh) - 1), half = (unsigned __int128)1 << (sh - 1);
                if (rest > half || (rest == half && (mant & 1))) ++mant;
                d = ldexp((double)mant, sh);
            }
        }
        mu0[t] = d * 0x1p-52 / l[t];
    }
    free(acc);
}

/* src/mmseq.cpp:741-811 in the reference's own summation order (per transcript over rows ascending,
 * log-likelihood over rows then transcripts): EM until llr <= epsilon or max_iter.  mu is updated in
 * place; returns the iteration count.  Kept to pin orc_em (below) to the reference's arithmetic.  Row denominators are cached (same value
 * the reference recomputes per (t,row) at :787-791). */
int orc_em_seq(uint64_t m, uint32_t n, const uint64_t *row_ptr, const uint32_t *col_idx, const uint32_t *k,
           const double *l, double *mu, int max_iter, double epsilon, double *loglik_out)
{
    double *d = (double *)malloc((size_t)m * sizeof(double));
    double *acc = (double *)malloc((size_t)n * sizeof(double));
    double loglik = 0.0;
    for (uint64_t i = 0; i < m; ++i) {
        double s = 0.0;
        for (uint64_t j = row_ptr[i]; j < row_ptr[i + 1]; ++j) s += mu[col_idx[j]];
        d[i] = s;
        loglik += (double)(k ? k[i] : 1u) * orc_log(s);
    }
    for (uint32_t t = 0; t < n; ++t) loglik -= mu[t] * l[t];
    double llr = INFINITY; /* the reference starts from epsilon+1 (src/mmseq.cpp:756): first sweep always runs */
    int iter = 0;
    while (iter < max_iter && llr > epsilon) {
        memset(acc, 0, (size_t)n * sizeof(double));
        for (uint64_t i = 0; i < m; ++i) {
            double r = (double)(k ? k[i] : 1u) / d[i];
            for (uint64_t j = row_ptr[i]; j < row_ptr[i + 1]; ++j) acc[col_idx[j]] += r;
        }
        for (uint32_t t = 0; t < n; ++t) mu[t] = mu[t] * acc[t] / l[t];
        double ll = 0.0;
        for (uint64_t i = 0; i < m; ++i) {
            double s = 0.0;
            for (uint64_t j = row_ptr[i]; j < row_ptr[i + 1]; ++j) s += mu[col_idx[j]];
            d[i] = s;
            ll += (double)(k ? k[i] : 1u) * orc_log(s);
        }
        for (uint32_t t = 0; t < n; ++t) ll -= mu[t] * l[t];
        llr = ll - loglik;
        loglik = ll;
        iter++;
    }
    if (loglik_out) *loglik_out = loglik;
    free(d); free(acc);
    return iter;
}


/* ---- EM with order-independent accumulation: the specification the device implements bit for bit.
 * Same fixed point as src/mmseq.cpp:761-811 (mu_t <- mu_t/l_t * S_t, S_t = sum_{i contains t} x_i, x_i = k_i/d_i,
 * d_i = sum_{t in i} mu_t, stop when the log-likelihood gain <= epsilon), but the two big sums are accumulated EXACTLY in
 * fixed point, so any traversal order (threads, workgroups, shards) gives the same bits:
 *   once: N_t = hits in column t, sl_t = 63 - bitlen(N_t).  N_t terms below 2^(sl_t-1) sum below 2^62.
 *   per rows pass, a scale exponent E_t per live transcript (mu_t == 0 or not finite: dead, it is left alone):
 *       measured:  XE_t = max_i ilogb(x_i) from a pass of its own (integer max), E_t = sl_t - 2 - XE_t: every term is
 *                  below 2^(sl_t-1) by construction.  Used for the first pass and for repeats.
 *       carried:   E_t = sl_t - 2 - ilogb(S_t(last pass)) - ORC_EM_MARGIN, i.e. the last sum with MARGIN bits of head room;
 *                  every term is checked (ilogb(x_i) + E_t < sl_t - 1), and so is every sum (HI_t >= 2^(sl_t-2-MARGIN-SHRINK):
 *                  the sum did not shrink by more than SHRINK bits); if any check fails anywhere, the whole pass is
 *                  repeated on measured exponents.
 *   row i: d_i in row order; rows with k_i = 0 or d_i outside [2^-900, 2^900] are skipped (degenerate state);
 *       for each hit t:  Y = x_i * 2^E_t;  HI_t += floor(Y);  LO_t += floor((Y - floor(Y)) * 2^sl_t)     -- both exact: the 53-bit
 *                        significand of x_i shifted into a 64.64 fixed-point number, integer part and the top sl_t fraction bits
 *       v = k_i*log(d_i)*2^12:  LLH += floor(v);  LLL += trunc((v - floor(v))*2^31)               (none can overflow 64 bits)
 *   S_t = ldexp((double)HI_t + ldexp((double)LO_t, -sl_t), -E_t);  mu_t <- mu_t*S_t/l_t
 *   loglik = ((double)LLH*2^-12 + (double)LLL*2^-43) - pen,  pen = sum_t mu_t l_t by 256-blocks (halving tree inside a
 *       block, blocks added in index order).
 * A term is truncated at 2^-(2 sl_t - 2) of the largest term (measured) or at most 2^-(2 sl_t - 18) of S_t (carried), with
 * sl_t >= 31 (>= 37 below 2^26 hits per transcript): below the fp64 rounding of the reference's own sums; tests pin this against orc_em_seq. */
static inline int em_bitlen(uint64_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }
#define ORC_EM_DEAD (-32768)
#define ORC_EM_MARGIN 8
#define ORC_EM_SHRINK 8
static inline int em_alive(double v) { return v > 0.0 && v < INFINITY; }
static double em_pen(uint32_t n, const double *mu, const double *l)
{
    double tot = 0.0, red[256];
    for (uint32_t b0 = 0; b0 < n; b0 += 256) {
        for (uint32_t i = 0; i < 256; ++i) red[i] = (b0 + i < n) ? mu[b0 + i] * l[b0 + i] : 0.0;
        for (int s = 128; s > 0; s >>= 1) for (int i = 0; i < s; ++i) red[i] += red[i + s];
        tot += red[0];
    }
    return tot;
}
/* One rows pass.  measure != 0: only XE (max ilogb(x_i) per column, INT32_MIN where none).  Otherwise HI/LO and the
 * log-likelihood limbs from mu / E; returns 1 if a checked term failed. */
static int em_rows_pass(int measure, uint64_t m, const uint64_t *row_ptr, const uint32_t *col_idx, const uint32_t *k,
                        const double *mu, const int32_t *E, const int32_t *cap, const int32_t *sl, int32_t *XE, uint64_t *HI,
                        uint64_t *LO, int64_t *llh, uint64_t *lll)
{
    int64_t h = 0; uint64_t lo = 0;
    int viol = 0;
    for (uint64_t i = 0; i < m; ++i) {
        const uint64_t b = row_ptr[i], e = row_ptr[i + 1];
        if (e == b) continue;
        double d = 0.0;
        for (uint64_t j = b; j < e; ++j) d += mu[col_idx[j]];
        if ((k && k[i] == 0) || !(d >= 0x1p-900 && d <= 0x1p900)) continue;
        const double kk = (double)(k ? k[i] : 1u), x = kk / d;
        const int xe = ilogb(x);
        uint64_t T;
        memcpy(&T, &x, 8);
        T = (T << 11) | (1ull << 63); /* x is normal here (d in [2^-900, 2^900], 1 <= k < 2^32) */
        if (measure) {
            for (uint64_t j = b; j < e; ++j) if (xe > XE[col_idx[j]]) XE[col_idx[j]] = xe;
            continue;
        }
        const double v = kk * orc_log(d) * 4096.0, fv = floor(v);
        h += (int64_t)fv;
        lo += (uint64_t)((v - fv) * 2147483648.0);
        for (uint64_t j = b; j < e; ++j) {
            const uint32_t t = col_idx[j];
            if (E[t] == ORC_EM_DEAD) continue;
            if (xe + E[t] >= cap[t]) { viol = 1; continue; }
            /* x = T * 2^(xe - 63) with T the significand, leading bit at bit 63; Y = T * 2^(p - 63), p = xe + E_t <= 62 */
            const int p = xe + E[t];
            uint64_t yh, fr; /* fr: the fraction of Y, left-aligned (bit 63 = 2^-1) */
            if (p >= 0) { yh = T >> (63 - p); fr = (T << 1) << p; }
            else { yh = 0; fr = p >= -64 ? T >> (-1 - p) : 0; }
            HI[t] += yh;
            LO[t] += fr >> (64 - sl[t]);
        }
    }
    if (!measure) { *llh = h; *lll = lo; }
    return viol;
}
/* redo_out (optional): number of passes that had to be repeated on measured exponents */
int orc_em_x(uint64_t m, uint32_t n, const uint64_t *row_ptr, const uint32_t *col_idx, const uint32_t *k,
             const double *l, double *mu, int max_iter, double epsilon, double *loglik_out, int *redo_out)
{
    int32_t *sl = (int32_t *)malloc(n * sizeof(int32_t)), *XE = (int32_t *)malloc(n * sizeof(int32_t));
    int32_t *E = (int32_t *)malloc(n * sizeof(int32_t)), *sexp = (int32_t *)malloc(n * sizeof(int32_t));
    int32_t *cap = (int32_t *)malloc(n * sizeof(int32_t));
    uint64_t *HI = (uint64_t *)calloc(n, sizeof(uint64_t)), *LO = (uint64_t *)malloc(n * sizeof(uint64_t));
    for (uint64_t j = 0; j < row_ptr[m]; ++j) HI[col_idx[j]]++;
    for (uint32_t t = 0; t < n; ++t) { sl[t] = 63 - em_bitlen(HI[t]); sexp[t] = INT32_MIN; }
    int64_t llh = 0; uint64_t lll = 0;
    int redo = 0, first = 1;
#define EM_PASS() do { \
        for (int measured = first; measured < 2; ++measured) { \
            if (measured) { \
                for (uint32_t t = 0; t < n; ++t) XE[t] = INT32_MIN; \
                em_rows_pass(1, m, row_ptr, col_idx, k, mu, NULL, NULL, sl, XE, NULL, NULL, NULL, NULL); \
            } \
            for (uint32_t t = 0; t < n; ++t) { \
                const int32_t ref = measured ? XE[t] : sexp[t]; \
                HI[t] = 0; LO[t] = 0; \
                if (!em_alive(mu[t]) || ref == INT32_MIN) { E[t] = ORC_EM_DEAD; cap[t] = 63; } \
                else if (measured) { E[t] = sl[t] - 2 - ref; cap[t] = 63; } \
                else { E[t] = sl[t] - 2 - ref - ORC_EM_MARGIN; cap[t] = sl[t] - 1; } \
            } \
            int viol = em_rows_pass(0, m, row_ptr, col_idx, k, mu, E, cap, sl, NULL, HI, LO, &llh, &lll); \
            if (!measured) for (uint32_t t = 0; t < n; ++t) \
                if (E[t] != ORC_EM_DEAD && (HI[t] >> (sl[t] - 2 - ORC_EM_MARGIN - ORC_EM_SHRINK)) == 0) viol = 1; \
            if (!viol) break; \
            ++redo; \
        } first = 0; } while (0)
#define EM_LL() (((double)llh * 0x1p-12 + (double)lll * 0x1p-43) - em_pen(n, mu, l))
    EM_PASS();
    double loglik = EM_LL();
    double llr = INFINITY; /* the reference starts from epsilon+1 (src/mmseq.cpp:756): first sweep always runs */
    int iter = 0;
    while (iter < max_iter && llr > epsilon) {
        for (uint32_t t = 0; t < n; ++t) {
            const double S = E[t] == ORC_EM_DEAD ? 0.0 : ldexp((double)HI[t] + ldexp((double)LO[t], -sl[t]), -E[t]);
            mu[t] = mu[t] * S / l[t];
            sexp[t] = em_alive(S) ? (int32_t)ilogb(S) : INT32_MIN;
        }
        EM_PASS();
        const double ll = EM_LL();
        llr = ll - loglik;
        loglik = ll;
        iter++;
    }
#undef EM_PASS
#undef EM_LL
    if (loglik_out) *loglik_out = loglik;
    if (redo_out) *redo_out = redo;
    free(sl); free(XE); free(E); free(sexp); free(cap); free(HI); free(LO);
    return iter;
}
int orc_em(uint64_t m, uint32_t n, const uint64_t *row_ptr, const uint32_t *col_idx, const uint32_t *k,
           const double *l, double *mu, int max_iter, double epsilon, double *loglik_out)
{
    return orc_em_x(m, n, row_ptr, col_idx, k, l, mu, max_iter, epsilon, loglik_out, NULL);
}

/* src/uh.cpp:3-26 literally: for each group g, sum k_i over rows whose every column
 * is a member of g (an empty row counts for every group).  member is n x G row-major. */
void orc_uh(uint64_t m, uint32_t n, const uint64_t *row_ptr, const uint32_t *col_idx, const uint32_t *k,
            uint32_t G, const uint8_t *member, int32_t *res)
{
    (void)n;
    for (uint32_t g = 0; g < G; ++g) {
        int32_t r = 0;
        for (uint64_t i = 0; i < m; ++i) {
            int uniq = 1;
            for (uint64_t j = row_ptr[i]; j < row_ptr[i + 1]; ++j)
                if (!member[(size_t)col_idx[j] * G + g]) { uniq = 0; break; }
            if (uniq) r += (int32_t)(k ? k[i] : 1u);
        }
        res[g] = r;
    }
}

/* ------------------------------------------------------------------------- */
/* Sokal IACT (src/sokal.cc:33-87) with a plain radix-2 FFT.  x is destroyed,   */
/* exactly like the reference.  Return codes as the reference: 100 too long,    */
/* 200 n<4, 201 not a power of two.                                            */
/* ------------------------------------------------------------------------- */
static void fft_radix2(double *re, double *im, int n)
{
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
    for (int len = 2; len <= n; len <<= 1) {
        double ang = -2.0 * M_PI / len;
        for (int i = 0; i < n; i += len)
            for (int j = 0; j < len / 2; ++j) {
                double wr = cos(ang * j), wi = sin(ang * j);
                int a = i + j, b = i + j + len / 2;
                double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
    }
}

int orc_sokal(int n, double *x, double *var, double *tau, int *m)
{
    if (n > (2 << 20)) return 100;
    if (n < 4) return 200;
    for (int t = n; t > 1; t >>= 1) if (t & 1) return 201;
    double *im = (double *)calloc((size_t)n, sizeof(double));
    fft_radix2(x, im, n);
    for (int i = 0; i < n; ++i) { x[i] = x[i] * x[i] + im[i] * im[i]; im[i] = 0.0; }
    x[0] = 0.0;
    fft_radix2(x, im, n);
    free(im);
    *var = x[0] / ((double)n * (n - 1));
    double c = 1.0 / x[0];
    for (int i = 0; i < n; ++i) x[i] *= c;
    double sum = -0.333333333333333333333;
    *m = n + 1;
    for (int i = 0; i < n; ++i) {
        sum += x[i] - 0.166666666666666666666;
        if (sum < 0) { *m = i + 1; break; }
    }
    *tau = 2 * (sum + (*m - 1.0) / 6.0);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Synthetic hits generator (the build's own spec, no         */
/* reference counterpart).  Every row is a pure function of (seed, row id).     */
/* ------------------------------------------------------------------------- */
/* transcript tables: efflen = max(50, round(exp(log 1500 + 0.6 z))), theta = exp(2 z')
 * with 30% zeros; cdf[t] = inclusive running sum of theta*efflen (fp64, sequential). */
void orc_synth_transcripts(uint64_t seed, uint32_t T, double *efflen, double *theta, double *cdf)
{
    double run = 0.0;
    for (uint32_t t = 0; t < T; ++t) {
        orc_stream s = stream_make(seed, 0, ORC_TAG_SYNTH_TX, t, 0);
        double z1 = keyed_normal(&s), z2 = keyed_normal(&s), ua, ub;
        stream_pair(&s, &ua, &ub);
        double e = floor(orc_exp(7.3132203870903014 + 0.6 * z1) + 0.5); /* log(1500) */
        if (e < 50.0) e = 50.0;
        double th = (ua < 0.3) ? 0.0 : orc_exp(2.0 * z2);
        efflen[t] = e;
        theta[t] = th;
        run += th * e;
        cdf[t] = run;
    }
}

/* Poisson(lambda) inclusive CDF table for 0..98 (row length = 1 + Poisson, clipped to 100) */
void orc_synth_len_cdf(double lambda, double *cdf99)
{
    double p = orc_exp(-lambda), run = 0.0;
    for (int j = 0; j < 99; ++j) {
        run += p;
        cdf99[j] = run;
        p = p * lambda / (double)(j + 1);
    }
}

static inline uint32_t synth_row_len(const double *len_cdf, double u)
{
    uint32_t j = 0;
    while (j < 99 && !(u < len_cdf[j])) ++j;
    return 1 + j; /* 1..100 */
}

uint32_t orc_synth_row_len(uint64_t seed, uint64_t row, const double *len_cdf)
{
    orc_stream s = stream_make(seed, 0, ORC_TAG_SYNTH_ROW, row, 0);
    double ua, ub;
    stream_pair(&s, &ua, &ub);
    return synth_row_len(len_cdf, ua);
}

/* gene-block mode of the generator (mmg_synth_desc.gene_size / far_family, include/mmgibbs.h; no reference counterpart: what an
 * aligner's output looks like, src/bam2hits.cpp:271-300).  The family bijection g -> a g mod n_genes: a multiplier coprime to n_genes
 * derived from the seed, and its inverse (mmg_types.h: synth_family_params restates this). */
void orc_synth_family_params(uint64_t seed, uint32_t n_genes, uint32_t *a, uint32_t *ainv)
{
    if (n_genes < 3) { *a = 1; *ainv = 1 % (n_genes ? n_genes : 1); return; } /* (2 genes: no multiplier in [2, 2] is coprime to 2: the identity) */
    uint64_t x = (seed + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
    x ^= x >> 31;
    const uint64_t span = n_genes - 1 ? n_genes - 1 : 1;
    uint64_t m = 2 + x % span;
    for (;; m = 2 + (m - 1) % span) {
        uint64_t u = m, v = n_genes;
        while (v) { uint64_t t = u % v; u = v; v = t; }
        if (u == 1) break;
    }
    int64_t t0 = 0, t1 = 1, r0 = n_genes, r1 = (int64_t)m;
    while (r1 != 1) { int64_t q = r0 / r1, r2 = r0 - q * r1, t2 = t0 - q * t1; r0 = r1; r1 = r2; t0 = t1; t1 = t2; }
    *a = (uint32_t)m;
    *ainv = (uint32_t)(((t1 % (int64_t)n_genes) + (int64_t)n_genes) % (int64_t)n_genes);
}

static uint32_t synth_first_transcript(uint32_t T, const double *cdf, double ub)
{
    double target = ub * cdf[T - 1];
    uint32_t lo = 0, hi = T - 1;
    while (lo < hi) { uint32_t mid = lo + (hi - lo) / 2; if (target < cdf[mid]) hi = mid; else lo = mid + 1; }
    return lo;
}

/* One row of the gene-block generator: the first transcript ~ cdf, the others distinct isoforms of ITS gene (an odd-stride walk over
 * the gene's other slots); a far row's last drawn hit is an isoform of another gene of the read's paralogue family (far_family >= 2)
 * or a transcript anywhere outside the gene (far_family = 0).  Sorted ascending.  Returns the length (<= gene size). */
uint32_t orc_synth_row_genes(uint64_t seed, uint64_t row, uint32_t T, const double *cdf, const double *len_cdf, double far_fraction,
                             uint32_t gene_size, uint32_t far_family, uint32_t *cols /* >= 100 */)
{
    const uint32_t n_genes = (T + gene_size - 1) / gene_size;
    orc_stream s = stream_make(seed, 0, ORC_TAG_SYNTH_ROW, row, 0);
    double ua, ub;
    stream_pair(&s, &ua, &ub);
    uint32_t L = synth_row_len(len_cdf, ua);
    if (L > T) L = T;
    const uint32_t t0 = synth_first_transcript(T, cdf, ub);
    const uint32_t wb = (t0 / gene_size) * gene_size;
    const uint32_t W = gene_size < T - wb ? gene_size : T - wb;
    if (L > W) L = W;
    cols[0] = t0;
    if (L <= 1) return L;
    uint32_t nslots = W - 1, Wp = 1;
    while (Wp < nslots) Wp <<= 1;
    double uc, ud;
    stream_pair(&s, &uc, &ud);
    uint32_t start = (uint32_t)(uc * (double)Wp);
    uint32_t stride = ((uint32_t)(ud * (double)(Wp / 2 ? Wp / 2 : 1)) << 1) | 1u;
    int far = 0;
    uint32_t tfar = 0;
    if (far_fraction > 0.0 && far_family >= 2u && n_genes >= 2u * far_family) {
        uint32_t fa, fainv;
        orc_synth_family_params(seed, n_genes, &fa, &fainv);
        double ue, uf;
        stream_pair(&s, &ue, &uf);
        const uint32_t F = far_family, g0 = t0 / gene_size;
        const uint32_t pg = (uint32_t)(((uint64_t)fa * g0) % n_genes), fam = pg / F, idx = pg % F;
        const uint32_t fsize = F < n_genes - fam * F ? F : n_genes - fam * F;
        if (fsize >= 2u) {
            far = ue < far_fraction;
            uint32_t j = 1u + (uint32_t)(uf * (double)(fsize - 1u));
            if (j > fsize - 1u) j = fsize - 1u;
            const uint32_t pm = fam * F + (idx + j) % fsize;
            const uint32_t gm = (uint32_t)(((uint64_t)fainv * pm) % n_genes);
            double ug, uh;
            stream_pair(&s, &ug, &uh);
            const uint32_t Wm = gene_size < T - gm * gene_size ? gene_size : T - gm * gene_size;
            uint32_t iso = (uint32_t)(ug * (double)Wm);
            if (iso >= Wm) iso = Wm - 1u;
            tfar = gm * gene_size + iso;
        }
    } else if (far_fraction > 0.0 && T > 2u * W) {
        double ue, uf;
        stream_pair(&s, &ue, &uf);
        far = ue < far_fraction;
        tfar = (uint32_t)(uf * (double)T);
        if (tfar >= T) tfar = T - 1;
        if (tfar >= wb && tfar < wb + W) tfar = (tfar + W) % T;
    }
    uint32_t got = 1, pos = start & (Wp - 1);
    while (got < L) {
        if (pos < nslots) {
            uint32_t t = wb + pos;
            if (t >= t0) t += 1;
            if (far && got == L - 1) t = tfar;
            cols[got++] = t;
        }
        pos = (pos + stride) & (Wp - 1);
    }
    for (uint32_t i = 1; i < L; ++i) {
        uint32_t v = cols[i], j = i;
        while (j > 0 && cols[j - 1] > v) { cols[j] = cols[j - 1]; --j; }
        cols[j] = v;
    }
    return L;
}

/* CSR of rows [row0, row0 + R) of the gene-block generator; col_idx NULL: sizing pass. */
void orc_synth_csr_genes(uint64_t seed, uint64_t row0, uint64_t R, uint32_t T, const double *cdf, const double *len_cdf, double far_fraction,
                         uint32_t gene_size, uint32_t far_family, uint64_t *row_ptr, uint32_t *col_idx)
{
    row_ptr[0] = 0;
    for (uint64_t r = 0; r < R; ++r) {
        uint32_t tmp[100];
        row_ptr[r + 1] = row_ptr[r] + orc_synth_row_genes(seed, row0 + r, T, cdf, len_cdf, far_fraction, gene_size, far_family, tmp);
    }
    if (!col_idx) return;
#pragma omp parallel for schedule(static) if (R > 100000)
    for (int64_t r = 0; r < (int64_t)R; ++r) {
        uint32_t tmp[100];
        uint32_t L = orc_synth_row_genes(seed, row0 + (uint64_t)r, T, cdf, len_cdf, far_fraction, gene_size, far_family, tmp);
        memcpy(col_idx + row_ptr[r], tmp, L * sizeof(uint32_t));
    }
}

/* Row contents: first transcript ~ cdf (lower bound of ub*total); the other len-1 are
 * distinct members of a 129-wide index window around it, visited by an odd-stride walk;
 * the row is returned sorted ascending (src/mmseq.cpp:412).  Returns the length. */
uint32_t orc_synth_row(uint64_t seed, uint64_t row, uint32_t T, const double *cdf, const double *len_cdf,
                       int uniform, double far_fraction, uint32_t *cols /* >= 100 */)
{
    orc_stream s = stream_make(seed, 0, ORC_TAG_SYNTH_ROW, row, 0);
    double ua, ub;
    stream_pair(&s, &ua, &ub);
    uint32_t L = synth_row_len(len_cdf, ua);
    if (L > T) L = T;
    /* first transcript */
    double target = ub * cdf[T - 1];
    uint32_t lo = 0, hi = T - 1;
    while (lo < hi) { uint32_t mid = lo + (hi - lo) / 2; if (target < cdf[mid]) hi = mid; else lo = mid + 1; }
    uint32_t t0 = lo;
    cols[0] = t0;
    if (L > 1) {
        uint32_t W = uniform ? T : (T < 129u ? T : 129u); /* window size incl. t0 */
        uint32_t wb;
        if (uniform) wb = 0;
        else {
            int64_t b = (int64_t)t0 - 64;
            if (b < 0) b = 0;
            if (b + (int64_t)W > (int64_t)T) b = (int64_t)T - (int64_t)W;
            wb = (uint32_t)b;
        }
        /* walk slots (start + i*stride) mod Wp over the W-1 slots that skip t0; Wp = next pow2 >= W-1 */
        uint32_t nslots = W - 1, Wp = 1;
        while (Wp < nslots) Wp <<= 1;
        double uc, ud;
        stream_pair(&s, &uc, &ud);
        uint32_t start = (uint32_t)(uc * (double)Wp);
        uint32_t stride = ((uint32_t)(ud * (double)(Wp / 2 ? Wp / 2 : 1)) << 1) | 1u;
        /* far rows: the last drawn hit becomes a transcript anywhere in [0, T) outside the window */
        int far = 0;
        uint32_t tfar = 0;
        if (far_fraction > 0.0 && !uniform && T > 2u * W) {
            double ue, uf;
            stream_pair(&s, &ue, &uf);
            far = ue < far_fraction;
            tfar = (uint32_t)(uf * (double)T);
            if (tfar >= T) tfar = T - 1;
            if (tfar >= wb && tfar < wb + W) tfar = (tfar + W) % T;
        }
        uint32_t got = 1, pos = start & (Wp - 1);
        while (got < L) {
            if (pos < nslots) {
                uint32_t t = wb + pos;
                if (t >= t0) t += 1; /* skip t0 */
                if (far && got == L - 1) t = tfar;
                cols[got++] = t;
            }
            pos = (pos + stride) & (Wp - 1);
        }
        /* insertion sort ascending */
        for (uint32_t i = 1; i < L; ++i) {
            uint32_t v = cols[i], j = i;
            while (j > 0 && cols[j - 1] > v) { cols[j] = cols[j - 1]; --j; }
            cols[j] = v;
        }
    }
    return L;
}

/* Whole CSR for rows [row0, row0+R): row_ptr has R+1 entries (row_ptr[0]=0). If
 * col_idx is NULL only row_ptr is filled (sizing pass). */
void orc_synth_csr(uint64_t seed, uint64_t row0, uint64_t R, uint32_t T, const double *cdf,
                   const double *len_cdf, int uniform, double far_fraction, uint64_t *row_ptr, uint32_t *col_idx)
{
    row_ptr[0] = 0;
    for (uint64_t r = 0; r < R; ++r) {
        uint32_t L = orc_synth_row_len(seed, row0 + r, len_cdf);
        if (L > T) L = T;
        row_ptr[r + 1] = row_ptr[r] + L;
    }
    if (!col_idx) return;
#pragma omp parallel for schedule(static) if (R > 100000)
    for (int64_t r = 0; r < (int64_t)R; ++r) {
        uint32_t tmp[100];
        uint32_t L = orc_synth_row(seed, row0 + (uint64_t)r, T, cdf, len_cdf, uniform, far_fraction, tmp);
        memcpy(col_idx + row_ptr[r], tmp, L * sizeof(uint32_t));
    }
}
