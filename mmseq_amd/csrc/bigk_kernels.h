// K1 for the rows on the conditional-binomial chain (kclass 3 of the canonical layout, mmg_types.h): hit sets shared by so many reads
// that drawing them one by one costs more than hits - 1 binomials -- what the abundant transcripts of every collapsed hits file
// produce (src/mmseq.cpp:409-440, the draw :880).
//
// The tile kernel (sell_kernels.h) ran such a row in the lane that owns it: a loop over the row's hits around binomial()
// (mmg_math.h), whose rejection loops and four-logarithm acceptance test every lane of the wave sits through as long as ONE lane needs
// them -- about 1 900 vector instructions per binomial and wave, the lanes doing something useful in a fifth of them -- and the tiles
// hold few such rows each (the canonical order is by band first), so most lanes had no row at all.
//
// Here the rows come from a LIST (the stored positions of the kclass-3 rows, ascending: layout.hip), a wave owns a contiguous
// piece of it, and every lane is a small state machine over its current row:
//     FETCH  take the next row of the piece: extents, k, the total of its weights (stored order), the row's keyed stream
//     STEP   hit j: p = w_j / remaining weight, the outcomes that need no draw, and (n p < 10) the inversion that ends at x = 0
//     BSET   the constants of BTRS;  TRY  one attempt: two uniforms, the candidate, the squeeze
//     SLOW   the exact acceptance test of an attempt that missed the squeeze: from an fp32 estimate with a bound on its own error where the
//            two settle the comparison (mmg_math.h: btrs_pretest), the four fp64 logarithms at once for the rest (1 in 1 000)
//     IFULL  the inversion whose uniform may exceed r0: the search on fp32 terms where no boundary of the cumulative sum is closer to
//            the uniform than the bound on the sum (binv_pretest), logarithm, exponential and the fp64 search for the rest (1 in 2 500)
// A trip of the wave's loop counts the lanes waiting in every phase and runs the phases that hold a tenth of them or more (the fullest
// one in any case): the frequent cheap work stays in step, the rare expensive paths wait until they are worth a run.  Lanes are
// different rows with their own keyed stream (mmg_math.h: Stream2), so a lane may run ahead of its neighbours: the work a lane does, and
// the uniforms it consumes, are exactly what the sequential loop does for its row -- the same bits as allocate_row / orc_gibbs_keyed --
// only the order BETWEEN lanes changes.  The expensive test then runs for a dozen lanes or more at a time instead of for one or two, and
// the two fp32 tests take the outcome of the fp64 code, never its place (the comparison is only taken where fp32 cannot get it wrong:
// mmg_selftest_btrs_pretest / _binv_pretest count disagreements, none in 10^10 each): about 720 vector instructions per 64 binomials
// (profiles/r06_bigk_ab.md).  A piece is 64 list entries (measured best at every list length:
// mmgibbs.hip, bigk_piece); a longer piece works -- a lane that finishes its row fetches the next -- and is what the tests run too.
//
// Weights are gathered from the global vector (L1 / L2: the rows of a piece are neighbours in the canonical order, i.e. in transcript
// space); counts go to a 255-wide LDS window that follows the piece through the bands (flushed with one atomic per touched transcript
// when it moves), hits outside it to the global vector directly.  Which way a count takes never changes a sum.
#pragma once
#include "gibbs_kernels.h"

namespace mmg {

#if defined(MMG_BIGK_STATS)
// diagnostics build (tools/bigk_stats.py): runs of every phase and the lanes they served, summed over the waves of all launches
__device__ unsigned long long g_bigk_stats[16];
#endif

enum : uint32_t { BK_FETCH = 0, BK_STEP = 1, BK_BSET = 2, BK_TRY = 3, BK_SLOW = 4, BK_IFULL = 5, BK_IDLE = 6 };
// A phase runs in a trip of the wave's loop when it holds at least BK_SHARE_NUM / BK_SHARE_DEN of the lanes that have work (or is the
// fullest one): a tenth.  1 / 64 is "every phase that has a lane, every trip" (the shortest dependent chain per row), 1 / 1 is "the fullest
// phase only" (the fewest instructions per row).  Measured at 64 rows per wave (profiles/r06_bigk_ab.md): 1/2 1.223, 1/3 1.175, 1/5 1.129,
// 1/8 1.198 ms on 2 M rows of k = 1000; the hit sets of a collapsed file 0.159 / 0.151 / 0.146 / 0.144 ms per sweep.  With the exact test and
// the full inversion decided in fp32 the expensive phases cost a third of what they did and waiting for lanes pays less: 1/4 0.875,
// 1/5 0.845, 1/7 0.831, 1/10 0.830, 1/16 0.839, 1/32 0.844, 1/64 0.842 ms.
#ifndef BK_SHARE_NUM
#define BK_SHARE_NUM 1
#endif
#ifndef BK_SHARE_DEN
#define BK_SHARE_DEN 10
#endif
#ifndef BK_STEPS_PER_TRIP
#define BK_STEPS_PER_TRIP 2
#endif

template <typename IdxT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_sample_bigk(const IdxT *__restrict__ row_ptr, const uint32_t *__restrict__ col_idx,
                                                    const uint32_t *__restrict__ kmult, const uint64_t *__restrict__ list, uint64_t n_list,
                                                    uint32_t per_wave, const double *__restrict__ gmu /* [grid.y][n] */,
                                                    int32_t *gcnt /* [grid.y][n] */, SampleArgs a)
{
    constexpr uint32_t WIN = SELL_WIN;
    __shared__ int32_t s_cnt[WIN + 1];
    const uint32_t lane = threadIdx.x;
    gmu += (size_t)blockIdx.y * a.n;
    gcnt += (size_t)blockIdx.y * a.n + (size_t)(blockIdx.x & a.cnt_rep_mask) * a.cnt_rep_stride; // (mmg_types.h: CNT_REPLICAS)
    a.chain += blockIdx.y;
    uint64_t next = (uint64_t)blockIdx.x * per_wave;                  // uniform: first list entry not handed out yet
    const uint64_t end = next + per_wave < n_list ? next + per_wave : n_list;
    if (next >= end) return;
    for (uint32_t i = lane; i < WIN + 1; i += 64) s_cnt[i] = 0;
    uint32_t W = 0;          // uniform: base of the count window
    bool have_win = false;

    auto flush_window = [&]() {
        for (uint32_t i = lane; i < WIN; i += 64) {
            const int32_t v = s_cnt[i];
            if (v) { global_count_add(gcnt, W + i, v); s_cnt[i] = 0; }
        }
    };
    auto add = [&](uint32_t col, int32_t x) {
        const uint32_t d = col - W;
        if (have_win && d < WIN) atomicAdd(&s_cnt[d], x);
        else global_count_add(gcnt, col, x);
    };

    uint32_t ph = BK_FETCH;
    // the lane's row
    uint64_t b = 0;                       // its first hit in col_idx
    uint32_t L = 0, j = 0, remaining = 0, c_first = 0, c_last = 0;
    double rem_w = 0.0;
    bool degenerate = false;
    uint32_t skey = 0, sc0 = 0, sblk = 0, shalf = 0;   // Stream2 of the row (mmg_math.h)
    // hit j and the two behind it, requested a trip or more before they are used
    uint32_t c_cur = 0, c_nx = 0, c_n2 = 0;
    double w_cur = 0.0, w_nx = 0.0;
    // the step in progress: Binomial(nn, P) for hit j
    uint32_t nn = 0, x = 0;
    double dn = 0.0, P = 0.0, U = 0.0;
    bool flip = false, resolved = false;
    double A = 0.0, B = 0.0, CC = 0.0, VR = 0.0, ALPHA = 0.0; // BTRS constants a, b, c, vr and sqrt(n p q) (SLOW derives r, alpha and m from them)
    double us = 0.0, kf = 0.0, vv = 0.0;                                       // the attempt waiting for the exact test

    auto next_unit = [&]() -> double { // Stream2::next()
        uint32_t xa = sc0, xb = a.iter;
        philox2x32_10(xa, xb, skey + sblk * 0xBB67AE85u);
        ++sblk;
        return u32_unit(shalf ? xb : xa);
    };
    auto count = [&](bool c) -> uint32_t { return (uint32_t)__popcll(__ballot(c)); };

#if defined(MMG_BIGK_STATS)
    uint32_t st_runs[7] = {0, 0, 0, 0, 0, 0, 0}, st_lanes[7] = {0, 0, 0, 0, 0, 0, 0};
#define BK_STAT(q, n) do { st_runs[q]++; st_lanes[q] += (n); } while (0)
#else
#define BK_STAT(q, n) do { } while (0)
#endif
    for (;;) {
        if (next >= end && ph == BK_FETCH) ph = BK_IDLE;
        uint32_t thr;
        {
            const uint32_t nF = count(ph == BK_FETCH), nS = count(ph == BK_STEP), nB = count(ph == BK_BSET || ph == BK_TRY), nW = count(ph == BK_SLOW), nI = count(ph == BK_IFULL);
            const uint32_t work = nF + nS + nB + nW + nI, fullest = max(max(max(nF, nS), max(nB, nW)), nI);
            if (work == 0) break;
            thr = min(fullest, max(1u, (work * BK_SHARE_NUM + BK_SHARE_DEN - 1) / BK_SHARE_DEN));
        }

        // ---- FETCH: the next rows of the piece for the lanes without one
        const uint64_t mF = __ballot(ph == BK_FETCH);
        if ((uint32_t)__popcll(mF) >= thr) {
            BK_STAT(0, (uint32_t)__popcll(mF));
            bool got = false;
            if (ph == BK_FETCH) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mF >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mF, 0u));
                const uint64_t idx = next + rank;
                if (idx < end) {
                    const uint64_t r = list[idx];
                    b = (uint64_t)row_ptr[r];
                    L = (uint32_t)((uint64_t)row_ptr[r + 1] - b);
                    remaining = kmult[r];
                    // the total of the row's weights, added in stored order from 0.0 (allocate_row: View::total)
                    double t = 0.0;
                    const uint32_t last = L - 1;
                    for (uint32_t i = 0; i < L; i += 4) {
                        const uint32_t c0 = col_idx[b + i], c1 = col_idx[b + min(i + 1, last)], c2 = col_idx[b + min(i + 2, last)], c3 = col_idx[b + min(i + 3, last)];
                        const double w0 = gmu[c0], w1 = gmu[c1], w2 = gmu[c2], w3 = gmu[c3];
                        t += w0;
                        if (i + 1 < L) t += w1;
                        if (i + 2 < L) t += w2;
                        if (i + 3 < L) t += w3;
                        if (i == 0) { c_first = c0; c_cur = c0; w_cur = w0; c_nx = c1; w_nx = w1; c_n2 = c2; }
                    }
                    c_last = col_idx[b + last];
                    rem_w = t;
                    degenerate = !(t > 0.0) || !(t < __builtin_huge_val());
                    j = 0;
                    const uint64_t row_id = a.row_id_base + r;
                    skey = stream2_key(a.seed, a.chain, TAG_ROW, (uint32_t)(row_id >> 33));
                    sc0 = (uint32_t)(row_id >> 1);
                    shalf = (uint32_t)row_id & 1u;
                    sblk = 0;
                    ph = BK_STEP;
                    got = true;
                } else ph = BK_IDLE;
            }
            const uint64_t taken = (uint64_t)__popcll(mF);
            next = next + taken < end ? next + taken : end;
            // the count window follows the piece: when fewer than half of the rows in flight start inside it, it moves to the band of the
            // first row just fetched (the rows behind it in the list lie in that band or above)
            const uint64_t mG = __ballot(got);
            if (mG) {
                const uint32_t first = (uint32_t)__ffsll((unsigned long long)mG) - 1u;
                const uint32_t cand = (uint32_t)__builtin_amdgcn_readlane((int)c_first, (int)first) & ~((1u << LAYOUT_BAND_SHIFT) - 1u);
                const bool active = ph >= BK_STEP && ph <= BK_IFULL;
                const uint32_t n_act = count(active), n_in = count(active && have_win && (c_first - W) < LAYOUT_NEAR_SPAN);
                if (!have_win || (2u * n_in < n_act && cand != W)) {
                    if (have_win) flush_window();
                    W = cand;
                    have_win = true;
                }
            }
        }

        // ---- the steps settled so far in this trip: the count, the next hit or the end of the row
        auto do_resolve = [&]() {
            if (__ballot(resolved)) {
                if (resolved) {
                    resolved = false;
                    const uint32_t xr = flip ? nn - x : x;
                    if (xr) add(c_cur, (int32_t)xr);
                    remaining -= xr;
                    rem_w -= w_cur;
                    ++j;
                    if (j + 1 < L && remaining > 0) {
                        ph = BK_STEP;
                        c_cur = c_nx; w_cur = w_nx;
                        c_nx = c_n2;
                        w_nx = gmu[c_nx];                        // (hit j + 1: its id arrived a trip ago)
                        c_n2 = col_idx[b + min(j + 2, L - 1)];
                    } else {
                        if (remaining > 0) add(c_last, (int32_t)remaining);
                        ph = BK_FETCH;
                    }
                }
            }
        };
        // ---- STEP: hit j of the row: p, the outcomes that need no draw, and the inversion that ends at x = 0
        auto do_step = [&]() {
            {
                const uint32_t nS = count(ph == BK_STEP);
                if (nS >= thr) {
                    BK_STAT(1, nS);
                    if (ph == BK_STEP) {
                        // (allocate_row: 1 / (hits left) for a degenerate row, w / (weight left) otherwise, 1 when nothing is left -- one division)
                        const bool whole = !degenerate && !(rem_w > 0.0);
                        double pr = (degenerate || whole ? 1.0 : w_cur) / (degenerate ? (double)(L - j) : (whole ? 1.0 : rem_w));
                        if (pr > 1.0) pr = 1.0;
                        nn = remaining;
                        flip = false;
                        // binomial(q, nn, pr) (mmg_math.h), its loops unrolled into the phases
                        if (!(pr > 0.0)) { x = 0; resolved = true; }
                        else if (pr >= 1.0) { x = nn; resolved = true; }
                        else {
                            double p = pr;
                            if (p > 0.5) { p = 1.0 - p; flip = true; }
                            dn = (double)nn;
                            P = p;
                            const double np = dn * p;
                            if (np < 10.0) {
                                // Inversion: x = 0 iff the uniform does not exceed r0 = exp(n log(1 - p)) as the sequential code computes it.  r0 is
                                // at least 1 - n p - n 2^-54 - 2^-46 (Bernoulli's inequality; 1 - p rounds to within 2^-54, log and exp to an ulp,
                                // |n log(1 - p)| < 20): a uniform below that bound less a margin settles x = 0 without the logarithm and the
                                // exponential -- the common case for the hits of transcripts that carry (next to) nothing.  Same uniform consumed.
                                const double u = next_unit();
                                if (u <= 1.0 - np * (1.0 + 0x1p-10) - dn * 0x1p-49 - 0x1p-40) { x = 0; resolved = true; }
                                else { U = u; ph = BK_IFULL; }
                            } else ph = BK_BSET;
                        }
                    }
                }
            }
        };
        // Six steps in ten settle inside STEP (no draw, or the inversion's x = 0): those lanes take their next hit in the same trip --
        // BK_STEPS_PER_TRIP rounds of STEP + settle before the expensive phases get their turn (measured: 1 -> 2 rounds -4 % on the hit sets of a
        // collapsed file, -5.5 % on 2 M rows of k = 1000; 3 rounds no better: profiles/r06_bigk_ab.md)
        do_step();
        for (int rep = 1; rep < BK_STEPS_PER_TRIP; ++rep) { do_resolve(); do_step(); }

        // ---- BTRS: the constants of a new step, then one attempt (also for the lanes whose last attempt was rejected)
        {
            const uint32_t nB = count(ph == BK_BSET || ph == BK_TRY);
            if (nB >= thr) {
                BK_STAT(2, nB);
                if (ph == BK_BSET) {
                    const double p = P;
                    const double qq = 1.0 - p, spq = dsqrt(dn * p * qq);
                    const double bb = 1.15 + 2.53 * spq;
                    const double aa = -0.0873 + 0.0248 * bb + 0.01 * p;
                    const double cc = dn * p + 0.5;
                    const double vr = 0.92 - 4.2 / bb;
                    A = aa; B = bb; CC = cc; VR = vr; ALPHA = spq; /* (SLOW finishes r, alpha, m: one attempt in seven gets there) */
                    ph = BK_TRY;
                }
                if (ph == BK_TRY) {
                    const double u = next_unit() - 0.5;
                    const double v = next_unit();
                    const double us_ = 0.5 - dabs(u);
                    const double kf_ = dfloor((2.0 * A / us_ + B) * u + CC);
                    if (!(kf_ < 0.0 || kf_ > dn)) {
                        if (us_ >= 0.07 && v <= VR) { x = (uint32_t)kf_; resolved = true; }
                        else { us = us_; kf = kf_; vv = v; ph = BK_SLOW; }
                    }
                }
            }
        }

        // ---- IFULL: the inversion whose uniform may exceed r0 -- the search in fp32 where no boundary comes closer to the uniform than the
        // bound on the fp32 sums (mmg_math.h: binv_pretest), the fp64 search at once for the rest
        {
            const uint32_t nI = count(ph == BK_IFULL);
            if (nI >= thr) {
                BK_STAT(4, nI);
                int pre = -1;
                if (ph == BK_IFULL) {
#if defined(BK_NO_FP32_TESTS) /* diagnostics build (tools/pretest_fullsize_check.py): the fp64 code only -- the chains must not change */
                    pre = -1;
#else
                    pre = binv_pretest(dn, P, U);
#endif
                    if (pre >= 0) { x = (uint32_t)pre; resolved = true; }
                }
                const bool undecided = ph == BK_IFULL && pre < 0;
                if (__ballot(undecided)) {
                    BK_STAT(6, count(undecided));
                    if (undecided) {
                        uint32_t xi;
                        if (binv_exact(dn, P, U, nn, xi)) { x = xi; resolved = true; }
                        else U = next_unit(); // (the sequential loop starts over with the next uniform)
                    }
                }
            }
        }

        // ---- SLOW: the exact acceptance test of the attempts that missed the squeeze -- from its fp32 estimate where that is farther from zero
        // than its own error bound (mmg_math.h: btrs_pretest; 99.9 % of the tests below n = 2000, 95 % above 10^7), in fp64 at once for the rest
        {
            const uint32_t nW = count(ph == BK_SLOW);
            if (nW >= thr) {
                BK_STAT(3, nW);
                int pre = 0;
                if (ph == BK_SLOW) {
#if defined(BK_NO_FP32_TESTS)
                    pre = 0;
#else
                    pre = btrs_pretest(dn, P, kf, us, vv, A, B, ALPHA);
#endif
                    if (pre > 0) { x = (uint32_t)kf; resolved = true; }
                    else if (pre < 0) ph = BK_TRY;
                }
                const bool undecided = ph == BK_SLOW && pre == 0;
                if (__ballot(undecided)) {
                    BK_STAT(5, count(undecided));
                    if (undecided) {
                        // (the expressions of binomial(); every argument of a logarithm is a ratio of positive finite numbers far from the
                        // subnormal range: dlog_pn)
                        if (btrs_exact_test(dn, P, kf, us, vv, A, B, ALPHA)) { x = (uint32_t)kf; resolved = true; }
                        else ph = BK_TRY;
                    }
                }
            }
        }

        do_resolve();
    }
#undef BK_STAT
    if (have_win) flush_window();
#if defined(MMG_BIGK_STATS)
    if (lane == 0)
        for (uint32_t q = 0; q < 7; ++q) { atomicAdd(&g_bigk_stats[q], (unsigned long long)st_runs[q]); atomicAdd(&g_bigk_stats[8 + q], (unsigned long long)st_lanes[q]); }
#endif
}

} // namespace mmg
