// EM (src/mmseq.cpp:741-811) on the device: the start value of the Gibbs chain.
//
// One sweep is mu_t <- mu_t/l_t * S_t with S_t = sum_{rows i containing t} k_i/d_i, d_i = sum_{t in i} mu_t.
// The rows pass is the Gibbs row walk with a different second half: instead of picking one hit, every hit of
// the row receives the row's x_i = k_i/d_i.  It runs on the same sliced-ELL stream and LDS window as K1 (mu
// gathers and the accumulators live in the window), so a sweep costs about as much HBM as a Gibbs iteration.
//
// The sums are accumulated in 2 x 64-bit fixed point (HI/LO limbs, integer atomics): exact, hence independent
// of the order in which lanes, workgroups or devices add their terms -- no transpose and no floating-point
// atomics, and the result equals oracle/mmseq_oracle.c:orc_em bit for bit.  The scale exponent of every
// transcript, the per-term / per-sum checks and the repeat rule are specified there.
#pragma once

namespace mmg {

constexpr int EM_DEAD = -32768;
constexpr int EM_MARGIN = 8;
constexpr int EM_SHRINK = 8;

// packed per-transcript word: 64 - sl in bits 0..5 (a 64-bit shift takes its count from the low six bits of the whole word),
// cap in bits 8..15, E (int16) in bits 16..31
__host__ __device__ __forceinline__ uint32_t em_pack(int E, int sl, int cap)
{
    return ((uint32_t)E << 16) | ((uint32_t)cap << 8) | (uint32_t)(64 - sl);
}
__host__ __device__ __forceinline__ int em_word_E(uint32_t w) { return (int)w >> 16; }
__host__ __device__ __forceinline__ int em_word_cap(uint32_t w) { return (int)((w >> 8) & 0xffu); }
__host__ __device__ __forceinline__ int em_word_sl(uint32_t w) { return 64 - (int)(w & 63u); }
constexpr uint32_t EM_WORD_DEAD = (0x8000u << 16) | (63u << 8) | 1u;

__device__ __forceinline__ double block_sum_256(double v, double *red)
{
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    return red[0];
}

// N_t: hits per column
__global__ __launch_bounds__(256) void k_em_colcount(const uint32_t *__restrict__ col_idx, uint64_t nnz, uint64_t *cnt)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nnz; j += stride) atomicAdd((unsigned long long *)&cnt[col_idx[j]], 1ull);
}

__global__ void k_fill_i32(int32_t *p, uint32_t n, int32_t v)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) p[t] = v;
}

// Before a rows pass: the scale word of every transcript (measured: from XE; carried: from the last sum),
// cleared accumulators, and the penalty sum_t mu_t l_t as one partial per 256-block.
__global__ __launch_bounds__(256) void k_em_prepare(uint32_t n, const double *__restrict__ mu, const double *__restrict__ l,
                                                    const uint64_t *__restrict__ colcnt, const int32_t *__restrict__ ref,
                                                    int measured, uint32_t *__restrict__ word, uint64_t *__restrict__ hi,
                                                    uint64_t *__restrict__ lo, double *__restrict__ partial, uint64_t *ll,
                                                    const uint32_t *__restrict__ int_of_ext)
{
    __shared__ double red[256];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    double pen = 0.0;
    if (t < n) {
        const double m = mu[t];
        const uint64_t N = colcnt[t];
        const int sl = 63 - (N ? 64 - __builtin_clzll(N) : 0);
        const int32_t r = ref[t];
        const bool alive = m > 0.0 && m < __builtin_huge_val();
        uint32_t w;
        if (!alive || r == INT32_MIN) w = em_pack(EM_DEAD, sl, 63);
        else if (measured) w = em_pack(sl - 2 - r, sl, 63);
        else w = em_pack(sl - 2 - r - EM_MARGIN, sl, sl - 1);
        word[t] = w;
        hi[t] = 0;
        lo[t] = 0;
        // the penalty is summed in the CALLER's transcript order (256-blocks of caller ids), whatever the device numbering
        const uint32_t pt = int_of_ext ? int_of_ext[t] : t;
        pen = mu[pt] * l[pt];
    }
    if (blockIdx.x == 0 && threadIdx.x < 3) ll[threadIdx.x] = 0;
    const double tot = block_sum_256(pen, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// carried exponents only: a sum that shrank by more than EM_SHRINK bits has lost precision -> repeat
__global__ void k_em_check(uint32_t n, const uint32_t *__restrict__ word, const uint64_t *__restrict__ hi, uint64_t *ll)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t w = word[t];
    const int E = em_word_E(w), sl = em_word_sl(w);
    if (E != EM_DEAD && (hi[t] >> (sl - 2 - EM_MARGIN - EM_SHRINK)) == 0) atomicOr((unsigned long long *)&ll[2], 1ull);
}

__global__ void k_em_finish(const double *partial, uint32_t np, const uint64_t *ll, EmOut *out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double pen = 0.0;
        for (uint32_t i = 0; i < np; ++i) pen += partial[i];
        out->loglik = ((double)(int64_t)ll[0] * 0x1p-12 + (double)ll[1] * 0x1p-43) - pen;
        out->flag = ll[2];
    }
}

// mu_t <- mu_t * S_t / l_t from the validated accumulators; sexp = ilogb(S_t) for the next carried scale
__global__ void k_em_apply(uint32_t n, double *mu, const double *__restrict__ l, const uint32_t *__restrict__ word,
                           const uint64_t *__restrict__ hi, const uint64_t *__restrict__ lo, int32_t *sexp)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t w = word[t];
    const int E = em_word_E(w), sl = em_word_sl(w);
    const double S = E == EM_DEAD ? 0.0 : dscalbn((double)hi[t] + dscalbn((double)lo[t], -sl), -E);
    mu[t] = mu[t] * S / l[t];
    sexp[t] = (S > 0.0 && S < __builtin_huge_val()) ? dilogb(S) : INT32_MIN;
}

// per-thread running log-likelihood limbs and check flag
struct EmAcc {
    int64_t llh = 0;
    uint64_t lll = 0;
    uint32_t viol = 0;
};

// Row head shared by all paths: returns false if the row is skipped.  x = k/d = T * 2^(xe - 63): xe = ilogb(x), T the 53-bit
// significand with its leading bit at bit 63.
template <bool MEASURE>
__device__ __forceinline__ bool em_row_head(double d, uint32_t kk, EmAcc &acc, uint64_t &T, int &xe)
{
    if (kk == 0 || !(d >= 0x1p-900 && d <= 0x1p900)) return false;
    const double dk = (double)kk;
    const uint64_t xb = bits_of(dk / d);
    xe = (int)((xb >> 52) & 0x7ff) - 1023; // x is normal here
    T = (xb << 11) | (1ull << 63);
    if (!MEASURE) {
        const double v = dk * dlog_pn(d) * 4096.0, fv = dfloor(v); // d is normal here (the range test above)
        acc.llh += (int64_t)fv;
        acc.lll += (uint64_t)((v - fv) * 2147483648.0);
    }
    return true;
}

// One term, Y = x * 2^E = T * 2^(p - 63) with p = xe + E: yh = floor(Y), yl = the top sl bits of its fraction (both exact: shifts of
// the significand).  Returns false if nothing is to be added: a dead transcript, or a failed check (flagged: the pass is repeated).
__device__ __forceinline__ bool em_term(uint64_t T, int xe, uint32_t w, EmAcc &acc, uint64_t &yh, uint64_t &yl)
{
    const int E = em_word_E(w);
    if (E == EM_DEAD) return false;
    const int p = xe + E;
    if (p >= em_word_cap(w)) { acc.viol |= 1u; return false; }
    uint64_t fr; // the fraction of Y, left-aligned
    if (p >= 0) { yh = T >> (63 - p); fr = (T << 1) << p; }
    else { yh = 0; fr = p >= -64 ? T >> (-1 - p) : 0; }
    yl = fr >> (w & 63u);
    return true;
}

__device__ __forceinline__ void em_acc_commit(const EmAcc &acc, uint64_t *s_ll, uint64_t *gll)
{
    // s_ll[3] zeroed by the caller before the walk
    if (acc.llh) atomicAdd((unsigned long long *)&s_ll[0], (unsigned long long)acc.llh);
    if (acc.lll) atomicAdd((unsigned long long *)&s_ll[1], (unsigned long long)acc.lll);
    if (acc.viol) atomicOr((unsigned long long *)&s_ll[2], 1ull);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_ll[0]) atomicAdd((unsigned long long *)&gll[0], (unsigned long long)s_ll[0]);
        if (s_ll[1]) atomicAdd((unsigned long long *)&gll[1], (unsigned long long)s_ll[1]);
        if (s_ll[2]) atomicOr((unsigned long long *)&gll[2], 1ull);
    }
}

// Any problem, any layout: one thread per row straight from the CSR, global atomics.
template <typename IdxT, bool MEASURE>
__global__ __launch_bounds__(256) void k_em_rows_global(const IdxT *__restrict__ row_ptr, const uint32_t *__restrict__ col_idx,
                                                        const uint32_t *__restrict__ kmult, uint64_t m, EmArgs a)
{
    __shared__ uint64_t s_ll[3];
    if (threadIdx.x < 3) s_ll[threadIdx.x] = 0;
    __syncthreads();
    EmAcc acc;
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < m) {
        const uint64_t b = row_ptr[r], e = row_ptr[r + 1];
        if (e > b) {
            double d = 0.0;
            for (uint64_t j = b; j < e; ++j) d += a.mu[col_idx[j]];
            uint64_t x;
            int xe;
            if (em_row_head<MEASURE>(d, kmult ? kmult[r] : 1u, acc, x, xe)) {
                for (uint64_t j = b; j < e; ++j) {
                    const uint32_t t = col_idx[j];
                    if (MEASURE) {
                        atomicMax(&a.xe[t], xe);
                    } else {
                        uint64_t yh, yl;
                        if (em_term(x, xe, a.word[t], acc, yh, yl)) {
                            if (yh) atomicAdd((unsigned long long *)&a.hi[t], (unsigned long long)yh);
                            if (yl) atomicAdd((unsigned long long *)&a.lo[t], (unsigned long long)yl);
                        }
                    }
                }
            }
        }
    }
    em_acc_commit(acc, s_ll, a.ll);
}

// The rows pass on the sliced-ELL 8-bit stream of k_sample_sell: a tile is one wave, every lane holds its row's window
// indices in registers, nothing is staged through LDS and there are no barriers on the tile path -- which leaves LDS for
// REP replicas of the accumulators (lane l adds to replica l % REP; same-address LDS atomics retire one per two clocks).
template <typename IdxT, bool HAS_K, bool MEASURE, int REP, int W = 1>
__global__ __launch_bounds__(64 * W) void k_em_sell(const IdxT *__restrict__ row_ptr, const uint32_t *__restrict__ col_idx,
                                                const uint32_t *__restrict__ kmult, const SellTile *__restrict__ tiles,
                                                const uint64_t *__restrict__ chunk_tile, const uint8_t *__restrict__ stream, EmArgs a)
{
    constexpr int WIN = (int)SELL_WIN;
    constexpr int AST = WIN + 1;                                            // entries per window array; [WIN] is the pad slot
    constexpr int RST = AST + 8;                                            // entries per accumulator replica: 8 more than a multiple
                                                                            // of 32, so that the same slot of the four replicas lies in four
                                                                            // different quarters of the 64 banks (a window's most abundant
                                                                            // transcript receives most of its terms)
    __shared__ __attribute__((aligned(16))) double s_mu[AST];              // [WIN] stays 0.0: what pad slots read
    __shared__ uint32_t s_w[MEASURE ? 1 : 2 * AST];                        // scale words, one per 8 bytes (a hit's byte offset addresses s_mu, s_w and the accumulators alike); [WIN] dead
    __shared__ uint64_t s_acc[MEASURE ? 1 : 2 * REP * RST];                // HI limbs of the REP replicas, then their LO limbs
    uint64_t *const s_hi = s_acc, *const s_lo = s_acc + (MEASURE ? 0 : REP * RST);
    constexpr uint32_t LO_OFF = (uint32_t)(REP * RST * 8);
    __shared__ int32_t s_xe[MEASURE ? AST : 1];
    __shared__ uint64_t s_ll[3];
    // W waves per workgroup share one window (mu, scale words, accumulators): the LDS per wave drops W-fold, more waves are
    // resident, and the walk of one wave overlaps the scatter of another.  Wave w owns tiles w, w+W, ... of the range; every
    // wave steps through EVERY window slide of the range in tile order (two barriers each), so the barrier counts match.
    constexpr int BS = 64 * W;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t rep_off = MEASURE ? 0u : (lane % REP) * (uint32_t)(RST * 8);

    const uint64_t t_begin = chunk_tile[blockIdx.x], t_end = chunk_tile[blockIdx.x + 1];
    if (t_begin >= t_end) return;
    const uint64_t nt = t_end - t_begin;
    const SellTile *__restrict__ T = tiles + t_begin;

    if (tid < 3) s_ll[tid] = 0;
    for (int i = tid; i < AST; i += BS) {
        if (MEASURE) s_xe[i] = INT32_MIN;
        else { for (int r = 0; r < REP; ++r) { s_hi[r * RST + i] = 0; s_lo[r * RST + i] = 0; } }
    }
    if (tid == 0) { s_mu[WIN] = 0.0; if (!MEASURE) s_w[2 * WIN] = EM_WORD_DEAD; }

    auto flush_window = [&](uint32_t base) {
        for (int i = tid; i < WIN; i += BS) {
            if (MEASURE) {
                const int32_t v = s_xe[i];
                if (v != INT32_MIN) { atomicMax(&a.xe[base + (uint32_t)i], v); s_xe[i] = INT32_MIN; }
            } else {
                uint64_t h = 0, l = 0;
                for (int r = 0; r < REP; ++r) { h += s_hi[r * RST + i]; l += s_lo[r * RST + i]; s_hi[r * RST + i] = 0; s_lo[r * RST + i] = 0; }
                if (h) atomicAdd((unsigned long long *)&a.hi[base + (uint32_t)i], (unsigned long long)h);
                if (l) atomicAdd((unsigned long long *)&a.lo[base + (uint32_t)i], (unsigned long long)l);
            }
        }
    };
    auto load_window = [&](uint32_t base) {
        for (int i = tid; i < WIN; i += BS) {
            const uint32_t c = base + (uint32_t)i;
            s_mu[i] = c < a.n ? a.mu[c] : 0.0;
            if (!MEASURE) s_w[2 * i] = c < a.n ? a.word[c] : EM_WORD_DEAD;
        }
    };
    auto wo = [&](uint32_t off) { return *(const double *)((const char *)s_mu + off); }; // off = window index * 8
#define EMS_OFF0(v) (((v) << 3) & 0x7f8u)
#define EMS_OFF1(v) (((v) >> 5) & 0x7f8u)
#define EMS_OFF2(v) (((v) >> 13) & 0x7f8u)
#define EMS_OFF3(v) (((v) >> 21) & 0x7f8u)
#define EMS_GROUPS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
    struct Buf {
        uint32_t g0, g1, g2, g3, g4, g5, g6, g7;
        uint32_t kk;
    };
    // unconditional, like k_sample_sell::issue (the number of loads per tile must not depend on the path)
    auto issue = [&](const SellTile &d, Buf &bf) {
        const bool fast = d.flags() & (SELL_FAST | SELL_FAR); // uniform: the tile has a block (a far tile's window part is a fast tile's)
        const SellBlock blk(stream + (fast ? d.off16 * 16 : 0), d.meta);
        if (HAS_K) bf.kk = kmult[(fast ? d.r0 : 0) + min(lane, (fast ? d.nrows() : 1u) - 1u)];
#define EMS_ISSUE(i) bf.g##i = blk.template group<i>(lane);
        EMS_GROUPS(EMS_ISSUE)
#undef EMS_ISSUE
    };

    EmAcc acc;
    // far (uniform): a far tile -- the hits of a row outside the window follow its window hits in the far list of the tile's block
    // (k_encode_sell), in stored order; their weights come from global memory, their terms go to the global accumulators
    auto walk = [&](const SellTile &d, const Buf &bf, bool far) {
        const uint32_t ng = d.ng();                                    // uniform
        const uint32_t *__restrict__ src = (const uint32_t *)(stream + d.off16 * 16) + lane; // groups beyond the cached ones
        double t = 0.0;
#define EMS_SUM(i)                                                                                          \
        if ((uint32_t)i < ng) {                                                                              \
            const uint32_t v = bf.g##i;                                                                      \
            const double w0 = wo(EMS_OFF0(v)), w1 = wo(EMS_OFF1(v)), w2 = wo(EMS_OFF2(v)), w3 = wo(EMS_OFF3(v)); \
            t += w0; t += w1; t += w2; t += w3;                                                              \
        }
        EMS_GROUPS(EMS_SUM)
#undef EMS_SUM
#pragma unroll 1
        for (uint32_t g = 8; g < ng; ++g) {
            const uint32_t v = src[(size_t)g * 64];
            const double w0 = wo(EMS_OFF0(v)), w1 = wo(EMS_OFF1(v)), w2 = wo(EMS_OFF2(v)), w3 = wo(EMS_OFF3(v));
            t += w0; t += w1; t += w2; t += w3;
        }
        uint32_t Lf = 0;
        const uint32_t *__restrict__ farp = nullptr;
        if (far) {
            const uint8_t *__restrict__ fb = stream + d.off16 * 16 + (size_t)ng * 256;
            Lf = fb[lane];
            farp = (const uint32_t *)(fb + 64) + lane;
            for (uint32_t f = 0; f < Lf; ++f) t += a.mu[farp[(size_t)f * 64]];
        }
        if ((bf.g0 & 0xffu) == 0xffu && Lf == 0) return; // no hit in this lane: the first slot of a row is a pad (255) only then
        uint64_t x;
        int xe;
        if (!em_row_head<MEASURE>(t, HAS_K ? bf.kk : 1u, acc, x, xe)) return;
        const uint64_t x2 = x << 1;
        // second half: every hit of the row receives x (pads point at the dead slot and add nothing)
        // the usual term (0 <= p < cap) is three 64-bit shifts whose counts are the low six bits of p, ~p (= 63 - p mod 64) and
        // the scale word itself; anything else -- a pad or a dead transcript (p far below 0), a term below one unit, a failed
        // check -- goes through em_term
        auto term = [&](uint32_t off, uint32_t w) {
            const int p = xe + em_word_E(w);
            char *const slot = (char *)s_acc + rep_off + off;
            if ((uint32_t)p < (uint32_t)em_word_cap(w)) {
                const uint64_t yh = x >> (~(uint32_t)p & 63u), yl = (x2 << ((uint32_t)p & 63u)) >> (w & 63u);
                atomicAdd((unsigned long long *)slot, (unsigned long long)yh);
                atomicAdd((unsigned long long *)(slot + LO_OFF), (unsigned long long)yl);
            } else if (p > -16384) {
                uint64_t yh, yl;
                if (em_term(x, xe, w, acc, yh, yl)) {
                    atomicAdd((unsigned long long *)slot, (unsigned long long)yh);
                    atomicAdd((unsigned long long *)(slot + LO_OFF), (unsigned long long)yl);
                }
            }
        };
        auto sw = [&](uint32_t off) { return *(const uint32_t *)((const char *)s_w + off); };
        // one group of four hits: the four scale words are requested together, then the terms
        auto give4 = [&](uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3) {
            if (MEASURE) {
                atomicMax((int32_t *)((char *)s_xe + (o0 >> 1)), xe);
                atomicMax((int32_t *)((char *)s_xe + (o1 >> 1)), xe);
                atomicMax((int32_t *)((char *)s_xe + (o2 >> 1)), xe);
                atomicMax((int32_t *)((char *)s_xe + (o3 >> 1)), xe);
            } else {
                uint32_t w0 = sw(o0), w1 = sw(o1), w2 = sw(o2), w3 = sw(o3);
                asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3)); // one wait for the four
                term(o0, w0); term(o1, w1); term(o2, w2); term(o3, w3);
            }
        };
#define EMS_GIVE(i)                                                                                          \
        if ((uint32_t)i < ng) { const uint32_t v = bf.g##i; give4(EMS_OFF0(v), EMS_OFF1(v), EMS_OFF2(v), EMS_OFF3(v)); }
        EMS_GROUPS(EMS_GIVE)
#undef EMS_GIVE
#pragma unroll 1
        for (uint32_t g = 8; g < ng; ++g) {
            const uint32_t v = src[(size_t)g * 64];
            give4(EMS_OFF0(v), EMS_OFF1(v), EMS_OFF2(v), EMS_OFF3(v));
        }
        for (uint32_t f = 0; f < Lf; ++f) { // (Lf is 0 unless the tile is a far tile)
            const uint32_t c = farp[(size_t)f * 64];
            if (MEASURE) {
                atomicMax(&a.xe[c], xe);
            } else {
                uint64_t yh, yl;
                if (em_term(x, xe, a.word[c], acc, yh, yl)) {
                    if (yh) atomicAdd((unsigned long long *)&a.hi[c], (unsigned long long)yh);
                    if (yl) atomicAdd((unsigned long long *)&a.lo[c], (unsigned long long)yl);
                }
            }
        }
    };
#undef EMS_GROUPS
#undef EMS_OFF0
#undef EMS_OFF1
#undef EMS_OFF2
#undef EMS_OFF3

    auto slow_tile = [&](const SellTile &d) {
        const uint32_t wbase = d.wbase;
        if (lane >= d.nrows()) return;
        const uint64_t st = (uint64_t)row_ptr[d.r0 + lane];
        const uint32_t L = (uint32_t)((uint64_t)row_ptr[d.r0 + lane + 1] - st);
        if (L == 0) return;
        const uint32_t *cl = col_idx + st;
        double dsum = 0.0;
        for (uint32_t j = 0; j < L; ++j) {
            const uint32_t c = cl[j], dd = c - wbase;
            dsum += dd < (uint32_t)WIN ? s_mu[dd] : a.mu[c];
        }
        uint64_t x;
        int xe;
        if (!em_row_head<MEASURE>(dsum, HAS_K ? kmult[d.r0 + lane] : 1u, acc, x, xe)) return;
        for (uint32_t j = 0; j < L; ++j) {
            const uint32_t c = cl[j], dd = c - wbase;
            const bool in = dd < (uint32_t)WIN;
            if (MEASURE) {
                if (in) atomicMax(&s_xe[dd], xe);
                else atomicMax(&a.xe[c], xe);
            } else {
                uint64_t yh, yl;
                if (em_term(x, xe, in ? s_w[2 * dd] : a.word[c], acc, yh, yl)) {
                    if (in) {
                        atomicAdd((unsigned long long *)&s_hi[dd], (unsigned long long)yh);
                        atomicAdd((unsigned long long *)&s_lo[dd], (unsigned long long)yl);
                    } else {
                        if (yh) atomicAdd((unsigned long long *)&a.hi[c], (unsigned long long)yh);
                        if (yl) atomicAdd((unsigned long long *)&a.lo[c], (unsigned long long)yl);
                    }
                }
            }
        }
    };

    SellTile none;
    none.off16 = 0; none.r0 = 0; none.wbase = 0; none.meta = sell_meta(0, 0, SELL_EMPTY);
    auto tile_at = [&](uint64_t i) { return i < nt ? T[i] : none; };
    uint32_t cur_base = tile_at(0).wbase;
    uint64_t scan = 0; // tiles of the range already examined for a window slide (by this wave; same sequence in every wave)
    auto slide_to = [&](uint32_t base) {
        __syncthreads();           // every wave has finished the tiles of the old window
        flush_window(cur_base);
        load_window(base);
        cur_base = base;
        __syncthreads();
    };
    // g: index of the tile in the range (own tiles: wave, wave + W, ...)
    auto process = [&](const SellTile &d, uint64_t g, const SellTile &refill, Buf &bf) {
        if (g < nt) {
            for (; scan <= g; ++scan) {
                const SellTile q = scan == g ? d : T[scan];
                if (!(q.flags() & SELL_EMPTY) && q.wbase != cur_base) slide_to(q.wbase);
            }
        }
        if (d.flags() & SELL_EMPTY) { issue(refill, bf); return; }
        if (d.flags() & SELL_FAST) walk(d, bf, false);
        else if (d.flags() & SELL_FAR) walk(d, bf, true);
        else slow_tile(d);
        issue(refill, bf);
    };

    SellTile dA = tile_at(wave), dB = tile_at(wave + W);
    Buf bufA, bufB;
    load_window(cur_base);
    __syncthreads();
    issue(dA, bufA);
    __builtin_amdgcn_sched_barrier(0); // A's block is requested before B's, here as in the loop (sell_multi_kernels.h): the waits are by count
    issue(dB, bufB);
    for (uint64_t g = wave; g < nt; g += 2 * W) {
        const SellTile nA = tile_at(g + 2 * W), nB = tile_at(g + 3 * W);
        process(dA, g, nA, bufA);
        process(dB, g + W, nB, bufB);
        dA = nA;
        dB = nB;
    }
    for (; scan < nt; ++scan) { // slides after this wave's last tile: the other waves still need them
        const SellTile q = T[scan];
        if (!(q.flags() & SELL_EMPTY) && q.wbase != cur_base) slide_to(q.wbase);
    }
    __syncthreads();
    flush_window(cur_base);
    if (!MEASURE) em_acc_commit(acc, s_ll, a.ll);
}

} // namespace mmg
