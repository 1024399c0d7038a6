// layout.hip -- device TU: the canonical row order of a problem (spec: mmg_types.h; restated in oracle/host_oracle.py).
//
// The reference keeps rows in first-seen order (src/mmseq.cpp:409-418), which is an accident of the input file: the model treats
// rows as exchangeable.  The library sorts them by (near/far, leading-transcript band, multiplicity class, length, content hash),
// stably, with rocPRIM radix sorts on the device, so that (a) a 64-row tile of the sliced-ELL stream touches one LDS window and
// holds rows of equal length, and (b) the stored order -- and with it the per-row random streams -- is a function of the SET of
// rows, not of the order the caller read them in.  Ties inside a key: the centre of the row (sum of its window offsets) before the
// content hash -- the lanes of a tile then gather neighbouring LDS slots (mmg_types.h).
#include <cstring>
#include <algorithm>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include "mmg_launch.h"

namespace mmg {

// SORT_HITS (canonical layout): a row's hits are put in ascending order first -- the order the reference's compressed_matrix
// iterates them in whatever order they were inserted (src/mmseq.cpp:871) -- so that key and hash are functions of the SET of hits.
template <bool SORT_HITS>
__global__ __launch_bounds__(256) void k_row_keys(uint64_t m, const uint64_t *__restrict__ rp, uint32_t *col,
                                                  const uint32_t *__restrict__ k, uint64_t *__restrict__ key,
                                                  uint64_t *__restrict__ hash, uint32_t *__restrict__ len)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    const uint64_t b = rp[r], e = rp[r + 1];
    const uint64_t L = e - b;
    const uint32_t kk = k ? k[r] : 1u;
    if (SORT_HITS) {
        bool ascending = true;
        for (uint64_t j = b + 1; j < e; ++j) ascending = ascending && col[j - 1] <= col[j];
        if (!ascending) { // rare (hits files list them sorted, src/bam2hits.cpp:271-300): sorted by the row's own thread
            uint32_t *a = col + b;
            if (L <= 64) { // insertion sort
                for (uint64_t j = 1; j < L; ++j) {
                    const uint32_t c = a[j];
                    uint64_t q = j;
                    while (q > 0 && a[q - 1] > c) { a[q] = a[q - 1]; --q; }
                    a[q] = c;
                }
            } else { // heap sort: a long unsorted row must not cost L^2
                auto sift = [&](uint64_t root, uint64_t end) {
                    for (;;) {
                        uint64_t child = 2 * root + 1;
                        if (child >= end) return;
                        if (child + 1 < end && a[child] < a[child + 1]) ++child;
                        if (a[root] >= a[child]) return;
                        const uint32_t t = a[root]; a[root] = a[child]; a[child] = t;
                        root = child;
                    }
                };
                for (uint64_t i = L / 2; i-- > 0;) sift(i, L);
                for (uint64_t end = L - 1; end > 0; --end) {
                    const uint32_t t = a[0]; a[0] = a[end]; a[end] = t;
                    sift(0, end);
                }
            }
        }
    }
    uint64_t h = 0x9E3779B97F4A7C15ull + L + ((uint64_t)kk << 32);
    uint32_t lo = 0xffffffffu, hi = 0;
    for (uint64_t j = b; j < e; ++j) {
        const uint32_t c = col[j];
        lo = min(lo, c);
        hi = max(hi, c);
        h = (h ^ (uint64_t)c) * 0xFF51AFD7ED558CCDull;
        h ^= h >> 32;
    }
    uint64_t kv = 0;
    if (L) {
        uint64_t band = lo >> LAYOUT_BAND_SHIFT;
        const bool near = L <= 255 && (uint64_t)hi - (band << LAYOUT_BAND_SHIFT) < LAYOUT_NEAR_SPAN;
        if (!near) { // home band: one below the band of the row's (lower) median hit
            uint32_t med = col[b + (L - 1) / 2];
            if (!SORT_HITS && L <= 4096) { // rows kept as given (e.g. a stored far row: window hits first): the median by rank, not by
                const uint64_t want = (L - 1) / 2; // position (rows longer than a tile can hold never become far tiles: any band does)
                for (uint64_t i = b; i < e; ++i) {
                    const uint32_t c = col[i];
                    uint64_t less = 0, equal = 0;
                    for (uint64_t j = b; j < e; ++j) { less += col[j] < c; equal += col[j] == c; }
                    if (less <= want && want < less + equal) { med = c; break; }
                }
            }
            const uint64_t mid = med >> LAYOUT_BAND_SHIFT;
            band = (mid > 1 ? mid : 1) - 1;
        }
        const uint64_t kclass = kk <= 1 ? 0 : (draws_categoricals(kk, (uint32_t)(L < 0xffffffffull ? L : 0xffffffffull)) ? 1 : 3);
        const uint64_t ksmall = kclass == 1 ? kk : k_bucket(kk); // rows that draw k times sort by k first: a tile draws max k times per lane;
                                                                 // rows of the binomial chain (class 3) never share more than one tile per band with them
        kv = ((uint64_t)(near ? 0 : 1) << 63) | (band << 18) | (kclass << 16) | (ksmall << 9) | (L < 0x1ff ? L : 0x1ff);
    }
    key[r] = kv;
    if (hash) {
        // tie order inside a key: the centre of the row's window hits first (LAYOUT_CSUM, mmg_types.h), then the content hash
        uint64_t cs = 0;
        if (L) {
            const uint32_t wbase = (uint32_t)(((kv >> 18) & LAYOUT_KEY_BAND_MASK) << LAYOUT_BAND_SHIFT);
            for (uint64_t j = b; j < e; ++j) { const uint32_t d = col[j] - wbase; cs += d < SELL_WIN ? d : 0u; }
        }
        h = (cs << 48) | (h >> 16);
    }
    if (hash) hash[r] = h;
    if (len) len[r] = (uint32_t)(L < 0xffffffffull ? L : 0xffffffffull);
}

__global__ __launch_bounds__(256) void k_iota(uint64_t m, uint32_t *idx)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < m) idx[r] = (uint32_t)r;
}

template <typename T>
__global__ __launch_bounds__(256) void k_gather(uint64_t m, const T *__restrict__ in, const uint32_t *__restrict__ idx, T *__restrict__ out)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < m) out[r] = in[idx[r]];
}

// stored row r <- caller row idx[r].  A far row (key bit 63) is stored with the hits inside its home window
// [band * 64, band * 64 + SELL_WIN) first and the others behind them, both in ascending order (mmg_types.h).
__global__ __launch_bounds__(256) void k_gather_csr(uint64_t m, const uint64_t *__restrict__ rp_old, const uint32_t *__restrict__ col_old,
                                                    const uint32_t *__restrict__ idx, const uint64_t *__restrict__ rp_new,
                                                    const uint64_t *__restrict__ key, uint32_t *__restrict__ col_new)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    const uint64_t s = rp_old[idx[r]], d = rp_new[r], L = rp_new[r + 1] - d;
    const uint64_t kv = key[r];
    if (!(kv >> 63)) {
        for (uint64_t j = 0; j < L; ++j) col_new[d + j] = col_old[s + j];
        return;
    }
    const uint32_t wbase = (uint32_t)(((kv >> 18) & LAYOUT_KEY_BAND_MASK) << LAYOUT_BAND_SHIFT);
    uint64_t o = d;
    for (uint64_t j = 0; j < L; ++j) { const uint32_t c = col_old[s + j]; if (c - wbase < SELL_WIN) col_new[o++] = c; }
    for (uint64_t j = 0; j < L; ++j) { const uint32_t c = col_old[s + j]; if (c - wbase >= SELL_WIN) col_new[o++] = c; }
}

__global__ __launch_bounds__(256) void k_segment_starts(uint64_t m, const uint64_t *__restrict__ key, uint64_t cap, uint64_t *out,
                                                        unsigned long long *count)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    // an empty row (key 0) belongs to the run around it: it neither starts a run nor ends one
    bool start = r == 0;
    if (!start && key[r] != 0) {
        uint64_t q = r;
        while (q > 0 && key[q - 1] == 0) --q;
        start = q == 0 ? false : (key[r] >> 18) != (key[q - 1] >> 18);
    }
    if (start) {
        const unsigned long long at = atomicAdd(count, 1ull);
        if (at < cap) out[at] = r;
    }
}

__global__ __launch_bounds__(256) void k_narrow(uint64_t m1, const uint64_t *__restrict__ in, uint32_t *__restrict__ out)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < m1) out[r] = (uint32_t)in[r];
}

__global__ __launch_bounds__(256) void k_max_len(uint64_t m, const uint64_t *__restrict__ rp, unsigned int *out)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t L = 0;
    if (r < m) { const uint64_t d = rp[r + 1] - rp[r]; L = (uint32_t)(d < 0xffffffffull ? d : 0xffffffffull); }
    for (int off = 32; off > 0; off >>= 1) L = max(L, (uint32_t)__shfl_xor((int)L, off));
    if ((threadIdx.x & 63) == 0 && L) atomicMax(out, L);
}

// caller transcript ids -> device ids, in place
__global__ __launch_bounds__(256) void k_map_cols(uint64_t nnz, uint32_t *__restrict__ col, const uint32_t *__restrict__ int_of_ext)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nnz; j += stride) col[j] = int_of_ext[col[j]];
}

static inline unsigned blocks_of(uint64_t n) { return (unsigned)((n + 255) / 256); }

hipError_t layout_map_cols(uint64_t nnz, uint32_t *d_col, const uint32_t *d_int_of_ext, hipStream_t s)
{
    if (nnz) hipLaunchKernelGGL(k_map_cols, dim3((unsigned)std::min<uint64_t>((nnz + 255) / 256, 65536)), dim3(256), 0, s, nnz, d_col, d_int_of_ext);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? hipStreamSynchronize(s) : e;
}

hipError_t layout_row_keys(uint64_t m, const uint64_t *d_rp, const uint32_t *d_col, const uint32_t *d_k, uint64_t *d_key, hipStream_t s)
{
    if (m) hipLaunchKernelGGL(k_row_keys<false>, dim3(blocks_of(m)), dim3(256), 0, s, m, d_rp, const_cast<uint32_t *>(d_col), d_k, d_key, (uint64_t *)nullptr, (uint32_t *)nullptr);
    return hipGetLastError();
}

hipError_t layout_scan_lens(uint64_t m, const uint32_t *d_len, uint64_t *d_rp, hipStream_t s)
{
    // d_rp[i] = sum of d_len[0..i) for i in [0, m]: an exclusive scan over m + 1 inputs whose last one is never read as a term
    if (m == 0) return hipMemsetAsync(d_rp, 0, sizeof(uint64_t), s);
    auto it = rocprim::make_transform_iterator(d_len, [] __device__(uint32_t x) { return (uint64_t)x; });
    size_t tmp = 0;
    hipError_t e = rocprim::inclusive_scan(nullptr, tmp, it, d_rp + 1, m, rocprim::plus<uint64_t>(), s);
    if (e != hipSuccess) return e;
    void *d_tmp = nullptr;
    if ((e = hipMalloc(&d_tmp, tmp ? tmp : 8)) != hipSuccess) return e;
    e = rocprim::inclusive_scan(d_tmp, tmp, it, d_rp + 1, m, rocprim::plus<uint64_t>(), s);
    if (e == hipSuccess) e = hipMemsetAsync(d_rp, 0, sizeof(uint64_t), s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_tmp);
    return e;
}

// ---- step 0 of the canonical layout: rows with a small multiplicity are stored as that many rows of multiplicity 1
__global__ __launch_bounds__(256) void k_expand_count(uint64_t m, const uint64_t *__restrict__ rp, const uint32_t *__restrict__ k, uint32_t *__restrict__ reps,
                                                      unsigned long long *stats)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t kk = r < m ? k[r] : 1u;
    const uint64_t L = r < m ? rp[r + 1] - rp[r] : 0;
    const bool ex = r < m && kk >= 2u && L >= 1 && draws_categoricals(kk, (uint32_t)(L < 0xffffffffull ? L : 0xffffffffull)); // (an empty row stays one row)
    if (r < m) reps[r] = ex ? kk : 1u;
    // one atomic per wave (an atomic per row had every row of a collapsed file queue up at two addresses: 25 ms for 5 M rows)
    const uint32_t n_ex = (uint32_t)__popcll(__ballot(ex)), n_stay = (uint32_t)__popcll(__ballot(r < m && !ex && kk != 1u));
    if ((threadIdx.x & 63u) == 0u) {
        if (n_ex) atomicAdd(&stats[0], (unsigned long long)n_ex);     // rows that expand
        if (n_stay) atomicAdd(&stats[1], (unsigned long long)n_stay); // multiplicities that stay (0, a single hit, or the conditional-binomial class)
    }
}

__global__ __launch_bounds__(256) void k_expand_rows(uint64_t m, const uint64_t *__restrict__ rp, const uint32_t *__restrict__ k,
                                                     const uint64_t *__restrict__ first /* m + 1: first stored copy of row r */,
                                                     uint32_t *__restrict__ len_new, uint32_t *__restrict__ k_new, uint32_t *__restrict__ src)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    const uint64_t a = first[r], b = first[r + 1];
    const uint64_t L = rp[r + 1] - rp[r];
    const uint32_t kk = k[r];
    for (uint64_t q = a; q < b; ++q) {
        len_new[q] = (uint32_t)(L < 0xffffffffull ? L : 0xffffffffull);
        k_new[q] = b - a > 1 ? 1u : kk;
        src[q] = (uint32_t)r;
    }
}

__global__ __launch_bounds__(256) void k_expand_cols(uint64_t m_new, const uint64_t *__restrict__ rp_old, const uint32_t *__restrict__ col_old,
                                                     const uint32_t *__restrict__ src, const uint64_t *__restrict__ rp_new, uint32_t *__restrict__ col_new)
{
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m_new) return;
    const uint64_t s0 = rp_old[src[q]], d0 = rp_new[q], L = rp_new[q + 1] - d0;
    for (uint64_t j = 0; j < L; ++j) col_new[d0 + j] = col_old[s0 + j];
}

hipError_t layout_expand_rows(uint64_t *m_io, uint64_t *nnz_io, uint64_t **d_rp, uint32_t **d_col, uint32_t **d_k, size_t col_pad, hipStream_t s)
{
    const uint64_t m = *m_io;
    if (m == 0 || !d_k || !*d_k) return hipSuccess;
    if (m >= 0xffffffffull) return hipSuccess;           // (the caller rejects such a problem for the canonical layout anyway)
    uint32_t *reps = nullptr, *len_new = nullptr, *k_new = nullptr, *src = nullptr, *col_new = nullptr;
    uint64_t *first = nullptr, *rp_new = nullptr;
    unsigned long long *d_stats = nullptr, stats[2] = {0, 0};
    hipError_t e = hipSuccess;
    auto done = [&](hipError_t rc) {
        for (void *x : {(void *)reps, (void *)len_new, (void *)k_new, (void *)src, (void *)col_new, (void *)first, (void *)rp_new, (void *)d_stats})
            if (x) (void)hipFree(x);
        return rc;
    };
#define X_TRY(expr) do { e = (expr); if (e != hipSuccess) return done(e); } while (0)
    X_TRY(hipMalloc((void **)&reps, m * 4));
    X_TRY(hipMalloc((void **)&d_stats, 16));
    X_TRY(hipMemsetAsync(d_stats, 0, 16, s));
    const unsigned g = blocks_of(m);
    hipLaunchKernelGGL(k_expand_count, dim3(g), dim3(256), 0, s, m, (const uint64_t *)*d_rp, (const uint32_t *)*d_k, reps, d_stats);
    X_TRY(hipGetLastError());
    X_TRY(hipMemcpyAsync(stats, d_stats, 16, hipMemcpyDeviceToHost, s));
    X_TRY(hipStreamSynchronize(s));
    if (stats[0] == 0) { // nothing to expand; an array of ones is no array
        if (stats[1] == 0) { (void)hipFree(*d_k); *d_k = nullptr; }
        return done(hipSuccess);
    }
    X_TRY(hipMalloc((void **)&first, (m + 1) * 8));
    X_TRY(layout_scan_lens(m, reps, first, s));
    uint64_t m_new = 0;
    X_TRY(hipMemcpy(&m_new, first + m, 8, hipMemcpyDeviceToHost));
    // A heavily collapsed file would be un-collapsed by this step: beyond LAYOUT_EXPAND_MAX_RATIO x the rows (or the row limit of one
    // device) the rows keep their multiplicities and the multiplicity kernel draws for them (k categoricals per row, the same draws).
    // The rule depends on the set of rows only, like everything else of the stored order.
    if (m_new > LAYOUT_EXPAND_MAX_RATIO * m || m_new >= 0xffffffffull) return done(hipSuccess);
    X_TRY(hipMalloc((void **)&len_new, m_new * 4));
    X_TRY(hipMalloc((void **)&k_new, m_new * 4));
    X_TRY(hipMalloc((void **)&src, m_new * 4));
    hipLaunchKernelGGL(k_expand_rows, dim3(g), dim3(256), 0, s, m, (const uint64_t *)*d_rp, (const uint32_t *)*d_k, (const uint64_t *)first, len_new, k_new, src);
    X_TRY(hipGetLastError());
    X_TRY(hipMalloc((void **)&rp_new, (m_new + 1) * 8));
    X_TRY(layout_scan_lens(m_new, len_new, rp_new, s));
    uint64_t nnz_new = 0;
    X_TRY(hipMemcpy(&nnz_new, rp_new + m_new, 8, hipMemcpyDeviceToHost));
    X_TRY(hipMalloc((void **)&col_new, (nnz_new + col_pad) * 4));
    X_TRY(hipMemsetAsync(col_new + nnz_new, 0, col_pad * 4, s));
    hipLaunchKernelGGL(k_expand_cols, dim3(blocks_of(m_new)), dim3(256), 0, s, m_new, (const uint64_t *)*d_rp, (const uint32_t *)*d_col, (const uint32_t *)src,
                       (const uint64_t *)rp_new, col_new);
    X_TRY(hipGetLastError());
    X_TRY(hipStreamSynchronize(s));
    (void)hipFree(*d_rp); *d_rp = rp_new; rp_new = nullptr;
    (void)hipFree(*d_col); *d_col = col_new; col_new = nullptr;
    (void)hipFree(*d_k);
    if (stats[1] == 0) *d_k = nullptr;                     // every multiplicity expanded away
    else { *d_k = k_new; k_new = nullptr; }
    *m_io = m_new;
    *nnz_io = nnz_new;
#undef X_TRY
    return done(hipSuccess);
}

static hipError_t sort_pairs(uint64_t m, uint64_t *k_in, uint64_t *k_out, uint32_t *v_in, uint32_t *v_out, hipStream_t s)
{
    size_t tmp = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp, k_in, k_out, v_in, v_out, m, 0, 64, s);
    if (e != hipSuccess) return e;
    void *d_tmp = nullptr;
    if ((e = hipMalloc(&d_tmp, tmp ? tmp : 8)) != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(d_tmp, tmp, k_in, k_out, v_in, v_out, m, 0, 64, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_tmp);
    return e;
}

hipError_t layout_canonical_sort(uint64_t m, uint64_t nnz, uint64_t **d_rp, uint32_t **d_col, uint32_t **d_k, uint64_t *d_key,
                                 size_t col_pad, hipStream_t s)
{
    if (m == 0) return hipSuccess;
    uint64_t *hash = nullptr, *k2 = nullptr, *rp_new = nullptr;
    uint32_t *len = nullptr, *idx = nullptr, *idx2 = nullptr, *col_new = nullptr, *kk_new = nullptr, *len2 = nullptr;
    hipError_t e = hipSuccess;
    auto done = [&](hipError_t rc) {
        for (void *x : {(void *)hash, (void *)k2, (void *)rp_new, (void *)len, (void *)idx, (void *)idx2, (void *)col_new, (void *)kk_new, (void *)len2})
            if (x) (void)hipFree(x);
        return rc;
    };
#define L_TRY(expr) do { e = (expr); if (e != hipSuccess) return done(e); } while (0)
    L_TRY(hipMalloc((void **)&hash, m * 8));
    L_TRY(hipMalloc((void **)&k2, m * 8));
    L_TRY(hipMalloc((void **)&len, m * 4));
    L_TRY(hipMalloc((void **)&idx, m * 4));
    L_TRY(hipMalloc((void **)&idx2, m * 4));
    const unsigned g = blocks_of(m);
    hipLaunchKernelGGL(k_row_keys<true>, dim3(g), dim3(256), 0, s, m, (const uint64_t *)*d_rp, *d_col, d_k ? (const uint32_t *)*d_k : (const uint32_t *)nullptr, d_key, hash, len);
    hipLaunchKernelGGL(k_iota, dim3(g), dim3(256), 0, s, m, idx);
    L_TRY(hipGetLastError());
    // least significant first: content hash, then the key; both sorts are stable, so equal (key, hash) keep the caller's order
    L_TRY(sort_pairs(m, hash, k2, idx, idx2, s));
    hipLaunchKernelGGL(k_gather<uint64_t>, dim3(g), dim3(256), 0, s, m, (const uint64_t *)d_key, (const uint32_t *)idx2, hash); // hash buffer reused: keys in hash order
    L_TRY(hipGetLastError());
    L_TRY(sort_pairs(m, hash, d_key, idx2, idx, s)); // d_key: sorted keys, idx: stored row -> caller row
    (void)hipFree(hash); hash = nullptr;
    (void)hipFree(k2); k2 = nullptr;
    (void)hipFree(idx2); idx2 = nullptr;
    L_TRY(hipMalloc((void **)&len2, m * 4));
    hipLaunchKernelGGL(k_gather<uint32_t>, dim3(g), dim3(256), 0, s, m, (const uint32_t *)len, (const uint32_t *)idx, len2);
    L_TRY(hipGetLastError());
    L_TRY(hipMalloc((void **)&rp_new, (m + 1) * 8));
    L_TRY(layout_scan_lens(m, len2, rp_new, s));
    (void)hipFree(len); len = nullptr;
    (void)hipFree(len2); len2 = nullptr;
    L_TRY(hipMalloc((void **)&col_new, (nnz + col_pad) * 4));
    L_TRY(hipMemsetAsync(col_new + nnz, 0, col_pad * 4, s));
    hipLaunchKernelGGL(k_gather_csr, dim3(g), dim3(256), 0, s, m, (const uint64_t *)*d_rp, (const uint32_t *)*d_col, (const uint32_t *)idx,
                       (const uint64_t *)rp_new, (const uint64_t *)d_key, col_new);
    if (d_k && *d_k) {
        L_TRY(hipMalloc((void **)&kk_new, m * 4));
        hipLaunchKernelGGL(k_gather<uint32_t>, dim3(g), dim3(256), 0, s, m, (const uint32_t *)*d_k, (const uint32_t *)idx, kk_new);
    }
    L_TRY(hipGetLastError());
    L_TRY(hipStreamSynchronize(s));
    (void)hipFree(*d_rp); *d_rp = rp_new; rp_new = nullptr;
    (void)hipFree(*d_col); *d_col = col_new; col_new = nullptr;
    if (kk_new) { (void)hipFree(*d_k); *d_k = kk_new; kk_new = nullptr; }
#undef L_TRY
    return done(hipSuccess);
}

hipError_t layout_segments(uint64_t m, const uint64_t *d_key, uint64_t max_segments, std::vector<uint64_t> &starts, hipStream_t s)
{
    starts.clear();
    if (m == 0) return hipSuccess;
    uint64_t *d_out = nullptr;
    unsigned long long *d_count = nullptr;
    hipError_t e = hipMalloc((void **)&d_out, (max_segments + 1) * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_count, 8);
    if (e == hipSuccess) e = hipMemsetAsync(d_count, 0, 8, s);
    unsigned long long count = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_segment_starts, dim3(blocks_of(m)), dim3(256), 0, s, m, d_key, max_segments, d_out, d_count);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&count, d_count, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess && count <= max_segments) {
        starts.resize(count);
        e = hipMemcpy(starts.data(), d_out, count * 8, hipMemcpyDeviceToHost);
        std::sort(starts.begin(), starts.end());
    }
    if (d_out) (void)hipFree(d_out);
    if (d_count) (void)hipFree(d_count);
    return e;
}

// out[r] = in[lo + r] - in[lo] for r in [0, cnt]: the row offsets of a shard, from 32- or 64-bit offsets of the whole problem
template <typename IdxT>
__global__ __launch_bounds__(256) void k_rebase(const IdxT *__restrict__ in, uint64_t lo, uint64_t cnt1, uint64_t *__restrict__ out)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < cnt1) out[r] = (uint64_t)in[lo + r] - (uint64_t)in[lo];
}
hipError_t layout_rebase_row_ptr(bool idx64, const void *d_rp, uint64_t lo, uint64_t rows, uint64_t *d_out, hipStream_t s)
{
    if (idx64) hipLaunchKernelGGL(k_rebase<uint64_t>, dim3(blocks_of(rows + 1)), dim3(256), 0, s, (const uint64_t *)d_rp, lo, rows + 1, d_out);
    else hipLaunchKernelGGL(k_rebase<uint32_t>, dim3(blocks_of(rows + 1)), dim3(256), 0, s, (const uint32_t *)d_rp, lo, rows + 1, d_out);
    return hipGetLastError();
}

hipError_t layout_narrow_row_ptr(uint64_t m, const uint64_t *d_rp64, uint32_t *d_rp32, hipStream_t s)
{
    hipLaunchKernelGGL(k_narrow, dim3(blocks_of(m + 1)), dim3(256), 0, s, m + 1, d_rp64, d_rp32);
    return hipGetLastError();
}

hipError_t layout_max_row_len(uint64_t m, const uint64_t *d_rp, uint32_t *max_len, hipStream_t s)
{
    *max_len = 0;
    if (m == 0) return hipSuccess;
    unsigned int *d = nullptr;
    hipError_t e = hipMalloc((void **)&d, 4);
    if (e == hipSuccess) e = hipMemsetAsync(d, 0, 4, s);
    if (e == hipSuccess) { hipLaunchKernelGGL(k_max_len, dim3(blocks_of(m)), dim3(256), 0, s, m, d_rp, d); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(max_len, d, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (d) (void)hipFree(d);
    return e;
}

// ---- the list of the rows on the conditional-binomial chain (mmg_types.h: bigk_row; sampled by k_sample_bigk) -------------------
template <typename IdxT>
struct BigkRowPred {
    const IdxT *rp;
    const uint32_t *k;
    __device__ bool operator()(const uint64_t &r) const { return bigk_row(k[r], (uint64_t)rp[r + 1] - (uint64_t)rp[r]); }
};
template <typename IdxT>
__global__ __launch_bounds__(256) void k_bigk_count(uint64_t m, const IdxT *__restrict__ rp, const uint32_t *__restrict__ k, unsigned long long *count)
{
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool is = r < m && bigk_row(k[r], (uint64_t)rp[r + 1] - (uint64_t)rp[r]);
    const uint32_t c = (uint32_t)__popcll(__ballot(is));
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, (unsigned long long)c);
}
template <typename IdxT>
static hipError_t bigk_rows(uint64_t m, const IdxT *d_rp, const uint32_t *d_k, uint64_t **d_list, uint64_t *n_list, hipStream_t s)
{
    unsigned long long *d_n = nullptr;
    void *d_tmp = nullptr;
    uint64_t *out = nullptr;
    auto done = [&](hipError_t rc) { if (d_n) (void)hipFree(d_n); if (d_tmp) (void)hipFree(d_tmp); if (rc != hipSuccess && out) (void)hipFree(out); return rc; };
#define L_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return done(_e); } while (0)
    L_TRY(hipMalloc((void **)&d_n, 16));
    L_TRY(hipMemsetAsync(d_n, 0, 16, s));
    hipLaunchKernelGGL(k_bigk_count<IdxT>, dim3(blocks_of(m)), dim3(256), 0, s, m, d_rp, d_k, d_n);
    L_TRY(hipGetLastError());
    unsigned long long n = 0;
    L_TRY(hipMemcpyAsync(&n, d_n, 8, hipMemcpyDeviceToHost, s));
    L_TRY(hipStreamSynchronize(s));
    if (n == 0) return done(hipSuccess);
    L_TRY(hipMalloc((void **)&out, n * sizeof(uint64_t)));
    const BigkRowPred<IdxT> pred{d_rp, d_k};
    rocprim::counting_iterator<uint64_t> rows(0);
    size_t tmp = 0;
    L_TRY(rocprim::select(nullptr, tmp, rows, out, d_n + 1, (size_t)m, pred, s));
    L_TRY(hipMalloc(&d_tmp, tmp ? tmp : 8));
    L_TRY(rocprim::select(d_tmp, tmp, rows, out, d_n + 1, (size_t)m, pred, s));
    L_TRY(hipStreamSynchronize(s));
#undef L_TRY
    *d_list = out;
    *n_list = n;
    return done(hipSuccess);
}
hipError_t layout_bigk_rows(bool idx64, uint64_t m, const void *d_rp, const uint32_t *d_k, uint64_t **d_list, uint64_t *n_list, hipStream_t s)
{
    *d_list = nullptr;
    *n_list = 0;
    if (m == 0 || !d_k) return hipSuccess;
    return idx64 ? bigk_rows<uint64_t>(m, (const uint64_t *)d_rp, d_k, d_list, n_list, s) : bigk_rows<uint32_t>(m, (const uint32_t *)d_rp, d_k, d_list, n_list, s);
}

} // namespace mmg
