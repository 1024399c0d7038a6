// mmg_launch.h -- what the host side of the C ABI (mmgibbs.hip) sees of the device translation units.
// Every launcher enqueues on the given stream and returns; the caller checks hipGetLastError().
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "mmg_types.h"

namespace mmg {

// ---- k1.hip: the sample kernels (src/mmseq.cpp:857-891) and their stream builders
const void *k1_sell_kernel(bool idx64, bool has_k, bool fixed_walk = false);   // k_sample_sell, 64 threads per workgroup
const void *k1_sell_far_kernel(bool idx64);            // k_sample_sell for the list of far / CSR-walked tiles (far list prefetched), no multiplicities
const void *k1_sell_multi_kernel(bool idx64, int nch); // k_sample_sell_multi for nch = 2 or 4 chains (no multiplicities), 64 threads
const void *k1_csr_kernel(bool idx64, bool has_k);    // k_sample, K1C_BS threads per workgroup
const void *k1_bigk_kernel(bool idx64);                 // k_sample_bigk: the rows on the conditional-binomial chain, from their list (64 threads)
void launch_tile_desc(bool idx64, const void *row_ptr, const uint32_t *col, const uint32_t *kmult, const uint64_t *tile_row, uint64_t n_tiles,
                      TileDesc *out, hipStream_t s);
void launch_tile_far(bool idx64, const void *row_ptr, const uint32_t *col, const uint64_t *key, const uint64_t *tile_row,
                     const uint32_t *cand, uint64_t n_cand, uint32_t *out /* [3][n_cand] */, hipStream_t s);
void launch_encode_sell(bool idx64, const void *row_ptr, const uint32_t *col, const SellTile *tiles, uint64_t n_tiles,
                        uint8_t *stream, hipStream_t s);

// ---- em.hip: EM sweeps (src/mmseq.cpp:741-811)
const void *em_sell_kernel(bool idx64, bool has_k, bool measure); // k_em_sell, EM_SELL_BS threads per workgroup
constexpr int EM_SELL_W = 2, EM_SELL_REP = 4; // waves per workgroup (one window), accumulator replicas
constexpr unsigned EM_SELL_BS = 64 * EM_SELL_W;
void launch_em_rows_global(bool idx64, bool measure, const void *row_ptr, const uint32_t *col, const uint32_t *k, uint64_t m,
                           EmArgs a, hipStream_t s);
void launch_em_colcount(const uint32_t *col, uint64_t nnz, uint64_t *cnt, unsigned grid, hipStream_t s);
void launch_fill_i32(int32_t *p, uint32_t n, int32_t v, hipStream_t s);
void launch_combine(void *dst, const void *src, size_t n, bool max_i32, hipStream_t s); // dst[i] += src[i] (uint64) or max (int32), one device
void launch_em_prepare(uint32_t n, const double *mu, const double *l, const uint64_t *colcnt, const int32_t *ref, int measured,
                       uint32_t *word, uint64_t *hi, uint64_t *lo, double *partial, uint64_t *ll, const uint32_t *int_of_ext,
                       hipStream_t s);
void launch_em_check(uint32_t n, const uint32_t *word, const uint64_t *hi, uint64_t *ll, hipStream_t s);
void launch_em_finish(const double *partial, uint32_t np, const uint64_t *ll, EmOut *out, hipStream_t s);
void launch_em_apply(uint32_t n, double *mu, const double *l, const uint32_t *word, const uint64_t *hi, const uint64_t *lo,
                     int32_t *sexp, hipStream_t s);

// ---- misc.hip: Gamma redraw / trace (src/mmseq.cpp:896-917), read-out, start values, generator, self tests
void launch_update(const UpdateArgs &a, hipStream_t s, hipEvent_t start = nullptr, hipEvent_t stop = nullptr); // events: the launch's own time stamps (hipExtLaunchKernel)
// out[t_ext * S + smp] = in[smp * n + int_of_ext[t_ext]]   (int_of_ext == nullptr: identity)
void launch_transpose(const double *in, double *out, uint32_t n, uint32_t S, const uint32_t *int_of_ext, hipStream_t s);
// out[r * n + t_ext] = in[r * n + int_of_ext[t_ext]] for r < rows (element size 4 or 8 bytes)
void launch_gather_rows(const void *in, void *out, uint32_t n, uint32_t rows, int elem_bytes, const uint32_t *int_of_ext, hipStream_t s);
void launch_fold_counts(int32_t *cnt, uint64_t stride, size_t n, hipStream_t s); // replicas 1.. of the count vectors added into replica 0, zeroed
void launch_add_i32(int32_t *dst, const int32_t *src, size_t n, hipStream_t s); // dst[i] += src[i], one device
void launch_start_values(bool idx64, const void *row_ptr, const uint32_t *col, const uint32_t *k, uint64_t m, uint32_t n,
                         uint64_t *acc3, int32_t *unique_hits, hipStream_t s);
void launch_synth_len(const SynthArgs &a, double far_fraction, uint32_t *lens, hipStream_t s);
void launch_synth_fill(const SynthArgs &a, double far_fraction, const uint64_t *row_ptr, uint32_t *col, hipStream_t s);
void launch_selftest_math(int64_t n, const double *x, double *ol, double *oe, double *os, double *orc, hipStream_t s);
void launch_selftest_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out, hipStream_t s);
void launch_selftest_gamma(uint64_t seed, double shape, double scale, int64_t n, double *out, hipStream_t s);
void launch_selftest_binomial(uint64_t seed, uint32_t nn, double p, int64_t n, uint32_t *out, hipStream_t s);
void launch_selftest_btrs_pretest(uint64_t seed, int64_t n_cases, double n_lo, double n_hi, unsigned long long *counts /* [5], device */, hipStream_t s);
void launch_selftest_binv_pretest(uint64_t seed, int64_t n_cases, double n_lo, double n_hi, float slack, unsigned long long *counts /* [5], device */, hipStream_t s);
// host instantiations of the same inline code (device == -1 paths of the self tests, synthetic transcript tables)
void host_math(int64_t n, const double *x, double *ol, double *oe, double *os, double *orc);
void host_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out);
void host_gamma(uint64_t seed, uint32_t tag, uint64_t id0, int by_iter, double shape, double scale, int64_t n, double *out);
void host_binomial(uint64_t seed, uint32_t nn, double p, int64_t n, uint32_t *out);
void host_synth_tables(uint64_t seed, uint32_t T, double lambda, std::vector<double> &efflen, std::vector<double> &cdf,
                       std::vector<double> &len_cdf);
// exact start-value share limbs (host side of k_start_values)
double host_start_value(uint64_t a0, uint64_t a1, uint64_t a2, double l);

// ---- order.hip: a transcript order from the hit graph when the caller passes none
// sorted, duplicate-free directed edges u << 32 | v (both directions) of the co-occurrence graph of a content-hash sample of the rows
hipError_t order_cooccurrence_edges(bool idx64, uint64_t m, uint64_t nnz, const void *d_rp, const uint32_t *d_col, std::vector<uint64_t> &edges, hipStream_t s,
                                    const uint32_t *d_label = nullptr);
// pos[t] = position of transcript t in the derived order (level structures from pseudo-peripheral vertices; host code)
void order_from_edges(uint32_t n, const std::vector<uint64_t> &edges, std::vector<uint32_t> &pos, uint32_t hub_floor = 256);

// ---- layout.hip: the canonical row order (mmg_types.h) on the device
// Row keys of the CSR in its current order.  d_key: m u64 (caller frees).
hipError_t layout_map_cols(uint64_t nnz, uint32_t *d_col, const uint32_t *d_int_of_ext, hipStream_t s);
hipError_t layout_row_keys(uint64_t m, const uint64_t *d_rp, const uint32_t *d_col, const uint32_t *d_k, uint64_t *d_key, hipStream_t s);
// Sorts the CSR canonically: on return *d_rp / *d_col / *d_k (k may be nullptr) are NEW device buffers in canonical order (the
// old ones are freed), d_key holds the sorted keys.  col_pad: extra u32 slots allocated (zeroed) behind col.  m < 2^32.
hipError_t layout_canonical_sort(uint64_t m, uint64_t nnz, uint64_t **d_rp, uint32_t **d_col, uint32_t **d_k, uint64_t *d_key,
                                 size_t col_pad, hipStream_t s);
// Canonical layout, step 0: a row that draws k >= 2 categoricals becomes k rows with k = 1 (mmg_types.h: draws_categoricals).  *d_rp / *d_col / *d_k are replaced when
// any row expands; *m, *nnz follow; *d_k is freed and set to nullptr when no multiplicity other than 1 is left.
hipError_t layout_expand_rows(uint64_t *m, uint64_t *nnz, uint64_t **d_rp, uint32_t **d_col, uint32_t **d_k, size_t col_pad, hipStream_t s);
// Start rows of the maximal runs of equal (near, band) in d_key (ascending; first entry 0).  Empty if there are more than
// max_segments runs (rows in no useful order).
hipError_t layout_segments(uint64_t m, const uint64_t *d_key, uint64_t max_segments, std::vector<uint64_t> &starts, hipStream_t s);
// d_out[0..rows] = row offsets of rows [lo, lo + rows) of a stored problem, rebased to 0 (d_out on the SAME device as d_rp)
hipError_t layout_rebase_row_ptr(bool idx64, const void *d_rp, uint64_t lo, uint64_t rows, uint64_t *d_out, hipStream_t s);
// narrow u64 row offsets to u32 on the device
hipError_t layout_narrow_row_ptr(uint64_t m, const uint64_t *d_rp64, uint32_t *d_rp32, hipStream_t s);
// d_rp[0..m] = running sum of d_len[0..m)
hipError_t layout_scan_lens(uint64_t m, const uint32_t *d_len, uint64_t *d_rp, hipStream_t s);
hipError_t layout_max_row_len(uint64_t m, const uint64_t *d_rp, uint32_t *max_len, hipStream_t s);
// The stored positions (ascending) of the rows on the conditional-binomial chain (mmg_types.h: bigk_row): *d_list is a NEW device buffer
// of *n_list entries (nullptr / 0 without such rows or without multiplicities).  d_rp: the problem's row offsets (u32 or u64).
hipError_t layout_bigk_rows(bool idx64, uint64_t m, const void *d_rp, const uint32_t *d_k, uint64_t **d_list, uint64_t *n_list, hipStream_t s);

} // namespace mmg
