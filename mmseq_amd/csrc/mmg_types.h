// mmg_types.h -- plain structs and constants shared by the kernels (device TUs) and the host side of the C ABI.
// No device code here: mmgibbs.hip fills these and hands them to the launchers declared in mmg_launch.h.
#pragma once
#include <stdint.h>
#include "../../include/mmgibbs.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MMG_TYPES_HD __host__ __device__
#else
#define MMG_TYPES_HD
#endif

namespace mmg {

constexpr uint32_t K_SMALL = MMG_K_SMALL;   // no row draws more than K_SMALL categoricals
constexpr uint32_t K_DRAWS_PER_HIT = MMG_K_DRAWS_PER_HIT;
// k categorical draws, or the conditional-binomial chain (L - 1 binomials)?  L = hits of the row (>= 1).  Draws while there are at most
// K_DRAWS_PER_HIT of them per binomial they replace, and never more than K_SMALL (spec version 8; a row of one hit needs neither).
MMG_TYPES_HD inline bool draws_categoricals(uint32_t k, uint32_t L)
{
    const uint64_t per_hit = (uint64_t)K_DRAWS_PER_HIT * (L - 1u);
    return k <= 1u || (uint64_t)k <= (per_hit < K_SMALL ? per_hit : (uint64_t)K_SMALL);
}
// the rows on the conditional-binomial chain (kclass 3 below) with something to draw: at least two hits, reads to allocate, and more of
// them than categorical draws pay for.  They are sampled from a list of their own (bigk_kernels.h), not by the tile kernel.
MMG_TYPES_HD inline bool bigk_row(uint32_t k, uint64_t L) { return L >= 2 && k > 0 && !draws_categoricals(k, (uint32_t)(L < 0xffffffffull ? L : 0xffffffffull)); }
// rows above K_SMALL sort by this bucket of k inside their class (7 bits: 8 steps per power of two from 64 on): the rows of a tile
// draw about equally often
MMG_TYPES_HD inline uint32_t k_bucket(uint32_t k)
{
    if (k <= K_SMALL) return 0;
    const uint32_t e = 31u - (uint32_t)__builtin_clz(k);          // >= 6
    const uint32_t b = 8u * (e - 6u) + ((k >> (e - 3u)) & 7u);
    return b < 127u ? b : 127u;
}

// ---- canonical layout (restated in oracle/binding.py:canonical_layout) -----------------------------
// Rows are exchangeable in the model (src/mmseq.cpp:857-891 visits them in file order only because that is how they were
// read), and a row is a SET of transcripts (the reference walks it in ascending order, :871).  The library puts every row's hits in
// ascending order and stores the rows sorted by row_key, ties by the tie word below, then by the caller's position.
//   step 0  a row that draws k >= 2 categoricals (draws_categoricals: k <= min(K_SMALL, K_DRAWS_PER_HIT (hits - 1))) is stored as k rows
//           with k = 1 (layout.hip: layout_expand_rows): identical reads -- what every collapsed hits file carries (src/mmseq.cpp:409-418) --
//           then run on the register path and in fused chain pairs like any other read, instead of through the multiplicity kernel (50M-read
//           file shape: 0.39 -> 0.34 ms per sweep, 8 chains 2.5k -> 4.0k chain-iterations/s); an all-ones k array is dropped.  Rows on the
//           conditional-binomial chain, rows of one hit and k = 0 stay.  The step is skipped when it would store more than
//           LAYOUT_EXPAND_MAX_RATIO rows per uploaded row (a heavily collapsed file would be un-collapsed: memory and work would scale
//           with the reads again) or 2^32 rows: such rows keep their k and draw their categoricals in the tile kernel.
//   lead    = smallest transcript of the row >> LAYOUT_BAND_SHIFT       (bands of 64 consecutive transcripts)
//   near    = every hit of the row lies in [lead * 64, lead * 64 + LAYOUT_NEAR_SPAN) and the row has <= 255 hits
//   band    = lead for a near row; for a far row its HOME band: max(median hit >> LAYOUT_BAND_SHIFT, 1) - 1 (lower median,
//             hit[(len - 1) / 2] of the ascending row): the window starting one band below the row's middle holds its bulk
//   kclass  = 0 (k <= 1), 1 (k categorical draws: draws_categoricals), 3 (conditional-binomial chain, or k >= 2 on a row of one hit)
//   key     = !near << 63 | band << 18 | kclass << 16 | (kclass == 1 ? k : k_bucket(k)) << 9 | min(len, 0x1ff)      (an empty row: key 0)
//   hash    = fold of (len, k, the ascending hits)
//   csum    = sum of (hit - band * 64) over the hits inside the window [band * 64, band * 64 + SELL_WIN)  (<= 255 * 254 < 2^16)
//   tie     = csum << 48 | hash >> 16: rows of equal key are ordered by their CENTRE first, then by content.  At every step of
//             its walk a tile's 64 lanes then gather neighbouring window slots -- neighbouring slots sit in different LDS banks,
//             equal slots broadcast -- instead of 64 random ones: measured LDS bank-conflict cycles of K1 50.1 M -> 24.9 M per
//             launch at config 3 (profiles/r03_*), a model of the bank rule predicts 4.1 -> 2.5 cycles per gather.
// A FAR row is stored with the hits inside its home window [band * 64, band * 64 + SELL_WIN) first and the others behind them, each part
// ascending: the stored order of a row's hits is the order every kernel and the oracle add its weights in.
// A tile of the sliced-ELL stream never crosses a (near, band) boundary.  Near tiles: all hits fall into ONE 255-wide LDS window
// starting at band * 64, rows of (nearly) equal lengths.  Far tiles: the window part of every row is encoded exactly like a near
// tile, the other hits are transcript ids in a per-lane far list behind it.
constexpr uint64_t LAYOUT_EXPAND_MAX_RATIO = 8; // step 0 happens only while it stores at most this many rows per row of the upload
constexpr uint32_t LAYOUT_BAND_SHIFT = 6;
constexpr uint32_t LAYOUT_NEAR_SPAN = 240;
constexpr uint32_t SELL_WIN = 255;          // transcripts per window; slot 255 holds 0.0
constexpr uint64_t LAYOUT_KEY_BAND_MASK = (1ull << 45) - 1; // key >> 18 & mask = band

// One tile of consecutive rows (k_tile_desc)
struct TileDesc {
    uint64_t nz0;   // first hit of the tile in col_idx
    uint64_t r0;    // first row
    uint32_t nrows;
    uint32_t nnz;   // CSR tiles: > tile capacity <=> a single long row handled by the slow path
    uint32_t cmin;  // leading transcript of the first non-empty row
    uint32_t clast; // leading transcript of the last non-empty row
    uint32_t cmax;  // largest transcript id in the tile
    uint32_t call;  // smallest transcript id in the tile
    uint32_t nnz4;  // hits when every row is padded to a multiple of 4
    uint32_t maxlen;// longest row of the tile
    uint32_t kmax;  // largest multiplicity in the tile (1 without a k array)
    uint32_t knot1; // rows whose multiplicity is not 1
};

// tile flags of the sliced-ELL sample and EM kernels
// FAST: register path (all hits in the window).  FAR: the block of a fast tile for the rows' window hits (64 count bytes, ng groups),
// then 64 far-count bytes and nf groups of 64 lanes x u32 transcript ids (the lane's hits outside the window, in stored order); the
// sample kernel walks the window part on the register path and adds the far list behind it.  Neither: the rows are walked from the
// CSR (rows of more than 255 hits, rows kept in an order that is not "window hits first").
// HASK: some row of the tile has a multiplicity other than 1.  A problem with multiplicities is sampled by TWO launches, each over its own
// list of tile descriptors: k_sample_sell<.., false> walks the tiles without the flag exactly as it walks a problem without a k array
// (identical reads are a minority, and the canonical order groups them: key field kclass), k_sample_sell<.., true> the flagged ones --
// but for their rows on the conditional-binomial chain, which a THIRD launch samples from a list (k_sample_bigk, bigk_kernels.h).
enum : uint32_t { SELL_FAST = 1, SELL_FAR = 2, SELL_EMPTY = 4, SELL_HASK = 8 };

struct SellTile {
    uint64_t off16;   // 16-byte-unit offset of the tile's block in the stream
    uint64_t r0;      // first row
    uint32_t wbase;   // LDS window base in force while this tile is walked
    uint32_t meta;    // nrows (<= 64) | ng << 8 (groups of 4 hits stored for every lane: longest row of the tile) | flags << 16 | nf << 24
    MMG_TYPES_HD uint32_t nrows() const { return meta & 0xffu; }
    MMG_TYPES_HD uint32_t ng() const { return (meta >> 8) & 0xffu; }
    MMG_TYPES_HD uint32_t flags() const { return (meta >> 16) & 0xffu; }
    MMG_TYPES_HD uint32_t nf() const { return meta >> 24; } // far tiles: entries per lane in the far list (most far hits of a row)
};  // dwords only: the descriptors are fetched with scalar loads
MMG_TYPES_HD inline uint32_t sell_meta(uint32_t nrows, uint32_t ng, uint32_t flags, uint32_t nf = 0) { return nrows | (ng << 8) | (flags << 16) | (nf << 24); }

// Replicas of the global count vectors.  A workgroup flushes the counts of its LDS window with one agent-scope atomic per touched
// transcript, and the workgroups that walk one band run side by side: the counts of a popular transcript are hit by hundreds of
// atomics to ONE address, which the memory side serialises (a read shard whose rows give a third of their reads to one transcript took
// 2.7 x the time of its neighbours; with the flush disabled every shard took the same).  Workgroup b adds into replica
// b % CNT_REPLICAS -- neighbouring ranges, different addresses -- and K2 sums the replicas (integer sums: any order, same bits).
// Replicas cost K2 eight count loads and stores per transcript instead of one (+1.5 us at 200 k transcripts), and they pay only where
// many ranges share a band -- read shards, small problems over many transcripts: a problem uses them when its launch has at least
// CNT_REPLICA_RANGES_PER_BAND ranges per run of equal band (mmg_problem::cnt_replicas; the whole config-3 problem has 7: one vector).
constexpr uint32_t CNT_REPLICAS = 8;
constexpr uint64_t CNT_REPLICA_RANGES_PER_BAND = 12;

struct SampleArgs {
    uint64_t seed;
    uint64_t row_id_base;
    uint64_t cnt_rep_stride; // elements between two replicas of the count vectors ([replicas][chains][n])
    uint32_t cnt_rep_mask;   // replicas - 1 (1 or CNT_REPLICAS replicas)
    uint32_t n;
    uint32_t chain;
    uint32_t iter;
};

struct UpdateArgs {
    int32_t *cnt;          // [cnt_replicas][C][n]  replicas summed, then zeroed
    int32_t *cnt_last;     // [C][n]
    const double *scale;   // n : 1/(beta + l[t])
    double *mu;            // [C][n]
    double *trace;         // [C][trace_len][n] or nullptr
    double *sum_log;       // [C][n]
    double *sum_log2;      // [C][n]
    const uint32_t *ext_of_int; // n: the caller's id of device transcript t (keys the Gamma stream), or nullptr = identity
    uint64_t seed;
    uint64_t cnt_rep_stride; // elements between two replicas of cnt (= n_chains * n)
    uint32_t cnt_replicas;   // 1 or CNT_REPLICAS
    double alpha;
    uint32_t n;
    uint32_t n_chains;
    uint32_t chain_base;
    uint32_t iter;
    int32_t sample_idx;    // >= 0: keep this iteration as trace sample; -1: not kept
    uint32_t trace_len;
};

struct EmOut {
    double loglik;
    uint64_t flag;
};

struct EmArgs {
    uint32_t n;
    const double *mu;      // n
    const uint32_t *word;  // n   packed scale words
    uint64_t *hi, *lo;     // n   accumulators (accumulate pass)
    int32_t *xe;           // n   max ilogb(x_i) (measure pass)
    uint64_t *ll;          // [0] LLH  [1] LLL  [2] repeat flag
};

struct SynthArgs {
    uint64_t seed, row0, rows;
    uint32_t n;
    int32_t uniform;
    const double *cdf;     // n   inclusive running sum of theta*efflen
    const double *len_cdf; // 99  Poisson(avg-1) inclusive cdf
    // gene-block mode (mmg_synth_desc.gene_size > 0): the hits of a read lie inside its gene (gene_size consecutive transcripts); a far
    // hit goes to a gene of the read's PARALOGUE FAMILY (far_family genes, scattered over the transcriptome by the bijection
    // g -> fam_a * g mod n_genes) or, with far_family = 0, anywhere
    uint32_t gene_size, far_family, n_genes, fam_a, fam_ainv;
};

// the family bijection of the gene-block generator: a multiplier coprime to n_genes derived from the seed, and its inverse
MMG_TYPES_HD inline void synth_family_params(uint64_t seed, uint32_t n_genes, uint32_t *a, uint32_t *ainv)
{
    if (n_genes < 3) { *a = 1; *ainv = 1 % (n_genes ? n_genes : 1); return; } // (2 genes: no multiplier in [2, 2] is coprime to 2 -- the search below would not end: the identity)
    uint64_t x = (seed + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
    x ^= x >> 31;
    uint64_t m = 2 + x % (n_genes - 1 ? n_genes - 1 : 1);
    for (;; m = 2 + (m - 1) % (n_genes - 1 ? n_genes - 1 : 1)) { // smallest candidate from m on that is coprime to n_genes
        uint64_t u = m, v = n_genes;
        while (v) { const uint64_t t = u % v; u = v; v = t; }
        if (u == 1) break;
    }
    int64_t t0 = 0, t1 = 1, r0 = n_genes, r1 = (int64_t)m; // extended Euclid: t1 * m == r1 (mod n_genes)
    while (r1 != 1) { const int64_t q = r0 / r1, r2 = r0 - q * r1, t2 = t0 - q * t1; r0 = r1; r1 = r2; t0 = t1; t1 = t2; }
    *a = (uint32_t)m;
    *ainv = (uint32_t)(((t1 % (int64_t)n_genes) + (int64_t)n_genes) % (int64_t)n_genes);
}

// geometry of the CSR fallback kernel k_sample: tiles of <= K1C_ELEMS - 8 hits and <= K1C_ROWS rows
constexpr int K1C_ELEMS = 2560, K1C_WIN = 256, K1C_UNR = 4, K1C_BS = 128, K1C_ROWS = 128;

} // namespace mmg
