/*
 * mmgibbs.h -- C ABI of libmmgibbs.so: the MI355X-native Gibbs hot path of mmseq.
 *
 * The reference (eturro/mmseq) has no in-process plugin / FFI boundary: the hot path is
 * the body of main() in src/mmseq.cpp.  This header is the boundary a maintainer would
 * bind instead of that body; each entry point cites the reference lines it replaces
 * (paths relative to the reference tree).  INTEGRATION.md shows the call sequence that
 * replaces src/mmseq.cpp:833-925 inside the reference's own main().
 *
 * Conventions: every function returns 0 on success, non-zero on error (message via
 * mmg_last_error(), thread-local).  No exceptions cross the boundary.  All host
 * buffers are caller-owned; handles own their device memory.  One handle is driven by
 * one host thread at a time.  There is NO CPU fallback: without a HIP device every
 * compute entry point fails with MMG_ERR_NO_DEVICE.
 */
#ifndef MMGIBBS_H
#define MMGIBBS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMG_ABI_VERSION 8
/* Version history of the SPEC behind the entry points: under one version a chain is a pure function of (problem, tx_order, seed,
 * chain, iteration); a bump means the same inputs may yield different bits (golden fixtures and the oracle move with it).
 *   3  (round 2) rows with 2 <= k <= 64 draw k categoricals (before: k <= 8), sorted by k inside their class.  The constant moved
 *      without a bump at the time; recorded here.
 *   4  (round 3) canonical row order: ties inside a key are broken by the CENTRE of the row -- the sum of its hits' offsets in the
 *      window its key names -- before the content hash, so that the 64 rows of a tile gather neighbouring window slots (LDS bank
 *      conflicts of the sample and EM kernels: -50 %).  The per-row random stream follows the stored position, so chains of
 *      problems stored in the canonical layout differ from version 3; MMG_LAYOUT_KEEP_ROWS problems (and the committed golden
 *      chain, which keeps its rows) do not.  The draw target is now specified as fma(x, t 2^-32, t 2^-33) (mmg_math.h:
 *      draw_target): the same real number rounded once, bit-identical to version 3 unless t 2^-33 underflows.
 *      EM (mmg_em_*): the LO limb of a term is the fraction of x 2^E truncated to sl bits (version 3: rounded to nearest through an
 *      fp64 addition); both limbs of a term are now shifts of the significand of x.  mu after a sweep may differ from version 3 in
 *      the last bit.
 *   5  (round 3) multiplicities: a row draws its k categoricals one by one not only for k <= MMG_K_SMALL but whenever
 *      k <= MMG_K_DRAWS_PER_HIT * (hits - 1): a draw costs about 1/22 of one step of the conditional-binomial chain, which has hits - 1
 *      steps whatever k (measured: 2 M rows of 20 hits with k = 65 in 0.32 instead of 1.9 ms per sweep, with k = 300 in 1.46 instead of
 *      2.0).  Rows above MMG_K_SMALL are
 *      ordered by a logarithmic bucket of k inside their class (a tile loops to its largest k).  Chains of problems with
 *      MMG_K_SMALL < k <= MMG_K_DRAWS_PER_HIT * (hits - 1) rows differ from version 4.  The committed golden chain
 *      (tests/golden/keyed_chain_tiny.json) has no such row and is byte for byte what it was; keyed_chain_k_draws.json was added for
 *      the new range.  Also version 4/5 (recorded late): a row with 2 <= k <= MMG_K_SMALL stored as k rows changes
 *      mmg_problem_info.m / mmg_problem_download (the STORED problem), the low bits of mmg_problem_start_values (k terms
 *      floor(2^52 / hits) instead of floor(k 2^52 / hits)) and of EM sweeps (k terms 1/d instead of one term k/d) of such problems.
 *   6  (round 4) a canonical problem WITHOUT tx_order whose rows do not fit LDS windows in the caller's numbering (modelled cost more
 *      than 1.25 x that of register-path tiles alone) gets a transcript order derived from the hit graph (order.hip; a pure function
 *      of the set of rows and the caller's numbering) when the model prices the result a fifth lower: stored order, and with it the
 *      chain, of exactly those problems differ from version 5 (they ran the CSR-tile kernel at 1/29 of the speed).  Problems with
 *      tx_order, kept rows, or locality in the caller's numbering are untouched.  New entry points: mmg_problem_shard_bounds_timed,
 *      mmg_selftest_gibbs_shards; mmg_problem_shard_bounds cuts by modelled cost instead of hits (any cut gives the same chain).
 *      Also version 6 (recorded late): the canonical layout's step 0 (rows with 2 <= k <= MMG_K_SMALL stored k times) is skipped when it
 *      would store more than 8 rows per uploaded row (LAYOUT_EXPAND_MAX_RATIO, layout.hip; oracle/binding.py mirrors it): such rows
 *      keep their k and draw their categoricals in the multiplicity kernel -- stored rows and chain bits of heavily collapsed problems
 *      differ from version 5.
 *   7  (round 5) tx_order with GROUPS: the high 32 bits of a key name the transcript's group (the CLI passes the gene, src/mmseq.cpp:
 *      337-357).  When a canonical problem built on the caller's keys does not fit LDS windows (modelled cost more than 1.25 x that of
 *      register-path tiles alone: reads that also hit paralogues, whose genes the caller's gene order puts anywhere) the library
 *      derives an order of the GROUPS from the group-level hit graph (order.hip, the machinery of version 6 on groups) and keeps the
 *      problem built on it -- the transcripts of a group stay together, in the caller's order -- if the model prices it a fifth lower.
 *      Stored order and chain of exactly those problems differ from version 6; keys whose high words are all equal (plain ranks) are
 *      untouched.  mmg_synth_desc gained gene_size / far_family (the struct grew: ABI 7); MMG_OPT_WIRE_CHECK; the first exchanges of an
 *      mmg_group are verified against the host's own reduction of the members' buffers.
 *   8  (round 6) multiplicities: a row draws its k categoricals one by one while k <= min(MMG_K_SMALL, MMG_K_DRAWS_PER_HIT * (hits - 1))
 *      (version 5: the LARGER of the two limits); above, the conditional-binomial chain.  The chain's rows are no longer drawn in the
 *      lane of a tile that owns them but from a list of their own, a lane per row running ahead of its neighbours (k_sample_bigk), at a
 *      third of the old cost per binomial; k categorical draws beyond 64 then cost more than the hits - 1 binomials they were meant to
 *      save (bench.py `heavy`: 0.74 -> 0.40 ms per sweep, `collapsed`: 0.30 -> 0.15, of which the new boundary is 0.21 and 0.11).
 *      Step 0 of the canonical layout follows the rule: the rows stored k times are the rows that draw categoricals (rows of ONE hit and
 *      empty rows are never stored twice: they need no draw).  Chains and stored rows of problems with rows of
 *      MMG_K_SMALL < k <= MMG_K_DRAWS_PER_HIT * (hits - 1), with k > MMG_K_DRAWS_PER_HIT * (hits - 1) on rows of 2-4 hits, or with
 *      k >= 2 on rows of one hit differ from version 7.  tests/golden/keyed_chain_tiny.json has no such row and is byte for byte what
 *      it was; keyed_chain_k_draws.json was regenerated for the new boundary (tools/gen_golden.py).  The conditional binomials themselves
 *      (mmg_math.h: binomial) are unchanged, uniform for uniform.  MMG_OPT_BIGK_PER_WAVE, MMG_OPT_BIGK_SIDE_STREAM.
 *      Also version 8: in the GROUP-level hit graph of version 7 a group that shares rows with more than max(32, 8 x the median) other groups
 *      is a hub (version 7: max(256, ...), the transcript-level rule) and stays out of the traversal -- a gene whose repeat-bearing UTR
 *      collects reads of hundreds of genes no longer ties the paralogue families into one component.  Stored order and chain of problems
 *      with such groups differ from version 7. */
/* Layout.  The model does not care about the order of rows or the numbering of transcripts (src/mmseq.cpp:399-418 uses
 * first-seen order for both); the kernels do: they keep a window of consecutive transcripts in LDS and want the 64 rows of a
 * wave to have equal lengths.  mmg_problem_create therefore stores the rows in a CANONICAL order of its own (sorted on the
 * device by leading-transcript band, multiplicity class, length, row centre and a content hash -- a pure function of the set of rows, so
 * the chain does not depend on the order the caller happened to read them in) and, given tx_order, renumbers the transcripts
 * internally.  Every per-transcript array crossing this ABI stays in the CALLER's numbering; rows never cross it. */
#define MMG_LAYOUT_CANONICAL 0u /* default: rows re-ordered by the library                                      */
#define MMG_LAYOUT_KEEP_ROWS 1u /* rows kept as given (row i of the upload draws from random stream row_id_base + i):
                                   shards cut from an already canonical problem, tests of specific row orders    */

enum {
    MMG_OK = 0,
    MMG_ERR_ARG = 1,       /* invalid argument                                 */
    MMG_ERR_NO_DEVICE = 2, /* no usable HIP device (no fallback exists)        */
    MMG_ERR_HIP = 3,       /* a HIP runtime call failed                        */
    MMG_ERR_STATE = 4,     /* call sequence error (e.g. trace not kept)        */
    MMG_ERR_IO = 5
};

/* rows with k <= MMG_K_SMALL and k <= MMG_K_DRAWS_PER_HIT * (hits - 1) draw k categoricals; above either limit, a conditional-binomial
 * chain over the row's hits (one binomial per hit but the last, whatever k: src/mmseq.cpp:880's gsl_ran_multinomial).  The rows that
 * draw k >= 2 categoricals are stored k times by the canonical layout. */
#define MMG_K_SMALL 64u
#define MMG_K_DRAWS_PER_HIT 16u

typedef struct mmg_problem mmg_problem; /* device-resident CSR hit-set matrix M + k + l */
typedef struct mmg_sampler mmg_sampler; /* chains' state: mu, counts, trace, moments     */

/* Host description of the sparse problem: the reference's boolMat M (m x n), vector<int> k
 * and vector<double> l  (src/mmseq.cpp:117, :385, :593-608).  Rows are hit sets (or single
 * reads, k == NULL => all 1), columns the observed transcripts in first-seen order (:403);
 * within a row columns ascend (:412) -- the canonical layout sorts them itself, any order may be passed.
 * A column that occurs twice in a row counts twice (weight and pick): the reference's boolean matrix
 * cannot hold that case, its reader drops the second occurrence (:404-409) and so does the CLI of this
 * build; device and oracle agree on the list semantics (tools/fuzz_parity.py, `dups`). */
typedef struct mmg_problem_desc {
    uint64_t m;              /* rows                                                    */
    uint32_t n;              /* columns (observed transcripts)                          */
    const uint64_t *row_ptr; /* m+1 offsets into col_idx, row_ptr[0] == 0                */
    const uint32_t *col_idx; /* row_ptr[m] column indices                                */
    const uint32_t *k;       /* m multiplicities, or NULL for all ones                   */
    const double *l;         /* n: effective_length * mapped_reads / 1e9  (:603), > 0    */
    uint64_t row_id_base;    /* global index of stored row 0 (read-shard mode: the shard offset):
                                stored row i draws from random stream row_id_base + i      */
    uint32_t layout;         /* MMG_LAYOUT_*                                              */
    const uint64_t *tx_order;/* optional, n keys: transcripts are laid out on the device by ascending
                                (key, index).  A caller that knows which transcripts share reads (the
                                isoforms of a gene, src/mmseq.cpp:358) passes gene_ordinal << 32 | ordinal
                                within the gene, so that a read's hits are neighbours.  The high 32 bits of a key
                                name the transcript's GROUP: the library may reorder the groups (never the transcripts
                                inside one) when the rows do not fit LDS windows in the caller's group order -- reads
                                that also hit paralogues, spec version 7.  NULL: the caller's
                                numbering is the device numbering -- unless the rows do not fit LDS windows in it
                                (first-seen numbering, src/mmseq.cpp:399-408) and the layout is canonical: the library
                                then derives an order from the hit graph, a pure function of the set of rows
                                (spec version 6), and uses it like a caller's                 */
} mmg_problem_desc;

/* Synthetic problem generated directly into device CSR (no reference counterpart; the
 * benchmark inputs of BASELINE.md).  Every row is a pure function of (seed, row id). */
typedef struct mmg_synth_desc {
    uint64_t seed;        /* generator seed (1234 = reference default -seed)              */
    uint64_t rows;        /* rows generated on this device                                */
    uint64_t row0;        /* global id of the first row (shard offset)                    */
    uint32_t n;           /* transcripts                                                  */
    double avg_hits;      /* row length = min(100, 1 + Poisson(avg_hits - 1))             */
    int32_t uniform;      /* 0: hits inside a +-64 index window; 1: uniform over n        */
    int32_t sorted;       /* 1: canonical layout (MMG_LAYOUT_CANONICAL); 0: generator order kept
                             (a name-sorted BAM's order; MMG_LAYOUT_KEEP_ROWS)            */
    uint64_t mapped_reads;/* N in l = efflen * N / 1e9; 0 => rows                         */
    double far_fraction;  /* fraction of the rows (>= 2 hits) whose last drawn hit is replaced by a
                             transcript anywhere in [0, n): reads that also hit a paralogue */
    uint32_t gene_size;   /* 0: the band generator above.  G > 0 (ABI 7): GENE-BLOCK mode -- transcripts [gG, gG + G) are the isoforms of gene g
                             and the hits of a read lie inside its gene (row length <= G): what an aligner's output looks like, where
                             reads of different genes share nothing but paralogues (src/bam2hits.cpp:271-300)              */
    uint32_t far_family;  /* gene-block mode: 0 = a far hit goes anywhere (as above); F >= 2 = to an isoform of another gene of the read's
                             PARALOGUE FAMILY: the genes are grouped F at a time by a seeded bijection of the gene indices, so a family's
                             members lie anywhere in the transcriptome but are the same for every read of a gene                */
} mmg_synth_desc;

#define MMG_ORDER_SKIPPED 0x100
typedef struct mmg_problem_info {
    uint64_t m, nnz, total_k, row_id_base;
    uint32_t n;
    uint32_t max_row_len;
    uint64_t n_tiles;      /* tiles the sample kernel walks                               */
    uint64_t device_bytes; /* HBM held by the problem                                     */
    int32_t index_bits;    /* 32 or 64: width of the device row_ptr                       */
    int32_t sample_kernel; /* what mmg_sampler_sample launches: 2 k_sample_sell (sliced-ELL 8-bit stream; the default),
                              0 k_sample (32-bit CSR tiles: problems whose rows mostly span more than an LDS window) */
    uint64_t stream_bytes; /* bytes of the tile stream that kernel reads per launch (0 for kernel 0) */
    uint64_t fast_tiles;   /* sliced-ELL tiles walked from the register stream                               */
    uint64_t far_tiles;    /* sliced-ELL tiles with a far list (rows with hits outside the window); the rest
                              (n_tiles - fast - far - empty) are walked from the CSR                         */
    uint64_t padded_slots; /* hit slots of the sliced-ELL stream incl. padding (>= nnz of the fast tiles) */
    int32_t layout;        /* MMG_LAYOUT_* in force                                       */
    int32_t tx_renumbered; /* low byte -- 1: tx_order was given; 2: the library derived an order from the hit graph; 3: tx_order was given and
                              the library reordered its groups (the genes) by the group-level hit graph.  | MMG_ORDER_SKIPPED: such an order was
                              called for (spec versions 6 / 7) but the attempt could not be made -- it needs about 3 x the problem's device memory
                              free at the time, plus up to 4 GB -- or failed: the problem stands in the caller's order, and its chain is the one of
                              THAT order.  Under memory pressure a chain is therefore a pure function of (problem, tx_order, seed, chain,
                              iteration) AND of this flag; callers that need run-to-run identity check it (the CLI warns) */
    int32_t sample_grid;   /* workgroups of the sample kernel: the resident count (waves the runtime reports x CUs) times the
                              number of generations (1..16: tile ranges of about 24 tiles once the problem is large) */
    int32_t cu_count;
} mmg_problem_info;

/* Parameters of the Gibbs loop: alpha/beta are the Gamma prior (src/mmseq.cpp:184-185),
 * seed the reference's -seed (:203), gibbs_iter/trace_len as :190-192, :284. */
typedef struct mmg_config {
    double alpha, beta;
    uint64_t seed;
    int32_t n_chains;   /* independent chains advanced per sweep on this device (>= 1)   */
    int32_t chain_base; /* global index of chain 0 here (multi-device chains mode)        */
    int32_t gibbs_iter; /* planned chain length; sample s kept when iter % (gibbs_iter/trace_len) == 0 (:911) */
    int32_t trace_len;  /* samples per chain: 1024 in the reference (:191)                */
    int32_t keep_trace; /* !=0: store the samples (n*trace_len doubles per chain); 0: moments only */
    int32_t timing;     /* N > 0: bracket the kernel launches of every N-th iteration with HIP events; 0: none */
} mmg_config;

typedef struct mmg_timing {
    double sample_ms, update_ms; /* sums of per-launch HIP-event durations                */
    uint64_t sample_launches, update_launches;
} mmg_timing;

const char *mmg_last_error(void);
int mmg_abi_version(void);
/* number of HIP devices visible (0 and MMG_OK when none). */
int mmg_device_count(int *count);

/* ---- problem ------------------------------------------------------------------------ */
/* Uploads M, k, l (src/mmseq.cpp:456-462, :582-608 produce them) to `device`. */
int mmg_problem_create(const mmg_problem_desc *desc, int device, mmg_problem **out);
int mmg_problem_create_synthetic(const mmg_synth_desc *desc, int device, mmg_problem **out);
int mmg_problem_info_get(const mmg_problem *p, mmg_problem_info *info);
/* Copies the device CSR back in STORED order (row_ptr m+1 u64, col_idx nnz u32 in the caller's transcript numbering,
 * k m u32; any may be NULL): what a checker needs to replay the chain row by row -- stored row i walks its hits in the
 * returned order and draws from random stream row_id_base + i. */
int mmg_problem_download(const mmg_problem *p, uint64_t *row_ptr, uint32_t *col_idx, uint32_t *k);
/* Rows [lo, hi) of a stored problem as a problem of its own on `device`: a read shard (its rows draw from the random streams
 * row_id_base + lo + i, like the rows of the parent) or, with lo = 0 and hi = m, a replica for chains mode.  Cut on the parent's
 * device and copied device to device (peer copy); rows, hit order and transcript numbering stay as stored. */
int mmg_problem_shard(const mmg_problem *full, uint64_t lo, uint64_t hi, int device, mmg_problem **out);
/* Contiguous ranges of the stored rows of p for `parts` read shards: bounds[i] = first row of shard i, bounds[parts] = m.  The ranges
 * have (nearly) equal modelled COST of a sweep, not equal hit counts: the library knows what its tiles cost (a register-path tile by
 * the bytes of its block, a far tile three times that per far entry, multiplicity tiles by their draws), and the canonical order puts
 * every far row behind every near row -- the reference's static split of the rows (src/mmseq.cpp:864) would give the last devices the
 * expensive ones.  Boundaries are tile starts on even random-stream ids.  Problems on the CSR-tile kernel are cut by hits
 * (mmg_shard_bounds). */
int mmg_problem_shard_bounds(const mmg_problem *p, int parts, uint64_t *bounds);
/* The same cut by MEASURED cost: every candidate shard is run as what it will be -- the one-chain sample kernels over its interval of
 * p's tile lists, in ranges cut the way a problem of that size is cut, with replicated count vectors and the weights mu (n doubles, the
 * caller's numbering: the start values or the EM optimum) -- on p's device, timed with HIP events on the null stream; tiles of a shard
 * that ran long get dearer in proportion and the rows are cut again, until the slowest shard is within 2 % of the mean: at most 6
 * rounds of parts x 7 launches behind a 50 ms warm-up (0.09 s at 50 M reads, 0.2-0.3 s at 400 M).  What a model cannot know -- what a
 * far entry costs on this part, hit sets that give most of their reads to one transcript -- is in the measurement.  The chain does
 * not depend on where the rows are cut. */
int mmg_problem_shard_bounds_timed(const mmg_problem *p, const double *mu, int parts, uint64_t *bounds);
int mmg_problem_get_l(const mmg_problem *p, double *l);
/* Start values and the unique-hit column, src/mmseq.cpp:617-638: mu0[t] = sum_{i: t in row i}
 * k_i/|row i| / l[t]; unique_hits[t] = sum of k_i over rows {t} (bit-exact integer). Either
 * output may be NULL.  The shares are summed exactly (fixed point, 2^-52 resolution), so mu0 does not
 * depend on the order of the rows. */
int mmg_problem_start_values(const mmg_problem *p, double *mu0, int32_t *unique_hits);
/* EM to convergence from mu (in/out), src/mmseq.cpp:741-811: stop when the log-likelihood
 * gain <= epsilon or after max_iter sweeps. */
int mmg_problem_em(const mmg_problem *p, double *mu, int max_iter, double epsilon, int *iters,
                   double *loglik);

/* ---- EM, sweep by sweep (src/mmseq.cpp:741-811) ------------------------------------------
 * mmg_em_create uploads the start value and returns its log-likelihood (:745-754);
 * mmg_em_step performs one sweep of :761-806 and returns the new log-likelihood, so the caller
 * owns the loop, its stopping rule (:761) and its per-iteration output (:762-768).  mu stays on
 * the device between sweeps; mmg_em_get_mu downloads it (n doubles).  The sums are accumulated
 * exactly (fixed point, integer atomics): results do not depend on the traversal order. */
typedef struct mmg_em mmg_em;
int mmg_em_create(const mmg_problem *p, const double *mu0, mmg_em **out, double *loglik0);
int mmg_em_step(mmg_em *e, double *loglik);
int mmg_em_get_mu(mmg_em *e, double *mu);
/* sweeps done, rows passes repeated on measured scale exponents, rows-pass kernel (2 sliced-ELL stream,
 * 1 16-bit tile stream, 0 row per thread from the CSR) */
int mmg_em_stats(const mmg_em *e, int *sweeps, int *repeated_passes, int *stream_kernel);
void mmg_em_destroy(mmg_em *e);
/* Samplers and EM handles created from a problem must be destroyed before the problem. */
void mmg_problem_destroy(mmg_problem *p);

/* ---- sampler ------------------------------------------------------------------------ */
/* mu0: n doubles (every chain starts there, like mu = EM optimum at src/mmseq.cpp:820). */
int mmg_sampler_create(const mmg_problem *p, const mmg_config *cfg, const double *mu0, mmg_sampler **out);
/* Launch on a caller-owned hipStream_t (e.g. the framework's current stream) instead of
 * the sampler's own stream.  NULL restores the own stream (a non-blocking stream: NOT ordered against the legacy
 * default stream); to run on the legacy default stream itself pass hipStreamLegacy, (void *)1. */
int mmg_sampler_set_stream(mmg_sampler *s, void *hip_stream);
/* n_iter full Gibbs iterations = src/mmseq.cpp:851-918 (sample+scatter, gamma redraw,
 * trace capture), enqueued asynchronously. */
int mmg_sampler_run(mmg_sampler *s, int n_iter);
/* The two halves of one iteration, for read-shard mode where the caller all-reduces the
 * counts in between:  sample = :857-891 (+ :887 column sums), update = :905-917. */
int mmg_sampler_sample(mmg_sampler *s);
int mmg_sampler_update(mmg_sampler *s);
/* Device pointers for collectives: counts int32 [n_chains][n]; moments double
 * [2][n_chains][n] (sum log mu, sum log^2 mu over kept samples).  Element t of these buffers is DEVICE transcript t
 * (mmg_problem_tx_perm); ranks that passed the same tx_order agree on it, which is all a sum needs. */
int mmg_sampler_counts_devptr(mmg_sampler *s, void **ptr, uint64_t *count);
int mmg_sampler_moments_devptr(mmg_sampler *s, void **ptr, uint64_t *count);
int mmg_sampler_sync(mmg_sampler *s);
/* Wait until the first n_done iterations (n_done <= mmg_sampler_iteration) have completed on the device -- not for what is enqueued
 * behind them.  The reference prints sample s inside its loop (src/mmseq.cpp:911-917); a caller that streams the trace out enqueues
 * the next stretch of iterations, THEN waits for the previous one and fetches its rows (mmg_sampler_get_trace_rows_done), so the device
 * does not idle while the host formats.  An event follows the iteration that stored sample 16 j - 1 (j = 1, 2, ...) and the last sample;
 * other values of n_done wait for the next such iteration (or the whole stream when there is none). */
int mmg_sampler_wait_iterations(mmg_sampler *s, int n_done);
int mmg_sampler_iteration(const mmg_sampler *s, int *iter);
/* Trace of one chain, transcript-major exactly like mu_trace at src/mmseq.cpp:914:
 * out[t*trace_len + s]. */
int mmg_sampler_get_trace(mmg_sampler *s, int chain, double *out);
/* Same samples, sample-major (the row order of .trace_gibbs.gz, :912-916): out[s*n + t]. */
int mmg_sampler_get_trace_rows(mmg_sampler *s, int chain, int first_sample, int n_samples, double *out);
/* The same rows for samples the device has FINISHED (the caller synchronised after the iteration that produced the last of them):
 * copied on a stream of their own, without waiting for -- or delaying -- iterations enqueued behind those samples.  May be called
 * from another thread than the one driving the sampler: the trace writers of src/mmseq.cpp:911-917 print sample s inside the loop,
 * right after iteration s * gibbs_ss; this is how a caller does the same while the device runs on. */
int mmg_sampler_get_trace_rows_done(mmg_sampler *s, int chain, int first_sample, int n_samples, double *out);
int mmg_sampler_get_mu(mmg_sampler *s, int chain, double *mu);
/* Xcolsum of the last completed iteration (src/mmseq.cpp:896-899). */
int mmg_sampler_get_counts(mmg_sampler *s, int chain, int32_t *cnt);
int mmg_sampler_get_moments(mmg_sampler *s, int chain, double *sum_log, double *sum_log2, int64_t *n_samples);
int mmg_sampler_get_timing(mmg_sampler *s, mmg_timing *t);
int mmg_sampler_reset_timing(mmg_sampler *s);
void mmg_sampler_destroy(mmg_sampler *s);

/* ---- posterior summary of the resident trace ---------------------------------------------------
 * Everything src/mmseq.cpp:927-1363 derives from mu_trace, computed where the trace lives: simulated traces of isoforms
 * without hits (:971-978), trace sums over sets of identical transcripts and over genes (:927-1008), proportions of gene
 * expression (:1014-1031), and per series the percentiles (:1110-1192), the mean of the logged trace (:1195-1227), Sokal's
 * variance / autocorrelation time of the logged trace (:1307-1363, src/sokal.cc:33-87) and the proportion summaries
 * (:1235-1305).  The host downloads summary columns and, for the trace writers (:1033-1108), sample rows. */
typedef struct mmg_summary mmg_summary;
typedef struct mmg_summary_desc {
    int32_t chain;
    /* isoforms that received no hit: trace v is Gamma(alpha) * virtual_scale[v] keyed (seed, id virtual_id[v], sample), :974 */
    uint32_t n_virtual;
    const uint64_t *virtual_id;
    const double *virtual_scale;
    /* groups whose trace is the sum of their members' traces, members added in the given order; a member < n is the caller's
     * transcript, n + v is virtual transcript v */
    uint32_t n_identical;            /* sets of identical transcripts, :927-945 */
    const uint64_t *identical_ptr;   /* n_identical + 1 offsets into identical_member */
    const uint32_t *identical_member;
    uint32_t n_genes;                /* genes, :947-1008; a transcript's proportion is relative to the gene that lists it */
    const uint64_t *gene_ptr;
    const uint32_t *gene_member;
    uint32_t n_percentiles;          /* positions in the sorted trace, round(p / 100 * (trace_len - 1)), :1111-1114 */
    const int32_t *percentile_index;
} mmg_summary_desc;
enum { MMG_SERIES_TRANSCRIPT = 0, MMG_SERIES_VIRTUAL = 1, MMG_SERIES_IDENTICAL = 2, MMG_SERIES_GENE = 3 };
/* Any trace_len: a series is sorted and Fourier-transformed in LDS up to 8192 samples (the reference's trace length is 1024,
 * src/mmseq.cpp:190) and in a global workspace beyond (the same steps, slower).  Sokal's estimator needs a power of two in
 * [4, 2^21] (src/sokal.cc:36-39); for other lengths the percentiles and means exist and sokal_rc says why var / tau do not. */
int mmg_summary_create(mmg_sampler *s, const mmg_summary_desc *d, mmg_summary **out);
/* The same in steps, for a caller that prints the trace files while the chain runs (src/mmseq.cpp:911-917 prints sample s inside the
 * loop): _begin checks and uploads the description and draws the simulated traces (the chain may be running: its trace is not read);
 * _advance computes the derived rows of the samples [done, samples_done) -- the caller vouches that the chain has finished them (it
 * synchronised after iteration samples_done * gibbs_iter / trace_len - 1); mmg_summary_get_rows serves rows below samples_done;
 * _finish, once every sample is in, computes the summary columns.  The summary works on a stream of its own: it neither waits for nor
 * delays iterations enqueued behind the samples it reads.  One thread at a time drives _advance / _finish; mmg_summary_get_rows may be
 * called from several threads for rows already advanced. */
int mmg_summary_begin(mmg_sampler *s, const mmg_summary_desc *d, mmg_summary **out);
int mmg_summary_advance(mmg_summary *q, int samples_done);
int mmg_summary_finish(mmg_summary *q);
/* Per series of `kind` (n, n_virtual, n_identical or n_genes of them): mean of the logged trace, Sokal's var and tau of the
 * logged trace with its return code (0; 200 / 201 when trace_len is no power of two >= 4: var = tau = 0), and the
 * n_percentiles order statistics of the trace itself, [series][percentile].  Any output may be NULL. */
int mmg_summary_get(mmg_summary *q, int kind, double *log_mean, double *var, double *tau, int32_t *sokal_rc, double *percentiles);
/* Proportions of gene expression, kind MMG_SERIES_TRANSCRIPT or MMG_SERIES_VIRTUAL: mean proportion, mean and sd of the probit
 * of the proportion clamped to [1e-9, 1 - 1e-9] (+inf terms for the only transcript of its gene, as at :1243-1262), percentiles. */
int mmg_summary_get_proportions(mmg_summary *q, int kind, double *mean_prop, double *mean_probit, double *sd_probit, double *percentiles);
/* Sample rows of the derived traces as the trace writers print them: out[r * width + i], r over [first_sample, first_sample +
 * n_samples).  MMG_SERIES_IDENTICAL / MMG_SERIES_GENE: the summed traces; MMG_SERIES_TRANSCRIPT: the proportions (width n). */
int mmg_summary_get_rows(mmg_summary *q, int kind, int first_sample, int n_samples, double *out);
void mmg_summary_destroy(mmg_summary *q);

/* ---- several GPUs of one node, one process (RCCL over xGMI) ----------------------------------------
 * The reference parallelises with OpenMP threads inside one process (src/mmseq.cpp:834-838, :864); here the unit is a device.
 * A group owns one RCCL communicator per device (ncclCommInitAll); sampler i of every call below must live on device i of the
 * group.  RCCL is loaded on first use. */
typedef struct mmg_group mmg_group;
int mmg_group_create(const int *devices, int n, mmg_group **out);
int mmg_group_size(const mmg_group *g, int *n);
void mmg_group_destroy(mmg_group *g);
/* Read-shard mode: the samplers hold contiguous ranges of the stored rows of ONE chain (same seed, chain_base, iteration;
 * row_id_base = offset of the range).  n_iter iterations of: sample on every device (:857-891), ncclAllReduce(int32, sum) of
 * the count vectors in place (:896-899 across devices), the identical update everywhere (:905-917).  Asynchronous, like
 * mmg_sampler_run; the chain equals the chain of the unsharded problem bit for bit. */
int mmg_group_run_sharded(mmg_group *g, mmg_sampler *const *samplers, int n_iter);
/* Chains mode: every sampler advances its own chains (distinct chain_base) by n_iter iterations; nothing is exchanged. */
int mmg_group_run_chains(mmg_group *g, mmg_sampler *const *samplers, int n_iter);
/* Both run calls drive every device from its own host thread (thread i binds device i and enqueues its kernels and collectives
 * in order).  Host time the slowest of them spent per iteration in the last run call, in microseconds: what a device waits
 * for between iterations when its kernels are shorter than that. */
int mmg_group_enqueue_us(const mmg_group *g, double *us_per_device_iteration);
/* ncclAllReduce(fp64, sum) of the posterior moments over the devices (into scratch buffers: the samplers keep their own moments,
 * the call may be repeated), then summed over the chains of a device: sum_log[n], sum_log2[n] (caller's numbering) over n_samples
 * kept samples of all chains. */
int mmg_group_pool_moments(mmg_group *g, mmg_sampler *const *samplers, double *sum_log, double *sum_log2, int64_t *n_samples);
/* EM (src/mmseq.cpp:741-811) over read shards: shards[i] on device i holds a contiguous range of the stored rows.  Every phase of a
 * sweep runs on every device, then xe (max), the exact fixed-point accumulators with the log-likelihood limbs (uint64 sums) and,
 * once, the hits per transcript are all-reduced: integer sums, so every device holds the bits the unsharded EM produces, takes the
 * same repeat decisions and applies the same update.  ems[0..G) are returned; step and read ems[0] with mmg_em_step /
 * mmg_em_get_mu, destroy every member with mmg_em_destroy. */
int mmg_group_em_create(mmg_group *g, const mmg_problem *const *shards, const double *mu0, mmg_em **ems, double *loglik0);
/* Host helper: contiguous row ranges of (nearly) equal hit counts for `parts` shards: bounds[i] = first row of part i (even),
 * bounds[parts] = m. */
int mmg_shard_bounds(const uint64_t *row_ptr, uint64_t m, int parts, uint64_t *bounds);

/* int_of_ext[t] = device index of the caller's transcript t (identity without tx_order). */
int mmg_problem_tx_perm(const mmg_problem *p, uint32_t *int_of_ext);

/* ---- host-side keyed draws ------------------------------------------------------------ */
/* out[i] = Gamma(shape, scale) from the keyed stream (seed, tag SIMU, id, i): the simulated traces
 * of isoforms without hits, src/mmseq.cpp:971-978 (which uses rg[0] there).  Pure host code. */
int mmg_host_gamma_trace(uint64_t seed, uint64_t id, double shape, double scale, int n, double *out);

/* ---- self-test hooks (used by tests only) ------------------------------------------- */
/* Process-wide overrides of choices the library normally makes itself, so that tests can reach every kernel on small
 * inputs.  They never change a result.  value < 0 restores the default. */
enum {
    MMG_OPT_SAMPLE_KERNEL = 0,     /* 0: k_sample (CSR tiles), 2: k_sample_sell even if few tiles qualify            */
    MMG_OPT_FORCE_IDX64 = 1,       /* 1: 64-bit device row offsets regardless of nnz                                 */
    MMG_OPT_SELL_WAVES_PER_CU = 2, /* cap on resident single-wave workgroups per CU (long tile ranges on small inputs) */
    MMG_OPT_EM_KERNEL = 3,         /* 0: row-per-thread EM kernel, 2: sliced-ELL EM kernel                           */
    MMG_OPT_EM_GRID = 4,           /* cap on the EM kernel's grid                                                    */
    MMG_OPT_FUSE_CHAINS = 5,       /* chains advanced per K1 launch: 1 (never fuse), 2 (the default), 4              */
    MMG_OPT_CNT_REPLICAS = 6,      /* 1: one global count vector per chain, 8: replicated (default: by ranges per band) */
    MMG_OPT_GROUP_FAIL = 7,        /* v >= 0: member v % size of a group fails in its second iteration of the next run call (error path) */
    MMG_OPT_DERIVE_ORDER = 8,      /* 0: never derive a transcript order from the hit graph, 1: try it on every canonical problem without tx_order */
    MMG_OPT_WIRE_CHECK = 9,        /* 0: no verification of a group's first exchanges, 1: verify in groups of one device too, 2: 1 + damage a word behind the exchange (the failure path) */
    MMG_OPT_BIGK_PER_WAVE = 10,    /* list entries per workgroup of k_sample_bigk (the rows on the conditional-binomial chain)                      */
    MMG_OPT_BIGK_SIDE_STREAM = 11, /* 0: k_sample_bigk on the sampler's stream, in front of the tile kernels instead of beside them                    */
    MMG_OPT_COUNT_ = 12
};
int mmg_selftest_option(int option, int value);
/* The sharded EM of mmg_group_em_create with every shard on ONE device and the exchange done by plain kernels: `sweeps` sweeps from
 * mu0, mu (caller's numbering), the log-likelihood and the number of repeated passes -- the same bits as mmg_problem_em on the
 * unsharded problem. */
int mmg_selftest_em_shards(const mmg_problem *const *shards, int n_shards, const double *mu0, int sweeps, double *mu, double *loglik,
                           int *repeated_passes);
/* The sharded chain of mmg_group_run_sharded with every shard on ONE device and the count exchange done by a plain kernel: n_iter
 * iterations of sample on every shard (src/mmseq.cpp:857-891), the count vectors summed over the shards into every sampler
 * (:896-899), the identical update everywhere (:905-917) -- the same bits as mmg_sampler_run on the unsharded problem.  With
 * cfg.timing set, mmg_sampler_get_timing of sampler i reports what shard i's sample kernels took. */
int mmg_selftest_gibbs_shards(mmg_sampler *const *samplers, int n_shards, int n_iter);
/* What the HIP runtime reports about the k == NULL sliced-ELL sample kernel on `device`: registers, LDS and scratch bytes per
 * thread, and resident 64-thread workgroups per CU (hipOccupancyMaxActiveBlocksPerMultiprocessor). */
int mmg_selftest_kernel_info(int device, int *vgprs, int *lds_bytes, int *scratch_bytes, int *resident_per_cu);
/* Evaluates the library's own log / exp / sqrt / 1/x on x[0..n) (device >= 0: in a kernel
 * on that device; device == -1: the host instantiation of the same inline code). */
int mmg_selftest_math(int device, int64_t n, const double *x, double *out_log, double *out_exp,
                      double *out_sqrt, double *out_rcp);
/* out[0..4) = Philox4x32-10(ctr[4], key[2]); out[4..6) = Philox2x32-10(ctr[0..2), key[0]). */
int mmg_selftest_philox(int device, const uint32_t *ctr, const uint32_t *key, uint32_t *out);
/* out[i] = shape-`shape` gamma draw of stream (seed, chain 0, iter 0, id i) times scale. */
int mmg_selftest_gamma(int device, uint64_t seed, double shape, double scale, int64_t n, double *out);
/* out[i] = Binomial(nn, p) draw of row stream (seed, id i). */
int mmg_selftest_binomial(int device, uint64_t seed, uint32_t nn, double p, int64_t n, uint32_t *out);
/* n_cases BTRS attempts over n log-uniform in [n_lo, n_hi] (21 <= n_lo <= n_hi < 2^32) and p log-uniform in [10 / n, 1/2]: counts[0] attempts that
 * reach the exact acceptance test, [1] of them decided by the fp32 estimate k_sample_bigk tries first (mmg_math.h: btrs_pretest), [2] decided
 * AND different from the fp64 test -- must be 0 --, [3] accepted by the fp64 test, [4] the largest |estimate - fp64 difference| seen, in
 * millionths of the estimate's own error bound (counts: 5 words). */
int mmg_selftest_btrs_pretest(int device, uint64_t seed, int64_t n_cases, double n_lo, double n_hi, uint64_t *counts);

/* Diagnostics: the same for the inversion (n p < 10): n_cases searches, n log-uniform in [n_lo, n_hi] (1 <= n_lo), n p log-uniform in
 * [1e-6, 10), through the fp32 search k_sample_bigk tries first (mmg_math.h: binv_pretest) and the fp64 search of the sequential code.
 * counts[0] cases, [1] decided in fp32, [2] decided AND different from the fp64 search -- must be 0 --, [3] cases whose fp64 search ran off
 * the end (the sequential code draws again), [4] the sum of the outcomes (counts: 5 words).  slack scales the error bound the fp32 search
 * allows itself: 1 is what the sampler runs; smaller values show how much room the bound has. */
int mmg_selftest_binv_pretest(int device, uint64_t seed, int64_t n_cases, double n_lo, double n_hi, double slack, uint64_t *counts);

#ifdef __cplusplus
}
#endif
#endif /* MMGIBBS_H */
