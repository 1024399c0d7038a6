// K1 on a sliced-ELL ("SELL-64") 8-bit stream: the default sampler kernel.
//
// The CSR-tile kernel (k_sample, gibbs_kernels.h) stages every tile through LDS: the column ids are written there, the row extents
// and the offsets are read back before the mu gathers can start, and the waves of a workgroup meet at barriers per tile -- LDS time
// bounds it, and about half of that is staging, not gathers.  Here a tile is 64 consecutive rows = one wave, stored column-major:
// group g of the tile is 64 x (4 u8 window indices) = 256 contiguous bytes, lane r owns bytes [4r, 4r+4): one byte per hit (the
// window holds 255 transcripts, index 255 is its 0.0 slot).  A lane loads its own row's groups straight into registers with
// perfectly coalesced 4-byte loads; nothing but the mu gathers and the count atomic touches LDS, and a workgroup is a single wave,
// so there is no barrier on the tile path at all.  Rows are padded to the longest row of their tile (the canonical order makes
// tiles homogeneous in length) with the offset of the window's 0.0 slot: every lane walks the same number of groups, a pad adds an
// exact 0.0 and can never be selected, so the draw equals the oracle's plain sequential walk bit for bit -- same keyed stream, same
// additions in the same order.  Rows with hits outside the window keep those in a far list behind the block (far tiles); problems
// with multiplicities run the tiles that hold them through the HAS_K instantiation in a second launch (mmg_types.h), and the rows
// on the conditional-binomial chain from a list of their own in a third (bigk_kernels.h).
#pragma once

namespace mmg {

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <bool B> struct BoolTag { static constexpr bool value = B; };
template <int K> struct IntTag { static constexpr int value = K; };

// A tile's block seen through a raw buffer descriptor of exactly its size (ng groups of 256 bytes).  The sampler and
// EM kernels request NGC groups per tile whatever its ng (loads retire in order and are waited for by count, so the number issued
// must not depend on the tile): through the descriptor a request past the block's end returns 0 WITHOUT a memory access -- the
// hardware's range check is the clamp, at no instruction per group (clamped scalar addresses cost 10 % of the kernel's time, plain
// over-reads a quarter more HBM traffic).  Descriptor words, offsets and the block are wave-uniform: everything stays in SGPRs.
struct SellBlock {
    __amdgpu_buffer_rsrc_t rs;
    // extra: bytes behind the window part that the descriptor covers as well (the far list of a far tile)
    __device__ SellBlock(const uint8_t *blk, uint32_t meta, uint32_t extra = 0)
        : rs(__builtin_amdgcn_make_buffer_rsrc((void *)blk, 0, (int)((meta & 0xff00u) + extra), 0x00020000)) {} // 0x00020000: gfx9 raw dword buffer
    // group i of the lane's row; the constant part of the offset folds into the instruction, aux 2 = nontemporal (streamed once)
    template <int I> __device__ uint32_t group(uint32_t lane) const
    {
        return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(lane * 4u + (uint32_t)I * 256u), 0, 2);
    }
};

// hits in a group word: pads (255) fill the tail of a row, a row's length is the number of bytes below 255 in its groups -- the stream
// carries no length (only the rare paths of a draw and the multiplicity paths ask for it)
__device__ __forceinline__ uint32_t sell_group_hits(uint32_t w)
{
    const uint32_t m = ~w;
    return (uint32_t)((m & 0xffu) != 0u) + (uint32_t)((m & 0xff00u) != 0u) + (uint32_t)((m & 0xff0000u) != 0u) + (uint32_t)((m & 0xff000000u) != 0u);
}

// Far-tile facts the host needs before it can lay the stream out: the window base a tile's rows were sorted for (the band field of
// the first non-empty row's key, mmg_types.h), the most window hits and the most other hits a row of the tile has -- or "not a far
// tile" (nf = ~0) when some row is not stored window-hits-first (rows kept in the caller's order).
template <typename IdxT>
__global__ __launch_bounds__(64) void k_tile_far(const IdxT *__restrict__ row_ptr, const uint32_t *__restrict__ col_idx,
                                                 const uint64_t *__restrict__ key, const uint64_t *__restrict__ tile_row,
                                                 const uint32_t *__restrict__ cand, uint64_t n_cand,
                                                 uint32_t *__restrict__ out /* [3][n_cand]: wbase, most window hits, nf */)
{
    if (blockIdx.x >= n_cand) return;
    const uint64_t tile = cand[blockIdx.x]; // the host asks only about tiles that missed the register path
    const uint64_t r0 = tile_row[tile], r1 = tile_row[tile + 1];
    const uint32_t lane = threadIdx.x;
    uint64_t first = ~0ull; // first non-empty row of the tile
    for (uint64_t r = r0 + lane; r < r1; r += 64)
        if (key[r] != 0) { first = r; break; }
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t o = ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(first >> 32), off) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)first, off);
        first = o < first ? o : first;
    }
    uint32_t wbase = 0, nn = 0, nf = 0;
    if (first != ~0ull) {
        wbase = (uint32_t)(((key[first] >> 18) & LAYOUT_KEY_BAND_MASK) << LAYOUT_BAND_SHIFT);
        for (uint64_t r = r0 + lane; r < r1; r += 64) {
            uint32_t n = 0, f = 0;
            bool ordered = true;
            for (uint64_t j = row_ptr[r], e = row_ptr[r + 1]; j < e; ++j) {
                const bool in = (col_idx[j] - wbase) < SELL_WIN;
                ordered = ordered && !(in && f > 0); // a window hit behind a far one
                n += in;
                f += !in;
            }
            nn = max(nn, n);
            nf = max(nf, ordered ? f : 0xffffffffu);
        }
        for (int off = 32; off > 0; off >>= 1) {
            nn = max(nn, (uint32_t)__shfl_xor((int)nn, off));
            nf = max(nf, (uint32_t)__shfl_xor((int)nf, off));
        }
    }
    if (lane == 0) { out[blockIdx.x] = wbase; out[n_cand + blockIdx.x] = nn; out[2 * n_cand + blockIdx.x] = nf; }
}

// Block of a fast tile: ng groups of 64 lanes x 4 u8 window indices (col - wbase), 255 = pad (a row's length is what is not a pad).
// Block of a far tile: the same for the window hits at the head of every row, then 64 far-count bytes and nf groups of 64 lanes x
// u32: the transcript ids of the lane's other hits in stored order (0 beyond the row's own count).
template <typename IdxT>
__global__ __launch_bounds__(64) void k_encode_sell(const IdxT *__restrict__ row_ptr, const uint32_t *__restrict__ col_idx,
                                                    const SellTile *__restrict__ tiles, uint64_t n_tiles, uint8_t *stream)
{
    const uint64_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const SellTile d = tiles[tile];
    if (!(d.flags() & (SELL_FAST | SELL_FAR))) return;
    uint8_t *blk = stream + d.off16 * 16;
    const uint32_t lane = threadIdx.x;
    uint64_t b = 0;
    uint32_t L = 0, Ln = 0;
    if (lane < d.nrows()) {
        b = row_ptr[d.r0 + lane];
        L = (uint32_t)((uint64_t)row_ptr[d.r0 + lane + 1] - b);
        Ln = L;
        if (d.flags() & SELL_FAR)
            for (Ln = 0; Ln < L && (col_idx[b + Ln] - d.wbase) < SELL_WIN; ++Ln) {}
    }
    uint32_t *grp = (uint32_t *)blk + lane; // (these tiles hold rows of at most 255 hits)
    for (uint32_t g = 0; g < d.ng(); ++g) {
        uint32_t w = 0;
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t idx = 4 * g + j;
            const uint32_t o = idx < Ln ? col_idx[b + idx] - d.wbase : SELL_WIN;
            w |= o << (8 * j);
        }
        grp[(size_t)g * 64] = w;
    }
    if (d.flags() & SELL_FAR) {
        uint8_t *fb = blk + (size_t)d.ng() * 256;
        fb[lane] = (uint8_t)(L - Ln);
        uint32_t *far = (uint32_t *)(fb + 64) + lane;
        for (uint32_t f = 0; f < d.nf(); ++f) far[(size_t)f * 64] = f < L - Ln ? col_idx[b + Ln + f] : 0u;
    }
}

// One row of a far tile for the multiplicity kernel: hit j is window byte j of the lane for j < Ln, entry j - Ln of its far list
// beyond.  Plain sequential loops (rows with multiplicities AND far hits are rare); the additions happen in stored order.
struct RowViewFarTile {
    const uint32_t *grp; // group words of this lane: grp[g * 64]
    const uint32_t *far; // far list of this lane: far[f * 64]
    uint32_t Ln, L;      // window hits, all hits
    uint32_t wbase;
    const double *s_mu;
    const double *gmu;
    __device__ __forceinline__ uint32_t col(uint32_t j) const
    {
        return j < Ln ? wbase + ((grp[(size_t)(j >> 2) * 64] >> (8u * (j & 3u))) & 0xffu) : far[(size_t)(j - Ln) * 64];
    }
    __device__ __forceinline__ double w(uint32_t j) const
    {
        const uint32_t c = col(j);
        return j < Ln ? s_mu[c - wbase] : gmu[c];
    }
    __device__ __forceinline__ double total() const
    {
        double t = 0.0;
        for (uint32_t j = 0; j < L; ++j) t += w(j);
        return t;
    }
    __device__ __forceinline__ uint32_t pick(double target) const
    {
        double acc = 0.0;
        for (uint32_t j = 0; j < L; ++j) {
            acc += w(j);
            if (target < acc) return j;
        }
        return L - 1;
    }
};

// FARPF: the instantiation that walks the list of far (and CSR-walked) tiles: the far count and the first far transcript of every lane
// travel with the block's prefetch, and the first far weight is gathered while the window part is walked.  Without it a far tile is
// three dependent memory round trips in the middle of its walk (count -> transcript id -> weight): 10 x a register-path tile.
// FIXW: register-path tiles of at most 8 groups take straight-line code per group count (walk_fixed) -- the instantiation for problems
// of fewer than 5 groups per tile on average.  Same-box A/B of K1, generic -> straight-line: 5 M reads x 8 hits (BASELINE configs[1])
// 0.0258 -> 0.0246 ms; 20 M x 12 hits 0.0792 -> 0.0762; 50 M x 16 hits 0.1859 -> 0.1816; at 50 M x 20 hits (5.4 groups per tile: the
// headline) the stream bounds the kernel and the generic walk is 0.4-0.7 % ahead (0.2254 vs 0.2264): the host picks per problem
// (mmg_problem::k1_fixed_walk)
template <typename IdxT, bool HAS_K, int NGC, int REP = 1, bool FARPF = false, bool FIXW = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(HAS_K ? 5 : (FARPF ? 6 : 7), HAS_K ? 5 : (FARPF ? 6 : 7)))) void k_sample_sell(const IdxT *__restrict__ row_ptr, const uint32_t *__restrict__ col_idx,
                                                    const uint32_t *__restrict__ kmult, const SellTile *__restrict__ tiles, const uint64_t *__restrict__ chunk_tile,
                                                    const double *__restrict__ gmu /* [grid.y][n] */, const uint8_t *__restrict__ stream,
                                                    int32_t *gcnt /* [grid.y][n] */, SampleArgs a)
{
    constexpr int WIN = (int)SELL_WIN;
    __shared__ __attribute__((aligned(16))) double s_mu[WIN + 1]; // [WIN] stays 0.0: what pad slots read
    // REP replicas of the counts (lane l adds to replica l % REP) were tried against same-address serialisation of the LDS
    // atomics: with one atomic per ROW they only cost (0.433 ms with 1, 0.445 / 0.455 / 0.473 with 2 / 4 / 8 at cfg 3)
    // counts at the stride of the weights (slot i at byte 8 i, the upper word unused): the offset that gathered a weight addresses
    // its count, no shift
    __shared__ int32_t s_cnt[REP * 2 * (WIN + 1)];
    // Counts of picks OUTSIDE the window (the far hit of a far row, a CSR-walked row's hits elsewhere): a small keyed table for the
    // workgroup's whole range, flushed with one global atomic per key at the end.  Without it every such pick is a global atomic of its
    // own, and the popular far targets -- a transcript whose repeat-bearing UTR collects reads of hundreds of genes -- take thousands of
    // them per sweep at ONE address each, which the memory side serialises: 1 % of the reads on 50 hub transcripts cost K1 +32 %
    // (tools/families_probe.py; profiles/r06_families_probe.txt).  A key keeps the slot it finds empty (two candidates); a pick whose slots hold other keys
    // goes to the global vector directly, as before.  (One wave per workgroup: no race beyond the atomics themselves.)
    constexpr uint32_t FAR_SLOTS = 128, FAR_EMPTY = 0xffffffffu;
    __shared__ uint32_t s_fkey[FAR_SLOTS];
    __shared__ int32_t s_fcnt[FAR_SLOTS];
    const uint32_t lane = threadIdx.x;
    // grid.y = chain: the tile lists that only some rows are on (multiplicities, far rows) are walked for every chain of a sampler
    // in one launch -- their launches are bound by a few long rows, eight of them side by side fill the GPU eight times better
    gmu += (size_t)blockIdx.y * a.n;
    gcnt += (size_t)blockIdx.y * a.n + (size_t)(blockIdx.x & a.cnt_rep_mask) * a.cnt_rep_stride; // (mmg_types.h: CNT_REPLICAS)
    a.chain += blockIdx.y;

    // the range's header (mmgibbs.hip: upload_ranges): first tile, end tile, the descriptors of its first two tiles -- one scalar load
    const uint64_t *__restrict__ hdr = chunk_tile + (size_t)blockIdx.x * 8;
    const uint64_t t_begin = hdr[0], t_end = hdr[1];
    if (t_begin >= t_end) return;
    SellTile d_first, d_second;
    d_first.off16 = hdr[2]; d_first.r0 = hdr[3]; d_first.wbase = (uint32_t)hdr[4]; d_first.meta = (uint32_t)(hdr[4] >> 32);
    d_second.off16 = hdr[5]; d_second.r0 = hdr[6]; d_second.wbase = (uint32_t)hdr[7]; d_second.meta = (uint32_t)(hdr[7] >> 32);
    // a range holds fewer than 2^31 tiles: 32-bit scalar loop arithmetic
    const uint32_t nt = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(t_end - t_begin));
    const SellTile *__restrict__ T = tiles + t_begin;

    for (int i = lane; i < REP * 2 * (WIN + 1); i += 64) s_cnt[i] = 0;
    for (uint32_t i = lane; i < FAR_SLOTS; i += 64) { s_fkey[i] = FAR_EMPTY; s_fcnt[i] = 0; }
    auto far_add = [&](uint32_t col, int32_t x) {
        // two candidate slots per key (the cold far targets of paralogue rows come first and sit in the table too: with one slot per key a
        // sixth of the hot ones found theirs taken)
        const uint32_t slot = (col * 0x9E3779B1u) >> 25; // 7 bits
        uint32_t old = atomicCAS(&s_fkey[slot], FAR_EMPTY, col);
        if (old == FAR_EMPTY || old == col) { atomicAdd(&s_fcnt[slot], x); return; }
        const uint32_t slot2 = ((col * 0x85EBCA6Bu) >> 25) ^ 64u;
        old = atomicCAS(&s_fkey[slot2], FAR_EMPTY, col);
        if (old == FAR_EMPTY || old == col) atomicAdd(&s_fcnt[slot2], x);
        else global_count_add(gcnt, col, x);
    };
    const uint32_t rep_off = (lane % REP) * (uint32_t)((WIN + 1) * 8);
    if (lane == 0) s_mu[WIN] = 0.0;

    auto flush_window = [&](uint32_t base) {
        for (int i = lane; i < WIN; i += 64) {
            int32_t v = 0;
#pragma unroll
            for (int r = 0; r < REP; ++r) { v += s_cnt[2 * (r * (WIN + 1) + i)]; s_cnt[2 * (r * (WIN + 1) + i)] = 0; }
            if (v) global_count_add(gcnt, base + (uint32_t)i, v);
        }
    };
    auto load_window = [&](uint32_t base) {
        for (int i = lane; i < WIN; i += 64) {
            const uint32_t c = base + (uint32_t)i;
            s_mu[i] = c < a.n ? gmu[c] : 0.0;
        }
    };
    auto wo = [&](uint32_t off) { return *(const double *)((const char *)s_mu + off); }; // off = window index * 8
    // byte k of a group word as an LDS byte offset
    // one instruction per byte (SDWA byte select + shift); written out because the compiler turns the first byte into shift + mask
    const uint32_t three = 3u;
    auto sdwa_off = [&](uint32_t v, auto sel_tag) -> uint32_t {
        constexpr int K = decltype(sel_tag)::value;
        uint32_t o;
        if (K == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(o) : "v"(three), "v"(v));
        else if (K == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(o) : "v"(three), "v"(v));
        else if (K == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(o) : "v"(three), "v"(v));
        else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(o) : "v"(three), "v"(v));
        return o;
    };
#define SELL_OFF0(v) sdwa_off(v, IntTag<0>())
#define SELL_OFF1(v) sdwa_off(v, IntTag<1>())
#define SELL_OFF2(v) sdwa_off(v, IntTag<2>())
#define SELL_OFF3(v) sdwa_off(v, IntTag<3>())

    // The cached groups and the prefix sums are NAMED registers (macro-expanded), not arrays: a select chain over an
    // array that a loop once indexed is turned back into a dynamic index by the optimiser, and the array lands in scratch.
#define SELL_GROUPS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
    static_assert(NGC == 8, "the group list above is written out for 8 cached groups");
    struct Buf {
        uint32_t g0, g1, g2, g3, g4, g5, g6, g7;
        uint32_t kk; // multiplicity of the lane's row (HAS_K)
        uint32_t lf, f0; // FARPF: entries in the lane's far list, the first of them
    };
    // request a tile's block: the lane's NGC groups, UNCONDITIONALLY (for tiles without a block -- empty, slow, past the end of
    // the range -- every request fails the range check).  Loads retire in order and are waited for by count, so the
    // number issued per tile must not depend on the path -- otherwise the compiler has to assume the fewest, and every walk
    // waits for the prefetch issued just before it.  Groups beyond the tile's ng fail the descriptor's range check: no memory access.
    auto issue = [&](const SellTile &d, Buf &bf) {
        const bool fast = d.flags() & (SELL_FAST | SELL_FAR); // uniform: the tile has a block (a far tile's window part is a fast tile's)
        const bool farb = FARPF && (d.flags() & SELL_FAR);
        const SellBlock blk(stream + (fast ? d.off16 * 16 : 0), d.meta, farb ? 64u + d.nf() * 256u : 0u);
        if (FARPF) { // two more loads, whatever the tile (a tile without a far list reads its own head again)
            const uint32_t fo = farb ? d.ng() * 256u : 0u; // the far part: 64 count bytes, then nf groups of 64 transcript ids
            bf.lf = __builtin_amdgcn_raw_buffer_load_b8(blk.rs, (int)lane, (int)fo, 0);
            bf.f0 = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(blk.rs, (int)(lane * 4u + 64u), (int)fo, 2);
        }
        if (HAS_K) bf.kk = kmult[(fast ? d.r0 : 0) + min(lane, (fast ? d.nrows() : 1u) - 1u)];
#define SELL_ISSUE(i) bf.g##i = blk.template group<i>(lane);
        SELL_GROUPS(SELL_ISSUE)
#undef SELL_ISSUE
    };

    // Pair RNG: ONE Philox2x32-10 block per lane serves the fast rows of BOTH tiles of the unrolled pair (mmg_math.h: a block
    // belongs to rows 2q and 2q+1): lanes 0..31 hold the blocks of tile A, lanes 32..63 those of tile B.  A tile spans at most
    // 32 blocks (the host cuts tiles so that parity + nrows <= 64).
    uint32_t xrowA = 0, xrowB = 0; // this lane's random word for its row of tile A / tile B
    auto pair_rng = [&](const SellTile &A, const SellTile &B) {
        const uint64_t qa = (a.row_id_base + A.r0) >> 1, qb = (a.row_id_base + B.r0) >> 1; // uniform
        const uint32_t l5 = lane & 31u;
        uint32_t x0, x1;
        // one key for the whole wave unless a multiple of 2^33 row ids lies between the first row of A and the last of B
        const bool one_key = (((a.row_id_base + A.r0) ^ (a.row_id_base + B.r0 + 63u)) >> 33) == 0 && A.r0 <= B.r0;
        if (one_key && (((a.row_id_base + A.r0) | (a.row_id_base + B.r0)) & 1u) == 0) {
            // Both tiles start on an even row id (all but the first tile of a run): rows l and l + 1 (l even) of a tile share a block.
            // The EVEN lane of a lane pair computes the pair's block for tile A, the ODD lane the one for tile B, and each takes the
            // word it lacks from its neighbour with a quad-permute DPP move -- no LDS instruction (ds_bpermute: 7 LDS cycles each,
            // four of them per pair of tiles; tools/issue_bench.hip).
            const uint32_t key = stream2_key(a.seed, a.chain, TAG_ROW, (uint32_t)(qa >> 32));
            x0 = ((lane & 1u) ? (uint32_t)qb : (uint32_t)qa) + (lane >> 1);
            x1 = a.iter;
            philox2x32_10(x0, x1, key);
            const uint32_t n0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)x0, 0xB1, 0xF, 0xF, true); // quad_perm [1, 0, 3, 2]: the neighbour's words
            const uint32_t n1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)x1, 0xB1, 0xF, 0xF, true);
            xrowA = (lane & 1u) ? n1 : x0;   // row l of A: word 0 of the even lane's block, word 1 for the odd lane
            xrowB = (lane & 1u) ? x1 : n0;   // row l of B: the odd lane's block
            return;
        }
        if (one_key) {
            // the usual case: one key for the whole wave, kept in scalar registers
            const uint32_t key = stream2_key(a.seed, a.chain, TAG_ROW, (uint32_t)(qa >> 32));
            x0 = (lane < 32u ? (uint32_t)qa : (uint32_t)qb) + l5;
            x1 = a.iter;
            philox2x32_10(x0, x1, key);
        } else { // a pair that straddles a multiple of 2^33 row ids: per-lane keys, same values
            const uint64_t q = (lane < 32u ? qa : qb) + l5;
            x0 = (uint32_t)q;
            x1 = a.iter;
            philox2x32_10(x0, x1, stream2_key(a.seed, a.chain, TAG_ROW, (uint32_t)(q >> 32)));
        }
        // row (parity + lane) of a tile: block (parity + lane) >> 1 of the tile's half-wave, word (parity + lane) & 1.  All 64
        // lanes are active here (ds_bpermute returns 0 for a source lane that is masked off).
        const uint32_t pa = ((uint32_t)(a.row_id_base + A.r0) & 1u) + lane, pb = ((uint32_t)(a.row_id_base + B.r0) & 1u) + lane;
        const int sa = (int)((pa >> 1) << 2), sb = (int)((32u + (pb >> 1)) << 2);
        const uint32_t a0 = (uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)x0), a1 = (uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)x1);
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_ds_bpermute(sb, (int)x0), b1 = (uint32_t)__builtin_amdgcn_ds_bpermute(sb, (int)x1);
        xrowA = (pa & 1u) ? a1 : a0;
        xrowB = (pb & 1u) ? b1 : b0;
    };

    // ---- the walk of a register-path tile of at most 8 groups, written out per group count (the k = 1 kernel without far lists) -----
    // One body for every ng (walk, below) costs a scalar test per group, a copy of the running sum behind every conditional group and a
    // branch ladder into the pick's sweep.  Here a tile's ng selects straight-line code: the boundaries are the sums themselves (no
    // copies), the sweep has no entry branches, PIPE: the gathers of group i + 1 are in flight while group i is added.  Same additions,
    // comparisons and draw in the same order: bit-identical (tests/test_gpu_parity.py, test_gpu_fullsize.py, tools/fuzz_parity.py).
    auto walk_fixed = [&](const SellTile &d, const Buf &bf, uint32_t which, auto ng_tag) {
        constexpr int NG = decltype(ng_tag)::value;
        constexpr bool PIPE = true; // (the gathers of group i + 1 in flight while group i is added: still 72 VGPRs)
        static_assert(!HAS_K && NG >= 1 && NG <= 8, "k = 1 kernel, cached groups only");
        const uint32_t gw[8] = {bf.g0, bf.g1, bf.g2, bf.g3, bf.g4, bf.g5, bf.g6, bf.g7};
        struct G4 { double w[4]; };
        auto gather = [&](uint32_t v) {
            G4 r;
            const uint32_t o0 = SELL_OFF0(v), o1 = SELL_OFF1(v), o2 = SELL_OFF2(v), o3 = SELL_OFF3(v);
            r.w[0] = wo(o0); r.w[1] = wo(o1); r.w[2] = wo(o2); r.w[3] = wo(o3);
            return r;
        };
        double P[NG];
        {
            G4 cur = gather(gw[0]), nxt = cur;
            if (PIPE && NG > 1) nxt = gather(gw[1]);
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                __builtin_amdgcn_sched_barrier(0);
                double t = i == 0 ? cur.w[0] : P[i - 1] + cur.w[0]; // 0.0 + w == w exactly
                t += cur.w[1]; t += cur.w[2]; t += cur.w[3];
                P[i] = t;
                __builtin_amdgcn_sched_barrier(0);
                if (PIPE) { cur = nxt; if (i + 2 < NG) nxt = gather(gw[i + 2]); }
                else if (i + 1 < NG) cur = gather(gw[i + 1]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t x = which ? xrowB : xrowA;
        const double ts = P[NG - 1] * 0x1p-32, hs = ts * 0.5;
        const double target = draw_target(x, ts, hs); // mmg_math.h
        uint32_t v;
        double acc;
        {
            uint64_t sv, tm;
#define SF_STEP(i, prev) "v_cmpx_lt_f64_e64 %[tm], %[t], %[p" #i "]\n\t" "v_mov_b32 %[v], %[g" #i "]\n\t" "v_mov_b64 %[acc], " prev "\n\t"
#define SF_HEAD "s_mov_b64 %[sv], exec\n\t" "v_mov_b32 %[v], 0\n\t" "v_mov_b64 %[acc], 0\n\t"
#define SF_TAIL "s_mov_b64 exec, %[sv]"
#define SF_OUT [v] "=&v"(v), [acc] "=&v"(acc), [sv] "=&s"(sv), [tm] "=&s"(tm)
#define SF_IN(n) [t] "v"(target), [p0] "v"(P[0]), [p1] "v"(P[n > 1 ? 1 : 0]), [p2] "v"(P[n > 2 ? 2 : 0]), [p3] "v"(P[n > 3 ? 3 : 0]),       \
                 [p4] "v"(P[n > 4 ? 4 : 0]), [p5] "v"(P[n > 5 ? 5 : 0]), [p6] "v"(P[n > 6 ? 6 : 0]), [p7] "v"(P[n > 7 ? 7 : 0]),            \
                 [g0] "v"(gw[0]), [g1] "v"(gw[n > 1 ? 1 : 0]), [g2] "v"(gw[n > 2 ? 2 : 0]), [g3] "v"(gw[n > 3 ? 3 : 0]), [g4] "v"(gw[n > 4 ? 4 : 0]), \
                 [g5] "v"(gw[n > 5 ? 5 : 0]), [g6] "v"(gw[n > 6 ? 6 : 0]), [g7] "v"(gw[n > 7 ? 7 : 0])
#define SF_S0 SF_STEP(0, "0")
#define SF_S1 SF_STEP(1, "%[p0]") SF_S0
#define SF_S2 SF_STEP(2, "%[p1]") SF_S1
#define SF_S3 SF_STEP(3, "%[p2]") SF_S2
#define SF_S4 SF_STEP(4, "%[p3]") SF_S3
#define SF_S5 SF_STEP(5, "%[p4]") SF_S4
#define SF_S6 SF_STEP(6, "%[p5]") SF_S5
#define SF_S7 SF_STEP(7, "%[p6]") SF_S6
            if constexpr (NG == 1) asm volatile(SF_HEAD SF_S0 SF_TAIL : SF_OUT : SF_IN(1));
            else if constexpr (NG == 2) asm volatile(SF_HEAD SF_S1 SF_TAIL : SF_OUT : SF_IN(2));
            else if constexpr (NG == 3) asm volatile(SF_HEAD SF_S2 SF_TAIL : SF_OUT : SF_IN(3));
            else if constexpr (NG == 4) asm volatile(SF_HEAD SF_S3 SF_TAIL : SF_OUT : SF_IN(4));
            else if constexpr (NG == 5) asm volatile(SF_HEAD SF_S4 SF_TAIL : SF_OUT : SF_IN(5));
            else if constexpr (NG == 6) asm volatile(SF_HEAD SF_S5 SF_TAIL : SF_OUT : SF_IN(6));
            else if constexpr (NG == 7) asm volatile(SF_HEAD SF_S6 SF_TAIL : SF_OUT : SF_IN(7));
            else asm volatile(SF_HEAD SF_S7 SF_TAIL : SF_OUT : SF_IN(8));
#undef SF_S0
#undef SF_S1
#undef SF_S2
#undef SF_S3
#undef SF_S4
#undef SF_S5
#undef SF_S6
#undef SF_S7
#undef SF_IN
#undef SF_OUT
#undef SF_TAIL
#undef SF_HEAD
#undef SF_STEP
        }
        const uint32_t o0 = SELL_OFF0(v), o1 = SELL_OFF1(v), o2 = SELL_OFF2(v), o3 = SELL_OFF3(v);
        double w0 = wo(o0), w1 = wo(o1), w2 = wo(o2);
        asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2)); // all three requested before the first is waited for
        const double p0 = acc + w0, p1 = p0 + w1, p2 = p1 + w2;
        uint32_t sel = o3;
        {
            uint64_t sv, tm;
            asm volatile("s_mov_b64 %[sv], exec\n\t"
                         "v_cmpx_lt_f64_e64 %[tm], %[t], %[p2]\n\t" "v_mov_b32 %[sel], %[o2]\n\t"
                         "v_cmpx_lt_f64_e64 %[tm], %[t], %[p1]\n\t" "v_mov_b32 %[sel], %[o1]\n\t"
                         "v_cmpx_lt_f64_e64 %[tm], %[t], %[p0]\n\t" "v_mov_b32 %[sel], %[o0]\n\t"
                         "s_mov_b64 exec, %[sv]"
                         : [sel] "+&v"(sel), [sv] "=&s"(sv), [tm] "=&s"(tm)
                         : [t] "v"(target), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [o0] "v"(o0), [o1] "v"(o1), [o2] "v"(o2));
        }
        if (v == 0u) { // rare: no row in this lane, a degenerate total, rounding (a stored group word is never 0)
            uint32_t L = 0;
#pragma unroll
            for (int i = 0; i < NG; ++i) L += sell_group_hits(gw[i]);
            auto off_of = [&](uint32_t j) -> uint32_t {
                uint32_t r = gw[0];
                asm("" : "+v"(r));
#pragma unroll
                for (int i = 1; i < NG; ++i) { r = ((j >> 2) == (uint32_t)i) ? gw[i] : r; asm("" : "+v"(r)); }
                return ((r >> (8u * (j & 3u))) & 0xffu) << 3;
            };
            const double tc = P[NG - 1];
            if (L == 0) sel = (uint32_t)WIN * 8u; // the count of the pad slot, which is never flushed
            else if (!(tc > 0.0) || !(tc < __builtin_huge_val())) {
                const uint32_t j = (uint32_t)(u32_unit(x) * (double)L);
                sel = off_of(j < L ? j : L - 1);
            } else sel = off_of(L - 1); // rounding left target >= total: the last real hit
        }
        atomicAdd((int32_t *)((char *)s_cnt + rep_off + sel), 1);
    };

    // FAR (a tag type): the walk of a far tile -- the same code plus the far list behind the window part; a separate instantiation, so
    // the walk of a fast tile carries none of it
    auto walk = [&](const SellTile &d, const Buf &bf, uint32_t which, auto far_tag) {
        constexpr bool FAR = decltype(far_tag)::value;
        const uint32_t ng = d.ng();                                    // uniform
        const uint32_t *__restrict__ src = (const uint32_t *)(stream + d.off16 * 16) + lane; // groups beyond the cached ones
        // the row's window hits, counted when a path asks for them (never on the path of an ordinary draw)
        auto row_len = [&]() -> uint32_t {
            uint32_t n = 0;
#define SELL_CNT(i) if ((uint32_t)i < ng) n += sell_group_hits(bf.g##i);
            SELL_GROUPS(SELL_CNT)
#undef SELL_CNT
            for (uint32_t g = NGC; g < ng; ++g) n += sell_group_hits(src[(size_t)g * 64]);
            return n;
        };
        const uint32_t xrow = which ? xrowB : xrowA;
        double t = 0.0;
        double m0 = 0.0; // FARPF: the weight of the lane's first far hit, requested before the window part is walked
        if (FARPF && FAR) { if (bf.lf) m0 = gmu[bf.f0]; }
        // Tiles of four or more groups (most: rows are sorted by length) run their first four groups without the per-group
        // test: a wave issues one instruction per slot, scalar compares and branches included, and they were a third of a tile's
        // instructions.
        double P0, P1, P2, P3, P4, P5, P6, P7;
#define SELL_ADD(i)                                                                                          \
        {                                                                                                    \
            const uint32_t v = bf.g##i;                                                                      \
            const double w0 = wo(SELL_OFF0(v)), w1 = wo(SELL_OFF1(v)), w2 = wo(SELL_OFF2(v)), w3 = wo(SELL_OFF3(v)); \
            __builtin_amdgcn_sched_barrier(0); /* one wait for the four gathers instead of one per addition */ \
            if (i == 0) t = w0; else t += w0; /* 0.0 + w0 == w0 exactly */                                   \
            t += w1; t += w2; t += w3;                                                                       \
        }
        // A boundary past the tile's last group is never looked at -- the sweep of the pick is entered at group ng - 1 -- except P7, the
        // total the tail of a long row continues from: no copies for the others,
        // an empty asm "defines" them (the sweep's operand list names all eight).
#define SELL_SUM_U(i) SELL_ADD(i) P##i = t;
#define SELL_SUM_C(i) if ((uint32_t)i < ng) { SELL_ADD(i) P##i = t; } else asm volatile("" : "=v"(P##i));
        if (ng >= 4) {
            SELL_SUM_U(0) SELL_SUM_U(1) SELL_SUM_U(2) SELL_SUM_U(3)
            SELL_SUM_C(4) SELL_SUM_C(5) SELL_SUM_C(6)
            if (7u < ng) { SELL_ADD(7) }
            P7 = t;
        } else {
            SELL_SUM_C(0) SELL_SUM_C(1) SELL_SUM_C(2)
            P7 = t;
            asm volatile("" : "=v"(P3), "=v"(P4), "=v"(P5), "=v"(P6));
        }
#undef SELL_SUM_U
#undef SELL_SUM_C
#undef SELL_ADD
#pragma unroll 1
        for (uint32_t g = NGC; g < ng; ++g) { // rows of more than 4 * NGC hits: rare, keep it small
            const uint32_t v = src[(size_t)g * 64];
            const double w0 = wo(SELL_OFF0(v)), w1 = wo(SELL_OFF1(v)), w2 = wo(SELL_OFF2(v)), w3 = wo(SELL_OFF3(v));
            t += w0; t += w1; t += w2; t += w3;
        }
        // far tile: the lane's hits outside the window follow in its far list (stored order: window hits first, mmg_types.h)
        uint32_t Lf = 0;
        const uint32_t *__restrict__ farp = nullptr;
        if (FAR) {
            const uint8_t *__restrict__ fb = stream + d.off16 * 16 + (size_t)ng * 256;
            farp = (const uint32_t *)(fb + 64) + lane;
            if (FARPF) {
                Lf = bf.lf;
                if (Lf) t += m0;
                for (uint32_t f = 1; f < Lf; ++f) t += gmu[farp[(size_t)f * 64]];
            } else {
                Lf = fb[lane];
                for (uint32_t f = 0; f < Lf; ++f) t += gmu[farp[(size_t)f * 64]];
            }
        }
        const uint32_t L_k = HAS_K ? row_len() : 0u, L = L_k; // (without multiplicities only the draw's rare path needs it)
        if (HAS_K && L + Lf == 0) return; // (without multiplicities an empty row falls out of the draw's rare path: one test less per tile)
        uint32_t farc = 0; // the transcript of a pick from the far list (draw() then returns FAR_PICK)
        constexpr uint32_t FAR_PICK = 0xffffffffu;
        // offsets of group g of this lane's row
        // 8-way selections: one compare-and-select per candidate with an empty asm between the steps (a visible chain or
        // tree of selects is rewritten into a dynamically indexed stack table, i.e. scratch memory traffic per row)
        auto group_of = [&](uint32_t g) -> uint32_t {
            uint32_t r = bf.g0;
            asm("" : "+v"(r));
#define SELL_SEL(i) { r = (g == (uint32_t)i) ? bf.g##i : r; asm("" : "+v"(r)); }
            SELL_GROUPS(SELL_SEL)
#undef SELL_SEL
            if (g >= (uint32_t)NGC) r = src[(size_t)g * 64];
            return r;
        };
        auto off_of = [&](uint32_t j) -> uint32_t { return ((group_of(j >> 2) >> (8u * (j & 3u))) & 0xffu) << 3; };
        auto add = [&](uint32_t off, int32_t x) {
            if (FAR && off == FAR_PICK) far_add(farc, x);
            else atomicAdd((int32_t *)((char *)s_cnt + rep_off + off), x);
        };
        const uint32_t kk = HAS_K ? bf.kk : 1u;
        if (HAS_K && kk == 0) return;
        // a single hit needs no draw; without multiplicities the general path picks it anyway (same result, one branch less)
        if (HAS_K && L == 1) { add(SELL_OFF0(bf.g0), (int32_t)kk); return; }
        const bool degenerate = !(t > 0.0) || !(t < __builtin_huge_val());
        const uint64_t row_id = a.row_id_base + d.r0 + lane;
        // one categorical draw: the window byte offset of the selected hit (allocate_row's pick + col)
        const double ts = t * 0x1p-32, hs = ts * 0.5; // draw_target (mmg_math.h): the target of word x is fma(x, ts, hs)
        // one categorical draw from the random word x: the window byte offset of the selected hit (allocate_row's pick + col)
        auto draw = [&](uint32_t x) -> uint32_t {
            const double target = draw_target(x, ts, hs);
            // First cached boundary above the target.  The boundaries never decrease, so the lanes whose target lies below boundary
            // i are a subset of those below boundary i + 1: v_cmpx narrows EXEC boundary by boundary, and plain moves under the
            // narrowed mask leave every lane with the words of the FIRST boundary above its target -- three instructions per
            // boundary (compare, 32-bit move, 64-bit move), no select (a v_cndmask_b32 that reads VCC costs 23 clocks on gfx950
            // unless it issues right behind the compare that wrote it: tools/issue_bench.hip).  Boundaries past the tile's last
            // group repeat the total: the sweep is entered at the last group (ng is uniform, one scalar branch).
            uint32_t v;
            double acc;
            {
                uint64_t sv, tm;
#define SELL_STEP(i, prev) "v_cmpx_lt_f64_e64 %[tm], %[t], %[p" #i "]\n\t" "v_mov_b32 %[v], %[g" #i "]\n\t" "v_mov_b64 %[acc], " prev "\n\t"
                asm volatile("s_mov_b64 %[sv], exec\n\t"
                             "v_mov_b32 %[v], 0\n\t"
                             "v_mov_b64 %[acc], 0\n\t"
                             "s_cmp_ge_u32 %[ng], 8\n\t" "s_cbranch_scc1 .Lsell_b7_%=\n\t"
                             "s_cmp_eq_u32 %[ng], 7\n\t" "s_cbranch_scc1 .Lsell_b6_%=\n\t"
                             "s_cmp_eq_u32 %[ng], 6\n\t" "s_cbranch_scc1 .Lsell_b5_%=\n\t"
                             "s_cmp_eq_u32 %[ng], 5\n\t" "s_cbranch_scc1 .Lsell_b4_%=\n\t"
                             "s_cmp_eq_u32 %[ng], 4\n\t" "s_cbranch_scc1 .Lsell_b3_%=\n\t"
                             "s_cmp_eq_u32 %[ng], 3\n\t" "s_cbranch_scc1 .Lsell_b2_%=\n\t"
                             "s_cmp_eq_u32 %[ng], 2\n\t" "s_cbranch_scc1 .Lsell_b1_%=\n\t"
                             "s_branch .Lsell_b0_%=\n"
                             ".Lsell_b7_%=:\n\t" SELL_STEP(7, "%[p6]")
                             ".Lsell_b6_%=:\n\t" SELL_STEP(6, "%[p5]")
                             ".Lsell_b5_%=:\n\t" SELL_STEP(5, "%[p4]")
                             ".Lsell_b4_%=:\n\t" SELL_STEP(4, "%[p3]")
                             ".Lsell_b3_%=:\n\t" SELL_STEP(3, "%[p2]")
                             ".Lsell_b2_%=:\n\t" SELL_STEP(2, "%[p1]")
                             ".Lsell_b1_%=:\n\t" SELL_STEP(1, "%[p0]")
                             ".Lsell_b0_%=:\n\t" SELL_STEP(0, "0")
                             "s_mov_b64 exec, %[sv]"
                             : [v] "=&v"(v), [acc] "=&v"(acc), [sv] "=&s"(sv), [tm] "=&s"(tm)
                             : [t] "v"(target), [ng] "s"(ng), [p0] "v"(P0), [p1] "v"(P1), [p2] "v"(P2), [p3] "v"(P3), [p4] "v"(P4), [p5] "v"(P5),
                               [p6] "v"(P6), [p7] "v"(P7), [g0] "v"(bf.g0), [g1] "v"(bf.g1), [g2] "v"(bf.g2), [g3] "v"(bf.g3), [g4] "v"(bf.g4),
                               [g5] "v"(bf.g5), [g6] "v"(bf.g6), [g7] "v"(bf.g7)
                             : "scc");
#undef SELL_STEP
            }
            // a stored group word is never 0 (four ascending offsets, pads 255): v == 0 <=> no boundary lies above the target
            const bool hit = v != 0u;
            // resolved inside the group without a branch: the three gathers go out together, the same narrowing picks the hit;
            // lanes without a hit (below) read slot 0 and are overridden
            const uint32_t o0 = SELL_OFF0(v), o1 = SELL_OFF1(v), o2 = SELL_OFF2(v), o3 = SELL_OFF3(v);
            double w0 = wo(o0), w1 = wo(o1), w2 = wo(o2);
            asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2)); // all three requested before the first is waited for
            const double p0 = acc + w0, p1 = p0 + w1, p2 = p1 + w2;
            uint32_t sel = o3;
            {
                uint64_t sv, tm;
                asm volatile("s_mov_b64 %[sv], exec\n\t"
                             "v_cmpx_lt_f64_e64 %[tm], %[t], %[p2]\n\t" "v_mov_b32 %[sel], %[o2]\n\t"
                             "v_cmpx_lt_f64_e64 %[tm], %[t], %[p1]\n\t" "v_mov_b32 %[sel], %[o1]\n\t"
                             "v_cmpx_lt_f64_e64 %[tm], %[t], %[p0]\n\t" "v_mov_b32 %[sel], %[o0]\n\t"
                             "s_mov_b64 exec, %[sv]"
                             : [sel] "+&v"(sel), [sv] "=&s"(sv), [tm] "=&s"(tm)
                             : [t] "v"(target), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [o0] "v"(o0), [o1] "v"(o1), [o2] "v"(o2));
            }
            if (!hit) { // rare: an empty row, a degenerate total (0, inf, NaN never compare below anything), a row of more than 32 hits, rounding
                const uint32_t L = HAS_K ? L_k : row_len();
                if (L + Lf == 0) sel = (uint32_t)WIN * 8u; // no row in this lane: the count of the pad slot, which is never flushed
                else if (degenerate) {
                    const uint32_t Lt = L + Lf;
                    uint32_t j = (uint32_t)(u32_unit(x) * (double)Lt);
                    j = j < Lt ? j : Lt - 1;
                    if (!FAR || j < L) sel = off_of(j);
                    else { farc = farp[(size_t)(j - L) * 64]; sel = FAR_PICK; }
                } else {
                    double accl = P7;
                    bool found = false;
#pragma unroll 1
                    for (uint32_t g = NGC; g < ng && !found; ++g) {
                        const uint32_t vv = src[(size_t)g * 64];
                        const uint32_t q0 = SELL_OFF0(vv), q1 = SELL_OFF1(vv), q2 = SELL_OFF2(vv), q3 = SELL_OFF3(vv);
                        const double r0 = accl + wo(q0), r1 = r0 + wo(q1), r2 = r1 + wo(q2), r3 = r2 + wo(q3);
                        if (target < r3) { sel = target < r0 ? q0 : (target < r1 ? q1 : (target < r2 ? q2 : q3)); found = true; }
                        accl = r3;
                    }
                    for (uint32_t f = 0; FAR && f < Lf && !found; ++f) { // the far list continues the row
                        const uint32_t c = (FARPF && f == 0) ? bf.f0 : farp[(size_t)f * 64];
                        accl += (FARPF && f == 0) ? m0 : gmu[c];
                        if (target < accl) { farc = c; sel = FAR_PICK; found = true; }
                    }
                    if (!found) { // rounding left target >= total: the last real hit
                        if (FAR && Lf) { farc = farp[(size_t)(Lf - 1) * 64]; sel = FAR_PICK; }
                        else sel = off_of(L - 1);
                    }
                }
            }
            return sel;
        };
        if (!HAS_K) { add(draw(xrow), 1); return; }
        if (draws_categoricals(kk, L)) {
            Stream2 s(a.seed, a.chain, TAG_ROW, row_id, a.iter);
            {
#pragma unroll 1
                for (uint32_t dd = 0; dd < kk; ++dd) add(draw(s.next_word()), 1);
            }
            return;
        }
        // (a row on the conditional-binomial chain is not drawn here: k_sample_bigk walks the list of those rows, bigk_kernels.h)
    };
#undef SELL_GROUPS
#undef SELL_OFF0
#undef SELL_OFF1
#undef SELL_OFF2
#undef SELL_OFF3

    auto slow_tile = [&](const SellTile &d) {
        // rows straight from the 32-bit CSR (window lookups / LDS counts where possible)
        const uint32_t wbase = d.wbase;
        if (lane < d.nrows()) {
            const uint64_t st = (uint64_t)row_ptr[d.r0 + lane];
            const uint32_t L = (uint32_t)((uint64_t)row_ptr[d.r0 + lane + 1] - st);
            auto add = [&](uint32_t col, int32_t x) {
                const uint32_t dd = col - wbase;
                if (dd < (uint32_t)WIN) atomicAdd(&s_cnt[2 * dd], x);
                else far_add(col, x);
            };
            RowViewGlobalWin<WIN> v{col_idx + st, L, wbase, s_mu, gmu};
            allocate_row<HAS_K, false>(v, add, HAS_K ? kmult[d.r0 + lane] : 1u, a, a.row_id_base + d.r0 + lane);
        }
    };

    auto far_tile = [&](const SellTile &d) {
        if (lane < d.nrows()) {
            const uint8_t *__restrict__ blk = stream + d.off16 * 16, *__restrict__ fb = blk + (size_t)d.ng() * 256;
            const uint32_t wbase = d.wbase;
            uint32_t Ln = 0;
            for (uint32_t g = 0; g < d.ng(); ++g) Ln += sell_group_hits(((const uint32_t *)blk + lane)[(size_t)g * 64]);
            auto add = [&](uint32_t col, int32_t x) {
                const uint32_t dd = col - wbase;
                if (dd < (uint32_t)WIN) atomicAdd(&s_cnt[2 * dd], x);
                else far_add(col, x);
            };
            const RowViewFarTile v{(const uint32_t *)blk + lane, (const uint32_t *)(fb + 64) + lane, Ln, Ln + fb[lane], wbase, s_mu, gmu};
            allocate_row<HAS_K, false>(v, add, HAS_K ? kmult[d.r0 + lane] : 1u, a, a.row_id_base + d.r0 + lane);
        }
    };

    auto process = [&](const SellTile &d, uint32_t &cur_base, const SellTile &refill, Buf &bf, uint32_t which) {
        if (d.flags() & SELL_EMPTY) { issue(refill, bf); return; }
        if (d.wbase != cur_base) {
            __syncthreads();
            flush_window(cur_base);
            load_window(d.wbase);
            cur_base = d.wbase;
            __syncthreads();
        }
        if (d.flags() & SELL_FAST) {
            if constexpr (!HAS_K && !FARPF && FIXW) {
                switch (d.ng()) { // uniform
                case 1: walk_fixed(d, bf, which, IntTag<1>()); break;
                case 2: walk_fixed(d, bf, which, IntTag<2>()); break;
                case 3: walk_fixed(d, bf, which, IntTag<3>()); break;
                case 4: walk_fixed(d, bf, which, IntTag<4>()); break;
                case 5: walk_fixed(d, bf, which, IntTag<5>()); break;
                case 6: walk_fixed(d, bf, which, IntTag<6>()); break;
                case 7: walk_fixed(d, bf, which, IntTag<7>()); break;
                case 8: walk_fixed(d, bf, which, IntTag<8>()); break;
                default: walk(d, bf, which, BoolTag<false>()); break; // rows of more than 32 hits
                }
            } else walk(d, bf, which, BoolTag<false>());
        }
        else if (!HAS_K && (d.flags() & SELL_FAR)) walk(d, bf, which, BoolTag<true>());
        else if (d.flags() & SELL_FAR) far_tile(d); // with multiplicities: the generic row walk over the tile's block
        else slow_tile(d);
        issue(refill, bf); // the registers are free again only now: tile i+2 travels while tile i+1 is walked
    };

    SellTile none;
    none.off16 = 0; none.r0 = 0; none.wbase = 0; none.meta = sell_meta(0, 0, SELL_EMPTY);
    auto tile_at = [&](uint32_t i) {
        SellTile d = T[min(i, nt - 1u)];
        d.meta = i < nt ? d.meta : none.meta;
        return d;
    };
    SellTile dA = d_first, dB = d_second; // (the host marks the second descriptor empty in a range of one tile)
    Buf bufA, bufB;
    load_window(dA.wbase);
    uint32_t cur_base = dA.wbase;
    __syncthreads();
    issue(dA, bufA);
    __builtin_amdgcn_sched_barrier(0); // A's block is requested before B's, here as in the loop (sell_multi_kernels.h): the waits are by count
    issue(dB, bufB);
    for (uint32_t i = 0; i < nt; i += 2) {
        const SellTile nA = tile_at(i + 2), nB = tile_at(i + 3);
        if (!HAS_K) pair_rng(dA, dB);
        process(dA, cur_base, nA, bufA, 0);
        process(dB, cur_base, nB, bufB, 1);
        dA = nA;
        dB = nB;
    }
    __syncthreads();
    flush_window(cur_base);
    for (uint32_t i = lane; i < FAR_SLOTS; i += 64) {
        const int32_t v = s_fcnt[i];
        if (v) global_count_add(gcnt, s_fkey[i], v);
    }
}

} // namespace mmg
