// K1 for NCH chains at once on the sliced-ELL stream (BASELINE configs[2]: several chains in one GPU).
//
// A tile's instructions are one third scalar bookkeeping (descriptors, branches, waits), one tenth stream and unpack work, and
// its LDS gathers are limited by instruction count, not bytes: all of that is independent of the chain.  k_sample_sell_multi
// walks a tile ONCE for NCH chains: the window holds mu interleaved [index][chain], so one 16-byte LDS read returns a hit's
// weight for two chains; the byte unpack, the stream loads and every scalar instruction are paid once.  Per chain remain the
// fp64 prefix additions, one Philox block per row pair, and the pick.  Every chain performs exactly the additions, draws and
// comparisons of k_sample_sell with its own key: chain c of a fused launch equals a single-chain launch bit for bit.
// Multiplicities (k != NULL) keep the single-chain kernel.
#pragma once

namespace mmg {

template <typename IdxT, int NCH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NCH == 2 ? 4 : 2, NCH == 2 ? 4 : 2))) void k_sample_sell_multi(const IdxT *__restrict__ row_ptr, const uint32_t *__restrict__ col_idx,
                                                          const SellTile *__restrict__ tiles, const uint64_t *__restrict__ chunk_tile,
                                                          const double *__restrict__ gmu /* [NCH][n] */, const uint8_t *__restrict__ stream,
                                                          int32_t *gcnt /* [NCH][n] */, SampleArgs a)
{
    static_assert(NCH == 2 || NCH == 4, "chains are fused in pairs or fours");
    constexpr int WIN = (int)SELL_WIN;
    constexpr int SH = 3;                                      // log2 of the bytes per window entry
    // One window per chain, CS entries apart: a hit's weights are NCH 8-byte reads at one offset register and NCH immediate offsets.
    // (Interleaved [index][chain] entries read with one ds_read_b128 per pair of chains moved the same bytes with half the
    // instructions, but at 51 % bank-conflict cycles where the 8-byte reads of k_sample_sell see 40 %: the fused kernel is bound by
    // the LDS pipe, so the conflicts decide.  CS = WIN + 2: not a multiple of 64 entries, or the compiler merges the reads of two
    // chains into one ds_read2st64_b64, which behaves like the 16-byte read.)  Entry [WIN] stays 0.0: what pad slots read.
    constexpr int CS = WIN + 2;
    __shared__ __attribute__((aligned(16))) double s_mu[CS * NCH];
    // counts laid out like the weights, 8 bytes per entry (upper word unused): the offset that gathered a hit's weight, plus chain c's
    // window offset, addresses chain c's count of it -- no address arithmetic per pick
    __shared__ int32_t s_cnt[2 * CS * NCH];
    const uint32_t lane = threadIdx.x;
    // grid.y = group of NCH chains: one launch advances all the fused chains of a sampler (the tail of one group overlaps the head of
    // the next instead of a launch boundary)
    gmu += (size_t)blockIdx.y * NCH * a.n;
    gcnt += (size_t)blockIdx.y * NCH * a.n + (size_t)(blockIdx.x & a.cnt_rep_mask) * a.cnt_rep_stride; // (mmg_types.h: CNT_REPLICAS)
    a.chain += blockIdx.y * (uint32_t)NCH;

    // the range's header (mmgibbs.hip: upload_ranges): first tile, end tile, the descriptors of its first two tiles -- one scalar load
    const uint64_t *__restrict__ hdr = chunk_tile + (size_t)blockIdx.x * 8;
    const uint64_t t_begin = hdr[0], t_end = hdr[1];
    if (t_begin >= t_end) return;
    SellTile d_first, d_second;
    d_first.off16 = hdr[2]; d_first.r0 = hdr[3]; d_first.wbase = (uint32_t)hdr[4]; d_first.meta = (uint32_t)(hdr[4] >> 32);
    d_second.off16 = hdr[5]; d_second.r0 = hdr[6]; d_second.wbase = (uint32_t)hdr[7]; d_second.meta = (uint32_t)(hdr[7] >> 32);
    const uint32_t nt = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(t_end - t_begin));
    const SellTile *__restrict__ T = tiles + t_begin;

    for (int i = lane; i < 2 * CS * NCH; i += 64) s_cnt[i] = 0;
    if (lane < NCH) s_mu[lane * CS + WIN] = 0.0;

    auto flush_window = [&](uint32_t base) {
        for (int i = lane; i < WIN; i += 64) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int32_t v = s_cnt[2 * (c * CS + i)];
                s_cnt[2 * (c * CS + i)] = 0;
                if (v) global_count_add(gcnt + (size_t)c * a.n, base + (uint32_t)i, v);
            }
        }
    };
    auto load_window = [&](uint32_t base) {
        for (int i = lane; i < WIN; i += 64) {
            const uint32_t col = base + (uint32_t)i;
#pragma unroll
            for (int c = 0; c < NCH; ++c) s_mu[c * CS + i] = col < a.n ? gmu[(size_t)c * a.n + col] : 0.0;
        }
    };
    // byte k of a group word as the LDS byte offset of the window entry (all chains)
    const uint32_t shv = (uint32_t)SH;
    auto sdwa_off = [&](uint32_t v, auto sel_tag) -> uint32_t { // one instruction per byte, as in k_sample_sell
        constexpr int K = decltype(sel_tag)::value;
        uint32_t o;
        if (K == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(o) : "v"(shv), "v"(v));
        else if (K == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(o) : "v"(shv), "v"(v));
        else if (K == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(o) : "v"(shv), "v"(v));
        else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(o) : "v"(shv), "v"(v));
        return o;
    };
#define SM_OFF0(v) sdwa_off(v, IntTag<0>())
#define SM_OFF1(v) sdwa_off(v, IntTag<1>())
#define SM_OFF2(v) sdwa_off(v, IntTag<2>())
#define SM_OFF3(v) sdwa_off(v, IntTag<3>())
    struct W { double c[NCH]; };
    auto wo = [&](uint32_t off) {
        W r;
        const double *p = (const double *)((const char *)s_mu + off);
#pragma unroll
        for (int c = 0; c < NCH; ++c) r.c[c] = p[c * CS];
        return r;
    };

#define SM_GROUPS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
    struct Buf {
        uint32_t g0, g1, g2, g3, g4, g5, g6, g7;
    };
    auto issue = [&](const SellTile &d, Buf &bf) {
        const bool fast = d.flags() & SELL_FAST; // uniform
        const SellBlock blk(stream + (fast ? d.off16 * 16 : 0), d.meta);
#define SM_ISSUE(i) bf.g##i = blk.template group<i>(lane);
        SM_GROUPS(SM_ISSUE)
#undef SM_ISSUE
    };

    // pair RNG per chain (k_sample_sell): this lane's random word for its row of tile A / tile B
    uint32_t xrowA[NCH], xrowB[NCH];
    auto pair_rng = [&](const SellTile &A, const SellTile &B) {
        const uint64_t qa = (a.row_id_base + A.r0) >> 1, qb = (a.row_id_base + B.r0) >> 1; // uniform
        const uint32_t l5 = lane & 31u;
        const bool one_key = (((a.row_id_base + A.r0) ^ (a.row_id_base + B.r0 + 63u)) >> 33) == 0 && A.r0 <= B.r0;
        const uint64_t q = (lane < 32u ? qa : qb) + l5;
        const uint32_t pa = ((uint32_t)(a.row_id_base + A.r0) & 1u) + lane, pb = ((uint32_t)(a.row_id_base + B.r0) & 1u) + lane;
        const int sa = (int)((pa >> 1) << 2), sb = (int)((32u + (pb >> 1)) << 2);
        if (one_key && (((a.row_id_base + A.r0) | (a.row_id_base + B.r0)) & 1u) == 0) { // both tiles start on an even row id: hand-out by DPP (k_sample_sell)
            // the chains' blocks round by round, side by side: a block is one dependent chain (multiply -> xor -> multiply ...), and a
            // wave that issues it alone waits out every result (and a hazard slot behind every v_mad_u64_u32)
            uint32_t x0[NCH], x1[NCH], key[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                x0[c] = ((lane & 1u) ? (uint32_t)qb : (uint32_t)qa) + (lane >> 1);
                x1[c] = a.iter;
                key[c] = stream2_key(a.seed, a.chain + (uint32_t)c, TAG_ROW, (uint32_t)(qa >> 32));
            }
#pragma unroll
            for (int r = 0; r < 10; ++r) {
                uint64_t pr[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(pr[c]) : "v"(x0[c]), "s"(0xD256D193u) : "vcc");
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    x0[c] = __builtin_amdgcn_bitop3_b32((uint32_t)(pr[c] >> 32), x1[c], key[c], 0x96);
                    x1[c] = (uint32_t)pr[c];
                    key[c] += 0x9E3779B9u;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const uint32_t n0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)x0[c], 0xB1, 0xF, 0xF, true);
                const uint32_t n1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)x1[c], 0xB1, 0xF, 0xF, true);
                xrowA[c] = (lane & 1u) ? n1 : x0[c];
                xrowB[c] = (lane & 1u) ? x1[c] : n0;
            }
            return;
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            uint32_t x0 = (uint32_t)q, x1 = a.iter;
            if (one_key) philox2x32_10(x0, x1, stream2_key(a.seed, a.chain + (uint32_t)c, TAG_ROW, (uint32_t)(qa >> 32)));
            else philox2x32_10(x0, x1, stream2_key(a.seed, a.chain + (uint32_t)c, TAG_ROW, (uint32_t)(q >> 32)));
            const uint32_t a0 = (uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)x0), a1 = (uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)x1);
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_ds_bpermute(sb, (int)x0), b1 = (uint32_t)__builtin_amdgcn_ds_bpermute(sb, (int)x1);
            xrowA[c] = (pa & 1u) ? a1 : a0;
            xrowB[c] = (pb & 1u) ? b1 : b0;
        }
    };


    // ---- the walk of a register-path tile of at most 8 groups, written out per group count (chain pairs only) --------------------
    // The generic walk below is what the compiler makes of one body for every ng: a wave issues a group's gathers, waits for all of
    // them, adds, tests ng, and walks the two chains' picks one behind the other -- about seven LDS round trips per tile that nothing
    // of the SAME wave overlaps, with four waves per SIMD to hide them (counters: 34 % of the wave cycles at s_waitcnt, VALU 0.70 and
    // LDS 0.68 busy).  Here a tile's ng selects straight-line code: the gathers of group i + 1 are in flight while group i is added
    // (two register sets), no boundary is copied, the sweep of the pick has no entry branches, and the two chains' picks are
    // interleaved -- chain 1's sweep runs while chain 0's in-group gathers travel.  Same additions, comparisons and draws in the same
    // order per chain: bit-identical to the generic walk (tests/test_gpu_parity.py, test_gpu_fullsize.py).
    // the weights of a group's four hits for the two chains: 4 unpacks, 8 gathers
    // (The gathers are plain loads, waited for by the compiler's counts: one s_waitcnt in front of every addition.  Issuing them from
    // inline asm with ONE hand-placed wait per group of additions saves 38 of a tile's ~350 instruction slots and measured +0.5 % --
    // and is unsafe: the compiler believes an asm's result is there at once, so a register copy it places between the load and the
    // wait (a phi at the end of a switch case, a live-range split at 127 VGPRs) reads the register before the data arrives.  Seen as
    // a bit difference in one chain of the full-size parity test, one run in three.  Not kept.)
    struct G4 { double w[4][2]; };
    auto lds_mu = [&](uint32_t off, auto chain_tag) -> double {
        constexpr int c = decltype(chain_tag)::value;
        return *(const double *)((const char *)(s_mu + c * CS) + off);
    };
    auto gather = [&](uint32_t v) {
        G4 r;
        const uint32_t o0 = SM_OFF0(v), o1 = SM_OFF1(v), o2 = SM_OFF2(v), o3 = SM_OFF3(v);
        const uint32_t o[4] = {o0, o1, o2, o3};
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            r.w[h][0] = lds_mu(o[h], IntTag<0>());
            r.w[h][1] = lds_mu(o[h], IntTag<1>());
        }
        return r;
    };
    // head: the weights of the tile's FIRST group, requested by the caller as early as the window allows (behind the Philox rounds of the
    // pair for tile A, before the picks of tile A for tile B); between(): called between the sums and the picks
    auto walk_fixed = [&](const SellTile &d, const Buf &bf, uint32_t which, auto ng_tag, const G4 &head, auto &&between) {
        constexpr int NG = decltype(ng_tag)::value;
        static_assert(NCH == 2 && NG >= 1 && NG <= 8, "chain pairs, cached groups only");
        const uint32_t gw[8] = {bf.g0, bf.g1, bf.g2, bf.g3, bf.g4, bf.g5, bf.g6, bf.g7};
        double P[NG][2];
        {
            G4 cur = head, nxt = cur;
            if (NG > 1) nxt = gather(gw[1]);
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                __builtin_amdgcn_sched_barrier(0);
                {   // the two chains' sums side by side: each is a dependent chain of additions
                    double t0 = i == 0 ? cur.w[0][0] : P[i - 1][0] + cur.w[0][0]; // 0.0 + w == w exactly
                    double t1 = i == 0 ? cur.w[0][1] : P[i - 1][1] + cur.w[0][1];
                    t0 += cur.w[1][0]; t1 += cur.w[1][1];
                    t0 += cur.w[2][0]; t1 += cur.w[2][1];
                    t0 += cur.w[3][0]; t1 += cur.w[3][1];
                    P[i][0] = t0; P[i][1] = t1;
                }
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
                if (i + 2 < NG) nxt = gather(gw[i + 2]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        between();
        __builtin_amdgcn_sched_barrier(0);
        uint32_t x[2], v[2], sel[2], oo[2][4];
        double target[2], acc[2], wi[2][3];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            x[c] = which ? xrowB[c] : xrowA[c];
            const double ts = P[NG - 1][c] * 0x1p-32, hs = ts * 0.5;
            target[c] = draw_target(x[c], ts, hs); // mmg_math.h
        }
#define SMF_STEP(i, prev) "v_cmpx_lt_f64_e64 %[tm], %[t], %[p" #i "]\n\t" "v_mov_b32 %[v], %[g" #i "]\n\t" "v_mov_b64 %[acc], " prev "\n\t"
#define SMF_HEAD "s_mov_b64 %[sv], exec\n\t" "v_mov_b32 %[v], 0\n\t" "v_mov_b64 %[acc], 0\n\t"
#define SMF_TAIL "s_mov_b64 exec, %[sv]"
#define SMF_OUT(c) [v] "=&v"(v[c]), [acc] "=&v"(acc[c]), [sv] "=&s"(sv), [tm] "=&s"(tm)
#define SMF_IN(c, n) [t] "v"(target[c]), [p0] "v"(P[0][c]), [p1] "v"(P[n > 1 ? 1 : 0][c]), [p2] "v"(P[n > 2 ? 2 : 0][c]), [p3] "v"(P[n > 3 ? 3 : 0][c]), \
                     [p4] "v"(P[n > 4 ? 4 : 0][c]), [p5] "v"(P[n > 5 ? 5 : 0][c]), [p6] "v"(P[n > 6 ? 6 : 0][c]), [p7] "v"(P[n > 7 ? 7 : 0][c]),          \
                     [g0] "v"(gw[0]), [g1] "v"(gw[n > 1 ? 1 : 0]), [g2] "v"(gw[n > 2 ? 2 : 0]), [g3] "v"(gw[n > 3 ? 3 : 0]), [g4] "v"(gw[n > 4 ? 4 : 0]), \
                     [g5] "v"(gw[n > 5 ? 5 : 0]), [g6] "v"(gw[n > 6 ? 6 : 0]), [g7] "v"(gw[n > 7 ? 7 : 0])
#define SMF_S0 SMF_STEP(0, "0")
#define SMF_S1 SMF_STEP(1, "%[p0]") SMF_S0
#define SMF_S2 SMF_STEP(2, "%[p1]") SMF_S1
#define SMF_S3 SMF_STEP(3, "%[p2]") SMF_S2
#define SMF_S4 SMF_STEP(4, "%[p3]") SMF_S3
#define SMF_S5 SMF_STEP(5, "%[p4]") SMF_S4
#define SMF_S6 SMF_STEP(6, "%[p5]") SMF_S5
#define SMF_S7 SMF_STEP(7, "%[p6]") SMF_S6
        auto sweep = [&](auto c_tag) { // first boundary above the chain's target: its group word and the prefix before it (k_sample_sell: draw)
            constexpr int c = decltype(c_tag)::value;
            uint64_t sv, tm;
            if constexpr (NG == 1) asm volatile(SMF_HEAD SMF_S0 SMF_TAIL : SMF_OUT(c) : SMF_IN(c, 1));
            else if constexpr (NG == 2) asm volatile(SMF_HEAD SMF_S1 SMF_TAIL : SMF_OUT(c) : SMF_IN(c, 2));
            else if constexpr (NG == 3) asm volatile(SMF_HEAD SMF_S2 SMF_TAIL : SMF_OUT(c) : SMF_IN(c, 3));
            else if constexpr (NG == 4) asm volatile(SMF_HEAD SMF_S3 SMF_TAIL : SMF_OUT(c) : SMF_IN(c, 4));
            else if constexpr (NG == 5) asm volatile(SMF_HEAD SMF_S4 SMF_TAIL : SMF_OUT(c) : SMF_IN(c, 5));
            else if constexpr (NG == 6) asm volatile(SMF_HEAD SMF_S5 SMF_TAIL : SMF_OUT(c) : SMF_IN(c, 6));
            else if constexpr (NG == 7) asm volatile(SMF_HEAD SMF_S6 SMF_TAIL : SMF_OUT(c) : SMF_IN(c, 7));
            else asm volatile(SMF_HEAD SMF_S7 SMF_TAIL : SMF_OUT(c) : SMF_IN(c, 8));
            // inside the group: the three gathers go out now and are waited for in resolve()
            oo[c][0] = SM_OFF0(v[c]); oo[c][1] = SM_OFF1(v[c]); oo[c][2] = SM_OFF2(v[c]); oo[c][3] = SM_OFF3(v[c]);
#pragma unroll
            for (int h = 0; h < 3; ++h) wi[c][h] = lds_mu(oo[c][h], IntTag<c>());
        };
#undef SMF_S0
#undef SMF_S1
#undef SMF_S2
#undef SMF_S3
#undef SMF_S4
#undef SMF_S5
#undef SMF_S6
#undef SMF_S7
#undef SMF_IN
#undef SMF_OUT
#undef SMF_TAIL
#undef SMF_HEAD
#undef SMF_STEP
        auto resolve = [&](int c) {
            const double p0 = acc[c] + wi[c][0], p1 = p0 + wi[c][1], p2 = p1 + wi[c][2];
            uint32_t s = oo[c][3];
            uint64_t sv, tm;
            asm volatile("s_mov_b64 %[sv], exec\n\t"
                         "v_cmpx_lt_f64_e64 %[tm], %[t], %[p2]\n\t" "v_mov_b32 %[sel], %[o2]\n\t"
                         "v_cmpx_lt_f64_e64 %[tm], %[t], %[p1]\n\t" "v_mov_b32 %[sel], %[o1]\n\t"
                         "v_cmpx_lt_f64_e64 %[tm], %[t], %[p0]\n\t" "v_mov_b32 %[sel], %[o0]\n\t"
                         "s_mov_b64 exec, %[sv]"
                         : [sel] "+&v"(s), [sv] "=&s"(sv), [tm] "=&s"(tm)
                         : [t] "v"(target[c]), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [o0] "v"(oo[c][0]), [o1] "v"(oo[c][1]), [o2] "v"(oo[c][2]));
            sel[c] = s;
        };
        sweep(IntTag<0>());
        __builtin_amdgcn_sched_barrier(0);
        sweep(IntTag<1>());
        __builtin_amdgcn_sched_barrier(0);
        resolve(0);
        resolve(1);
        if ((v[0] == 0u) | (v[1] == 0u)) { // rare: no row in this lane, a degenerate total, rounding (a stored group word is never 0)
            uint32_t L = 0;
#pragma unroll
            for (int i = 0; i < NG; ++i) L += sell_group_hits(gw[i]);
            auto off_of = [&](uint32_t j) -> uint32_t {
                uint32_t r = gw[0];
                asm("" : "+v"(r));
#pragma unroll
                for (int i = 1; i < NG; ++i) { r = ((j >> 2) == (uint32_t)i) ? gw[i] : r; asm("" : "+v"(r)); }
                return ((r >> (8u * (j & 3u))) & 0xffu) << SH;
            };
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (v[c] != 0u) continue;
                const double tc = P[NG - 1][c];
                if (L == 0) sel[c] = (uint32_t)WIN << SH; // the count of the pad slot, which is never flushed
                else if (!(tc > 0.0) || !(tc < __builtin_huge_val())) {
                    const uint32_t j = (uint32_t)(u32_unit(x[c]) * (double)L);
                    sel[c] = off_of(j < L ? j : L - 1);
                } else sel[c] = off_of(L - 1); // rounding left target >= total: the last real hit
            }
        }
        atomicAdd((int32_t *)((char *)s_cnt + sel[0]), 1);
        atomicAdd((int32_t *)((char *)s_cnt + (CS * 8) + sel[1]), 1);
    };

    auto walk = [&](const SellTile &d, const Buf &bf, uint32_t which) {
        const uint32_t ng = d.ng();                                    // uniform
        const uint32_t *__restrict__ src = (const uint32_t *)(stream + d.off16 * 16) + lane; // groups beyond the cached ones
        auto row_len = [&]() -> uint32_t { // the row's hits, counted on the rare paths only (sell_kernels.h: sell_group_hits)
            uint32_t n = 0;
#define SM_CNT(i) if ((uint32_t)i < ng) n += sell_group_hits(bf.g##i);
            SM_GROUPS(SM_CNT)
#undef SM_CNT
            for (uint32_t g = 8; g < ng; ++g) n += sell_group_hits(src[(size_t)g * 64]);
            return n;
        };
        double t[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) t[c] = 0.0;
        double P0[NCH], P1[NCH], P2[NCH], P3[NCH], P4[NCH], P5[NCH], P6[NCH], P7[NCH];
#define SM_ADD(i)                                                                                            \
        {                                                                                                    \
            const uint32_t v = bf.g##i;                                                                      \
            const W w0 = wo(SM_OFF0(v)), w1 = wo(SM_OFF1(v)), w2 = wo(SM_OFF2(v)), w3 = wo(SM_OFF3(v));      \
            __builtin_amdgcn_sched_barrier(0); /* one wait for the four gathers */                           \
            _Pragma("unroll") for (int c = 0; c < NCH; ++c) {                                                \
                if (i == 0) t[c] = w0.c[c]; else t[c] += w0.c[c]; /* 0.0 + w == w exactly */                 \
                t[c] += w1.c[c]; t[c] += w2.c[c]; t[c] += w3.c[c];                                           \
            }                                                                                                \
        }
        // boundaries past the tile's last group are never looked at (the pick enters its sweep at group ng - 1), except P7, the total a
        // long row's tail continues from: no copies, an empty asm "defines" them (k_sample_sell)
#define SM_SET(i) _Pragma("unroll") for (int c = 0; c < NCH; ++c) P##i[c] = t[c];
#define SM_UNDEF(i) _Pragma("unroll") for (int c = 0; c < NCH; ++c) asm volatile("" : "=v"(P##i[c]));
#define SM_SUM_U(i) SM_ADD(i) SM_SET(i)
#define SM_SUM_C(i) if ((uint32_t)i < ng) { SM_ADD(i) SM_SET(i) } else { SM_UNDEF(i) }
        if (ng >= 4) {
            SM_SUM_U(0) SM_SUM_U(1) SM_SUM_U(2) SM_SUM_U(3)
            SM_SUM_C(4) SM_SUM_C(5) SM_SUM_C(6)
            if (7u < ng) { SM_ADD(7) }
            SM_SET(7)
        } else {
            SM_SUM_C(0) SM_SUM_C(1) SM_SUM_C(2)
            SM_UNDEF(3) SM_UNDEF(4) SM_UNDEF(5) SM_UNDEF(6)
            SM_SET(7)
        }
#undef SM_UNDEF
#undef SM_SUM_U
#undef SM_SUM_C
#undef SM_SET
#undef SM_ADD
#pragma unroll 1
        for (uint32_t g = 8; g < ng; ++g) { // rows of more than 32 hits: rare, keep it small
            const uint32_t v = src[(size_t)g * 64];
            const W w0 = wo(SM_OFF0(v)), w1 = wo(SM_OFF1(v)), w2 = wo(SM_OFF2(v)), w3 = wo(SM_OFF3(v));
#pragma unroll
            for (int c = 0; c < NCH; ++c) { t[c] += w0.c[c]; t[c] += w1.c[c]; t[c] += w2.c[c]; t[c] += w3.c[c]; }
        }
        auto group_of = [&](uint32_t g) -> uint32_t {
            uint32_t r = bf.g0;
            asm("" : "+v"(r));
#define SM_SEL(i) { r = (g == (uint32_t)i) ? bf.g##i : r; asm("" : "+v"(r)); }
            SM_GROUPS(SM_SEL)
#undef SM_SEL
            if (g >= 8u) r = src[(size_t)g * 64];
            return r;
        };
        auto off_of = [&](uint32_t j) -> uint32_t { return ((group_of(j >> 2) >> (8u * (j & 3u))) & 0xffu) << SH; };
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const double tc = t[c];
            const bool degenerate = !(tc > 0.0) || !(tc < __builtin_huge_val());
            const uint32_t x = which ? xrowB[c] : xrowA[c];
            const double ts = tc * 0x1p-32, hs = ts * 0.5;
            const double target = draw_target(x, ts, hs); // mmg_math.h
            // v_cmpx narrowing, entered at the tile's last group, as in k_sample_sell (sell_kernels.h: draw)
            uint32_t v;
            double acc;
            {
                uint64_t sv, tm;
#define SM_STEP(i, prev) "v_cmpx_lt_f64_e64 %[tm], %[t], %[p" #i "]\n\t" "v_mov_b32 %[v], %[g" #i "]\n\t" "v_mov_b64 %[acc], " prev "\n\t"
                asm volatile("s_mov_b64 %[sv], exec\n\t"
                             "v_mov_b32 %[v], 0\n\t"
                             "v_mov_b64 %[acc], 0\n\t"
                             "s_cmp_ge_u32 %[ng], 8\n\t" "s_cbranch_scc1 .Lsm_b7_%=\n\t"
                             "s_cmp_eq_u32 %[ng], 7\n\t" "s_cbranch_scc1 .Lsm_b6_%=\n\t"
                             "s_cmp_eq_u32 %[ng], 6\n\t" "s_cbranch_scc1 .Lsm_b5_%=\n\t"
                             "s_cmp_eq_u32 %[ng], 5\n\t" "s_cbranch_scc1 .Lsm_b4_%=\n\t"
                             "s_cmp_eq_u32 %[ng], 4\n\t" "s_cbranch_scc1 .Lsm_b3_%=\n\t"
                             "s_cmp_eq_u32 %[ng], 3\n\t" "s_cbranch_scc1 .Lsm_b2_%=\n\t"
                             "s_cmp_eq_u32 %[ng], 2\n\t" "s_cbranch_scc1 .Lsm_b1_%=\n\t"
                             "s_branch .Lsm_b0_%=\n"
                             ".Lsm_b7_%=:\n\t" SM_STEP(7, "%[p6]")
                             ".Lsm_b6_%=:\n\t" SM_STEP(6, "%[p5]")
                             ".Lsm_b5_%=:\n\t" SM_STEP(5, "%[p4]")
                             ".Lsm_b4_%=:\n\t" SM_STEP(4, "%[p3]")
                             ".Lsm_b3_%=:\n\t" SM_STEP(3, "%[p2]")
                             ".Lsm_b2_%=:\n\t" SM_STEP(2, "%[p1]")
                             ".Lsm_b1_%=:\n\t" SM_STEP(1, "%[p0]")
                             ".Lsm_b0_%=:\n\t" SM_STEP(0, "0")
                             "s_mov_b64 exec, %[sv]"
                             : [v] "=&v"(v), [acc] "=&v"(acc), [sv] "=&s"(sv), [tm] "=&s"(tm)
                             : [t] "v"(target), [ng] "s"(ng), [p0] "v"(P0[c]), [p1] "v"(P1[c]), [p2] "v"(P2[c]), [p3] "v"(P3[c]), [p4] "v"(P4[c]),
                               [p5] "v"(P5[c]), [p6] "v"(P6[c]), [p7] "v"(P7[c]), [g0] "v"(bf.g0), [g1] "v"(bf.g1), [g2] "v"(bf.g2), [g3] "v"(bf.g3),
                               [g4] "v"(bf.g4), [g5] "v"(bf.g5), [g6] "v"(bf.g6), [g7] "v"(bf.g7)
                             : "scc");
#undef SM_STEP
            }
            const bool hit = v != 0u; // a stored group word is never 0
            const double *m = s_mu + c * CS;
            uint32_t sel; // LDS byte offset of the selected window entry
            {
                const uint32_t o0 = SM_OFF0(v), o1 = SM_OFF1(v), o2 = SM_OFF2(v), o3 = SM_OFF3(v);
                double w0 = *(const double *)((const char *)m + o0), w1 = *(const double *)((const char *)m + o1), w2 = *(const double *)((const char *)m + o2);
                asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2)); // all three requested before the first is waited for
                const double p0 = acc + w0, p1 = p0 + w1, p2 = p1 + w2;
                sel = o3;
                uint64_t sv, tm;
                asm volatile("s_mov_b64 %[sv], exec\n\t"
                             "v_cmpx_lt_f64_e64 %[tm], %[t], %[p2]\n\t" "v_mov_b32 %[sel], %[o2]\n\t"
                             "v_cmpx_lt_f64_e64 %[tm], %[t], %[p1]\n\t" "v_mov_b32 %[sel], %[o1]\n\t"
                             "v_cmpx_lt_f64_e64 %[tm], %[t], %[p0]\n\t" "v_mov_b32 %[sel], %[o0]\n\t"
                             "s_mov_b64 exec, %[sv]"
                             : [sel] "+&v"(sel), [sv] "=&s"(sv), [tm] "=&s"(tm)
                             : [t] "v"(target), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [o0] "v"(o0), [o1] "v"(o1), [o2] "v"(o2));
            }
            if (!hit) { // rare: an empty row, a degenerate total, a row of more than 32 hits, rounding
                const uint32_t L = row_len();
                if (L == 0) sel = (uint32_t)WIN << SH; // no row in this lane: the count of the pad slot, which is never flushed
                else if (degenerate) {
                    const uint32_t j = (uint32_t)(u32_unit(x) * (double)L);
                    sel = off_of(j < L ? j : L - 1);
                } else {
                    double accl = P7[c];
                    bool found = false;
#pragma unroll 1
                    for (uint32_t g = 8; g < ng && !found; ++g) {
                        const uint32_t vv = src[(size_t)g * 64];
                        const uint32_t o0 = SM_OFF0(vv), o1 = SM_OFF1(vv), o2 = SM_OFF2(vv), o3 = SM_OFF3(vv);
                        const double p0 = accl + *(const double *)((const char *)m + o0), p1 = p0 + *(const double *)((const char *)m + o1),
                                     p2 = p1 + *(const double *)((const char *)m + o2), p3 = p2 + *(const double *)((const char *)m + o3);
                        if (target < p3) { sel = target < p0 ? o0 : (target < p1 ? o1 : (target < p2 ? o2 : o3)); found = true; }
                        accl = p3;
                    }
                    if (!found) sel = off_of(L - 1); // rounding left target >= total: the last real hit
                }
            }
            atomicAdd((int32_t *)((char *)s_cnt + c * (CS * 8) + sel), 1);
        }
    };
#undef SM_GROUPS

    // rows of a slow tile: straight from the 32-bit CSR, chain by chain
    struct RowViewMulti {
        const uint32_t *cl;
        uint32_t L, wbase, n;
        const double *s_mu_c; // the chain's window
        const double *gmu_c;  // gmu + chain * n
        __device__ __forceinline__ uint32_t col(uint32_t j) const { return cl[j]; }
        __device__ __forceinline__ double w(uint32_t j) const
        {
            const uint32_t col = cl[j], dd = col - wbase;
            return dd < (uint32_t)SELL_WIN ? s_mu_c[dd] : gmu_c[col];
        }
        __device__ __forceinline__ double total() const
        {
            double tt = 0.0;
            for (uint32_t j = 0; j < L; ++j) tt += w(j);
            return tt;
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
    auto slow_tile = [&](const SellTile &d) {
        const uint32_t wbase = d.wbase;
        if (lane < d.nrows()) {
            const uint64_t st = (uint64_t)row_ptr[d.r0 + lane];
            const uint32_t L = (uint32_t)((uint64_t)row_ptr[d.r0 + lane + 1] - st);
#pragma unroll 1
            for (int c = 0; c < NCH; ++c) {
                int32_t *gc = gcnt + (size_t)c * a.n;
                auto add = [&](uint32_t col, int32_t x) {
                    const uint32_t dd = col - wbase;
                    if (dd < (uint32_t)WIN) atomicAdd(&s_cnt[2 * (c * CS + dd)], x);
                    else global_count_add(gc, col, x);
                };
                RowViewMulti v{col_idx + st, L, wbase, a.n, s_mu + c * CS, gmu + (size_t)c * a.n};
                SampleArgs ac = a;
                ac.chain = a.chain + (uint32_t)c;
                allocate_row<false>(v, add, 1u, ac, a.row_id_base + d.r0 + lane);
            }
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
        if (d.flags() & SELL_FAST) walk(d, bf, which);
        else slow_tile(d);
        issue(refill, bf);
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
    // tile A's block is requested BEFORE tile B's, here as in the loop: the waits are by count, and with the two requests swapped (the
    // scheduler is free to) the count valid on entry is 0 for tile A -- which the loop header then inherits for every iteration: the
    // walk of every tile A would wait for the block of tile B requested just before it
    __builtin_amdgcn_sched_barrier(0);
    issue(dB, bufB);
    for (uint32_t i = 0; i < nt; i += 2) {
        const SellTile nA = tile_at(i + 2), nB = tile_at(i + 3);
        if constexpr (NCH == 2) {
            // ONE path for every pair of tiles (and one place per buffer where the next block is requested: the registers of a buffer
            // then stay where they are from iteration to iteration).  A tile's kind: 0 = none (past the end of the range), 1..8 = a
            // register-path tile of that many groups (straight-line walk), 9 = anything else (generic walk: same bits).
            auto kind = [&](const SellTile &d) -> uint32_t { // uniform
                if (d.flags() & SELL_EMPTY) return 0u;
                return ((d.flags() & SELL_FAST) && d.ng() - 1u < 8u) ? d.ng() : 9u;
            };
            auto window_for = [&](const SellTile &d) -> bool {
                if (d.wbase == cur_base) return false;
                __syncthreads();
                flush_window(cur_base);
                load_window(d.wbase);
                cur_base = d.wbase;
                __syncthreads();
                return true;
            };
            const uint32_t kA = kind(dA), kB = kind(dB);
            if (kA) window_for(dA);
            const G4 hA = gather(bufA.g0); // tile A's first gathers travel while the pair's Philox blocks are computed
            pair_rng(dA, dB);
            // the descriptors' scalar loads are waited for HERE, behind the Philox rounds: scalar loads return out of order, so with
            // one of them outstanding the walk's first wait for its gathers would be a wait for everything (lgkmcnt(0)) -- including
            // the gathers of the group it has just requested ahead
            asm volatile("" ::"s"(nA.off16), "s"(nA.r0), "s"(nA.wbase), "s"(nA.meta), "s"(nB.off16), "s"(nB.r0), "s"(nB.wbase), "s"(nB.meta));
            // tile B's first gathers go out before tile A's picks -- except behind a tile of 7 or 8 groups (one in seven at 20 hits per
            // read), whose 28 / 32 boundary registers leave no room for the 16 of the gathers at 4 waves per SIMD
            G4 hB;
            auto early = [&]() { hB = gather(bufB.g0); };
            auto nothing = []() {};
            switch (kA) {
            case 0: early(); break;
            case 1: walk_fixed(dA, bufA, 0, IntTag<1>(), hA, early); break;
            case 2: walk_fixed(dA, bufA, 0, IntTag<2>(), hA, early); break;
            case 3: walk_fixed(dA, bufA, 0, IntTag<3>(), hA, early); break;
            case 4: walk_fixed(dA, bufA, 0, IntTag<4>(), hA, early); break;
            case 5: walk_fixed(dA, bufA, 0, IntTag<5>(), hA, early); break;
            case 6: walk_fixed(dA, bufA, 0, IntTag<6>(), hA, early); break;
            case 7: walk_fixed(dA, bufA, 0, IntTag<7>(), hA, nothing); early(); break;
            case 8: walk_fixed(dA, bufA, 0, IntTag<8>(), hA, nothing); early(); break;
            default:
                if (dA.flags() & SELL_FAST) walk(dA, bufA, 0);
                else slow_tile(dA);
                early();
                break;
            }
            issue(nA, bufA);
            if (kB && window_for(dB)) hB = gather(bufB.g0); // (the early gathers read the window before)
            switch (kB) {
            case 0: break;
            case 1: walk_fixed(dB, bufB, 1, IntTag<1>(), hB, nothing); break;
            case 2: walk_fixed(dB, bufB, 1, IntTag<2>(), hB, nothing); break;
            case 3: walk_fixed(dB, bufB, 1, IntTag<3>(), hB, nothing); break;
            case 4: walk_fixed(dB, bufB, 1, IntTag<4>(), hB, nothing); break;
            case 5: walk_fixed(dB, bufB, 1, IntTag<5>(), hB, nothing); break;
            case 6: walk_fixed(dB, bufB, 1, IntTag<6>(), hB, nothing); break;
            case 7: walk_fixed(dB, bufB, 1, IntTag<7>(), hB, nothing); break;
            case 8: walk_fixed(dB, bufB, 1, IntTag<8>(), hB, nothing); break;
            default:
                if (dB.flags() & SELL_FAST) walk(dB, bufB, 1);
                else slow_tile(dB);
                break;
            }
            issue(nB, bufB);
        } else {
            pair_rng(dA, dB);
            process(dA, cur_base, nA, bufA, 0);
            process(dB, cur_base, nB, bufB, 1);
        }
        dA = nA;
        dB = nB;
    }
    __syncthreads();
    flush_window(cur_base);
#undef SM_OFF0
#undef SM_OFF1
#undef SM_OFF2
#undef SM_OFF3
}

} // namespace mmg
