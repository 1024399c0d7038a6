// Issue-rate microbenchmark for gfx950 (MI355X): how many shader cycles does one SIMD need per wave64 instruction of a given kind,
// with 1..8 waves resident on it?  The denominator of every "VALU busy" figure in DESIGN.md / bench.py.
//
// Method: every wave runs REPS x an unrolled block of 64 INDEPENDENT instructions of one kind (8 accumulators round-robin, written in
// inline asm so nothing is folded) between two s_memtime reads, and records WHERE it ran: HW_ID (SIMD, CU, SH, SE) and XCC_ID.
// Workgroups are 256 threads = four waves; grid = 256 CUs x k.  Round 3 assumed that this puts k waves on every SIMD and divided a
// wave's own elapsed ticks by k -- the dispatcher does not promise that, and the two columns of that table disagreed by up to 5 x.
// Now the host groups the waves by the SIMD they ran on and computes, per SIMD,
//     cycles per instruction = (last t1 - first t0 of its waves) / (instructions its waves executed)
//     waves resident         = sum of the waves' own (t1 - t0) / that interval              (printed in brackets)
// and reports the median over the SIMDs.  The tick rate of s_memtime is measured against the HIP-event wall time of the k = 1 launch
// (a wave's own ticks / wall): printed as "GHz" -- it is the shader clock if s_memtime counts shader cycles
// (MI355X_MICROARCH.md), and then "cycles" above are real ones whatever the clock did under this load.
// "mix" rows interleave two kinds to see whether they share an issue port.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

enum Kind {
    ADD_U32, XOR_B32, CNDMASK, BFE_U32, LSHL_ADD, MOV_B32, ADD_F32, FMA_F32, PK_FMA_F32, ADD_F64, FMA_F64, MUL_F64, CMP_F64, CMP_U32,
    MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, MUL_U24, CVT_F64_U32, S_ADD, MIX_VALU_SALU, MIX_ADD_F64_U32, DS_READ_B64, DS_READ_B128,
    DS_ADD_U32, MIX_DSREAD_VALU, SDWA_BYTE, XOR3, CNDMASK_SGPR, CNDMASK_VCC_SET, MIX_CMP_CNDMASK, CMP_F64_SGPR, ADD_F64_DEP, ADD_U32_DEP, DS_READ_B64_BCAST, DS_READ_B64_LINEAR, DS_READ_B128_BCAST, DS_READ_B32_RANDOM, DS_BPERMUTE, MAD_U64_SCONST, MIX_F64_CNDMASK, PICK_VCC, PICK_SGPR, PICK_VCC_1, PICK_VCC_SPACED, N_KINDS
};
static const char *kind_name[N_KINDS] = {
    "v_add_u32", "v_xor_b32", "v_cndmask_b32", "v_bfe_u32", "v_lshl_add_u32", "v_mov_b32", "v_add_f32", "v_fma_f32", "v_pk_fma_f32",
    "v_add_f64", "v_fma_f64", "v_mul_f64", "v_cmp_lt_f64", "v_cmp_lt_u32", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24",
    "v_cvt_f64_u32", "s_add_u32", "mix: v_add_u32 + s_add_u32 (1:1)", "mix: v_add_f64 + v_add_u32 (1:1)", "ds_read_b64 (random 8-B slots of a 2 KB window)",
    "ds_read_b128 (random 16-B slots of a 4 KB window)", "ds_add_u32 (random slots of a 1 KB window)", "mix: ds_read_b64 + 4 v_add_u32", "v_lshlrev_b32_sdwa (byte select)", "v_and_or_b32",
    "v_cndmask_b32 e64 (mask in an SGPR pair)", "v_cndmask_b32 vcc (vcc set by s_mov before the loop)", "mix: v_cmp_lt_f64 vcc + v_cndmask_b32 vcc (1:1)", "v_cmp_lt_f64 e64 -> SGPR pair",
    "v_add_f64 DEPENDENT chain (latency)", "v_add_u32 DEPENDENT chain (latency)", "ds_read_b64 (all lanes one address)", "ds_read_b64 (lane i -> slot i, conflict-free)", "ds_read_b128 (all lanes one address)",
    "ds_read_b32 (random 4-B slots of a 1 KB window)", "ds_bpermute_b32", "v_mad_u64_u32 (SGPR constant multiplier, +0)", "mix: v_add_f64 + v_cndmask_b32 e64 (1:1)",
    "K1 pick step: v_cmp_lt_f64 vcc + 3 v_cndmask_b32 e32 vcc   (per instruction)", "same with v_cmp_lt_f64 e64 -> SGPR pair + 3 v_cndmask_b32 e64", "v_cmp_lt_f64 vcc + 1 v_cndmask e32 vcc + 2 v_add_u32", "v_cmp_lt_f64 vcc + v_add_u32 + 3 v_cndmask e32 vcc (5 per step)"
};
// instructions counted per unrolled block (the mixes count BOTH kinds)
static int kind_block(int k) { return (k == MIX_VALU_SALU || k == MIX_ADD_F64_U32 || k == MIX_CMP_CNDMASK || k == MIX_F64_CNDMASK) ? 128 : (k == PICK_VCC || k == PICK_SGPR || k == PICK_VCC_1) ? 256 : (k == PICK_VCC_SPACED) ? 320 : (k == MIX_DSREAD_VALU ? 80 : 64); }

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define R64(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)

template <int K>
__global__ __launch_bounds__(256) void k_issue(uint64_t *out, int reps, uint32_t seed)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[4096];
    const uint32_t tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) lds[i] = i * 2654435761u + seed;
    __syncthreads();
    uint32_t a0 = tid + seed, a1 = a0 * 3u + 1u, a2 = a0 * 5u + 2u, a3 = a0 * 7u + 3u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
    double d0 = 1.0 + tid, d1 = 2.0 + tid, d2 = 3.0 + tid, d3 = 4.0 + tid, d4 = 5.0 + tid, d5 = 6.0 + tid, d6 = 7.0 + tid, d7 = 8.0 + tid;
    float f0 = 1.f + tid, f1 = 2.f, f2 = 3.f, f3 = 4.f, f4 = 5.f, f5 = 6.f, f6 = 7.f, f7 = 8.f;
    uint64_t q0 = tid, q1 = 1, q2 = 2, q3 = 3, q4 = 4, q5 = 5, q6 = 6, q7 = 7;
    const uint32_t b = (tid * 2654435761u + seed) | 1u;
    const double e = 1.0000001;
    const float g = 1.0001f;
    uint32_t s0 = seed, s1 = 1, s2 = 2, s3 = 3;
    // random but in-range LDS byte addresses (fixed per lane: the conflict pattern of K1's gathers, 8-B slots of a 255-entry window)
    const uint32_t ad64 = ((tid * 2654435761u + seed) >> 7) % 255u * 8u, ad128 = ((tid * 40503u + seed * 7u) >> 3) % 255u * 16u,
                   ad32 = ((tid * 2654435761u + seed) >> 9) % 255u * 4u;
    const uint32_t adb = (seed & 127u) * 16u, adl = (tid & 63u) * 8u, adp = ((tid * 2654435761u + seed) >> 11) % 64u * 4u;
    uint64_t m64 = 0x5555aaaa3333ccccull ^ seed;
    uint32_t c0 = tid, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7, h0 = tid, h1 = 1, h2 = 2, h3 = 3, h4 = 4, h5 = 5, h6 = 6, h7 = 7;
    if (K == CNDMASK_VCC_SET) asm volatile("s_mov_b64 vcc, %0" : : "s"(m64) : "vcc");
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 p0 = {1.f, 2.f}, p1 = p0, p2 = p0, p3 = p0, p4 = p0, p5 = p0, p6 = p0, p7 = p0;
    const f32x2 pg = {g, g};
    f64x2 w0 = {0, 0}, w1 = w0, w2 = w0, w3 = w0;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
        if (K == ADD_U32) {
#define I(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
            R64(I)
#undef I
        } else if (K == XOR_B32) {
#define I(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a##i) : "v"(b));
            R64(I)
#undef I
        } else if (K == XOR3) {
#define I(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "s"(s0));
            R64(I)
#undef I
        } else if (K == CNDMASK) {
#define I(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b) : );
            R64(I)
#undef I
        } else if (K == BFE_U32) {
#define I(i) asm volatile("v_bfe_u32 %0, %0, 3, 11" : "+v"(a##i));
            R64(I)
#undef I
        } else if (K == SDWA_BYTE) {
#define I(i) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(a##i) : "v"(3));
            R64(I)
#undef I
        } else if (K == LSHL_ADD) {
#define I(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a##i) : "v"(b));
            R64(I)
#undef I
        } else if (K == MOV_B32) {
#define I(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a##i) : "v"(b));
            R64(I)
#undef I
        } else if (K == ADD_F32) {
#define I(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f##i) : "v"(g));
            R64(I)
#undef I
        } else if (K == FMA_F32) {
#define I(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f##i) : "v"(g));
            R64(I)
#undef I
        } else if (K == PK_FMA_F32) {
#define I(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p##i) : "v"(pg));
            R64(I)
#undef I
        } else if (K == ADD_F64) {
#define I(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d##i) : "v"(e));
            R64(I)
#undef I
        } else if (K == FMA_F64) {
#define I(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d##i) : "v"(e));
            R64(I)
#undef I
        } else if (K == MUL_F64) {
#define I(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d##i) : "v"(e));
            R64(I)
#undef I
        } else if (K == CMP_F64) {
#define I(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d##i), "v"(e) : "vcc");
            R64(I)
#undef I
        } else if (K == CMP_U32) {
#define I(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a##i), "v"(b) : "vcc");
            R64(I)
#undef I
        } else if (K == MAD_U64_U32) {
#define I(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q##i) : "v"(b), "v"(a##i) : "vcc");
            R64(I)
#undef I
        } else if (K == MUL_LO_U32) {
#define I(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
            R64(I)
#undef I
        } else if (K == MUL_HI_U32) {
#define I(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
            R64(I)
#undef I
        } else if (K == MUL_U24) {
#define I(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a##i) : "v"(b));
            R64(I)
#undef I
        } else if (K == CVT_F64_U32) {
#define I(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d##i) : "v"(a##i));
            R64(I)
#undef I
        } else if (K == S_ADD) {
#define I(i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc"); asm volatile("s_add_u32 %0, %0, %1" : "+s"(s2) : "s"(s3) : "scc");
            R8(I) R8(I) R8(I) R8(I)
#undef I
        } else if (K == MIX_VALU_SALU) {
#define I(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b)); asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
            R64(I)
#undef I
        } else if (K == MIX_ADD_F64_U32) {
#define I(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d##i) : "v"(e)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
            R64(I)
#undef I
        } else if (K == DS_READ_B64) {
#define I(i) asm volatile("ds_read_b64 %0, %1" : "=v"(q##i) : "v"(ad64) : "memory");
            R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef I
        } else if (K == DS_READ_B128) {
#define I(i) asm volatile("ds_read_b128 %0, %1" : "=v"(w0) : "v"(ad128) : "memory"); asm volatile("ds_read_b128 %0, %1" : "=v"(w1) : "v"(ad128) : "memory");
            R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef I
        } else if (K == DS_ADD_U32) {
#define I(i) asm volatile("ds_add_u32 %0, %1" : : "v"(ad32), "v"(b) : "memory");
            R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef I

        } else if (K == CNDMASK_SGPR) {
#define I(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "s"(m64));
            R64(I)
#undef I
        } else if (K == CNDMASK_VCC_SET) {
#define I(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b) : );
            R64(I)
#undef I
        } else if (K == MIX_CMP_CNDMASK) {
#define I(i) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(a##i) : "v"(d##i), "v"(e), "v"(b) : "vcc");
            R64(I)
#undef I
        } else if (K == CMP_F64_SGPR) {
#define I(i) asm volatile("v_cmp_lt_f64_e64 %0, %1, %2" : "=s"(m64) : "v"(d##i), "v"(e));
            R64(I)
#undef I
        } else if (K == ADD_F64_DEP) {
#define I(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(e));
            R64(I)
#undef I
        } else if (K == ADD_U32_DEP) {
#define I(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(b));
            R64(I)
#undef I
        } else if (K == DS_READ_B64_BCAST) {
#define I(i) asm volatile("ds_read_b64 %0, %1" : "=v"(q##i) : "v"(adb) : "memory");
            R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef I
        } else if (K == DS_READ_B64_LINEAR) {
#define I(i) asm volatile("ds_read_b64 %0, %1" : "=v"(q##i) : "v"(adl) : "memory");
            R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef I
        } else if (K == DS_READ_B128_BCAST) {
#define I(i) asm volatile("ds_read_b128 %0, %1" : "=v"(w0) : "v"(adb) : "memory"); asm volatile("ds_read_b128 %0, %1" : "=v"(w1) : "v"(adb) : "memory");
            R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef I
        } else if (K == DS_READ_B32_RANDOM) {
#define I(i) asm volatile("ds_read_b32 %0, %1" : "=v"(a##i) : "v"(ad32) : "memory");
            R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef I
        } else if (K == DS_BPERMUTE) {
#define I(i) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(a##i) : "v"(adp), "v"(b) : "memory");
            R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef I
        } else if (K == MAD_U64_SCONST) {
#define I(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q##i) : "v"(a##i), "s"(0xD256D193u) : "vcc");
            R64(I)
#undef I
        } else if (K == MIX_F64_CNDMASK) {
#define I(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d##i) : "v"(e)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "s"(m64));
            R64(I)
#undef I

        } else if (K == PICK_VCC) {
#define I(i) asm volatile("v_cmp_lt_f64 vcc, %3, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %5, vcc" : "+v"(a##i), "+v"(c##i), "+v"(h##i) : "v"(d##i), "v"(e), "v"(b) : "vcc");
            R64(I)
#undef I
        } else if (K == PICK_SGPR) {
#define I(i) asm volatile("v_cmp_lt_f64_e64 %6, %3, %4\n v_cndmask_b32_e64 %0, %0, %5, %6\n v_cndmask_b32_e64 %1, %1, %5, %6\n v_cndmask_b32_e64 %2, %2, %5, %6" : "+v"(a##i), "+v"(c##i), "+v"(h##i) : "v"(d##i), "v"(e), "v"(b), "s"(m64));
            R64(I)
#undef I
        } else if (K == PICK_VCC_1) {
#define I(i) asm volatile("v_cmp_lt_f64 vcc, %3, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %5" : "+v"(a##i), "+v"(c##i), "+v"(h##i) : "v"(d##i), "v"(e), "v"(b) : "vcc");
            R64(I)
#undef I
        } else if (K == PICK_VCC_SPACED) {
#define I(i) asm volatile("v_cmp_lt_f64 vcc, %3, %4\n v_add_u32 %0, %0, %5\n v_cndmask_b32 %0, %0, %5, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %5, vcc" : "+v"(a##i), "+v"(c##i), "+v"(h##i) : "v"(d##i), "v"(e), "v"(b) : "vcc");
            R64(I)
#undef I
        } else if (K == MIX_DSREAD_VALU) { // 16 gathers + 64 adds per block, the adds independent of the gathers
#define I(i) asm volatile("ds_read_b64 %0, %1" : "=v"(q##i) : "v"(ad64) : "memory"); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b)); \
             asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
            R8(I) R8(I) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef I
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    // keep every accumulator alive
    uint64_t sink = (uint64_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) + (uint64_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + (uint64_t)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) +
                    (q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) + s0 + s2 + m64 + (c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7) + (h0 ^ h1 ^ h2 ^ h3 ^ h4 ^ h5 ^ h6 ^ h7) + (uint64_t)(p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y) + (uint64_t)(w0.x + w1.y + w2.x + w3.y);
    if (sink == 0x123456789abcdefull) out[0] = sink;
    if ((tid & 63u) == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        uint64_t *o = out + 1 + ((size_t)blockIdx.x * 4 + (tid >> 6)) * 3;
        o[0] = t0; o[1] = t1; o[2] = ((uint64_t)(xcc & 0xfu) << 32) | hw;
    }
}

template <int K>
static void run_kind(uint64_t *d_out, int reps)
{
    printf("%-52s", kind_name[K]);
    double ghz_sum = 0.0;
    for (int k : {1, 2, 4, 8}) {
        const int grid = 256 * k;
        const size_t words = 1 + (size_t)grid * 4 * 3;
        CK(hipMemset(d_out, 0, words * 8));
        hipLaunchKernelGGL(k_issue<K>, dim3(grid), dim3(256), 0, 0, d_out, 4, 1u); // warm-up
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_issue<K>, dim3(grid), dim3(256), 0, 0, d_out, reps, 2u);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint64_t> h(words);
        CK(hipMemcpy(h.data(), d_out, words * 8, hipMemcpyDeviceToHost));
        // waves by the SIMD they ran on: XCC, then SE / SH / CU (HW_ID bits 8-15) and SIMD (bits 4-5)
        struct W { uint64_t key, t0, t1; };
        std::vector<W> w((size_t)grid * 4);
        for (size_t i = 0; i < w.size(); ++i) { const uint64_t *o = &h[1 + i * 3]; w[i] = {((o[2] >> 32) << 16) | (o[2] & 0xff30u), o[0], o[1]}; }
        std::sort(w.begin(), w.end(), [](const W &a, const W &b) { return a.key < b.key; });
        const double n_inst = (double)reps * kind_block(K);
        std::vector<double> cyc, res;
        for (size_t i = 0; i < w.size();) {
            size_t j = i;
            uint64_t lo = ~0ull, hi = 0, own = 0;
            while (j < w.size() && w[j].key == w[i].key) { lo = std::min(lo, w[j].t0); hi = std::max(hi, w[j].t1); own += w[j].t1 - w[j].t0; ++j; }
            cyc.push_back((double)(hi - lo) / ((double)(j - i) * n_inst));
            res.push_back((double)own / (double)(hi - lo));
            i = j;
        }
        std::sort(cyc.begin(), cyc.end());
        std::sort(res.begin(), res.end());
        // the clock the ticks count: one wave per SIMD (k = 1) runs from launch to end -- its own ticks over the wall time of the launch
        // (the few microseconds of launch overhead inside the wall time make this a slight underestimate)
        if (k == 1) {
            std::vector<uint64_t> own(w.size());
            for (size_t i = 0; i < w.size(); ++i) own[i] = w[i].t1 - w[i].t0;
            std::sort(own.begin(), own.end());
            ghz_sum = (double)own[own.size() / 2] / ((double)ms * 1e6);
        }
        printf("  k=%d %5.2f [%3.1f on %4zu]", k, cyc[cyc.size() / 2], res[res.size() / 2], cyc.size());
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
    printf("  %.2f GHz\n", ghz_sum);
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 2000;
    uint64_t *d_out;
    CK(hipMalloc(&d_out, (1 + 256 * 8 * 4 * 3) * 8));
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    printf("# %s, %d CUs, clockRate %d kHz; reps %d x 64-instruction blocks per wave; k = waves per SIMD\n", pr.gcnArchName, pr.multiProcessorCount, pr.clockRate, reps);
    printf("# per k (workgroups per CU): s_memtime ticks per instruction per SIMD, median over the SIMDs [waves resident on a SIMD, SIMDs that ran waves];\n");
    printf("# last column: a wave's own s_memtime ticks per nanosecond of HIP-event wall time in the k = 1 launch (the clock the ticks count)\n");
    run_kind<ADD_U32>(d_out, reps);
    run_kind<XOR_B32>(d_out, reps);
    run_kind<XOR3>(d_out, reps);
    run_kind<CNDMASK>(d_out, reps);
    run_kind<BFE_U32>(d_out, reps);
    run_kind<SDWA_BYTE>(d_out, reps);
    run_kind<LSHL_ADD>(d_out, reps);
    run_kind<MOV_B32>(d_out, reps);
    run_kind<ADD_F32>(d_out, reps);
    run_kind<FMA_F32>(d_out, reps);
    run_kind<PK_FMA_F32>(d_out, reps);
    run_kind<ADD_F64>(d_out, reps);
    run_kind<FMA_F64>(d_out, reps);
    run_kind<MUL_F64>(d_out, reps);
    run_kind<CMP_F64>(d_out, reps);
    run_kind<CMP_U32>(d_out, reps);
    run_kind<CVT_F64_U32>(d_out, reps);
    run_kind<MAD_U64_U32>(d_out, reps);
    run_kind<MUL_LO_U32>(d_out, reps);
    run_kind<MUL_HI_U32>(d_out, reps);
    run_kind<MUL_U24>(d_out, reps);
    run_kind<S_ADD>(d_out, reps);
    run_kind<MIX_VALU_SALU>(d_out, reps);
    run_kind<MIX_ADD_F64_U32>(d_out, reps);
    run_kind<DS_READ_B64>(d_out, reps);
    run_kind<DS_READ_B128>(d_out, reps);
    run_kind<DS_ADD_U32>(d_out, reps);
    run_kind<MIX_DSREAD_VALU>(d_out, reps);
    run_kind<PICK_VCC>(d_out, reps);
    run_kind<PICK_SGPR>(d_out, reps);
    run_kind<PICK_VCC_1>(d_out, reps);
    run_kind<PICK_VCC_SPACED>(d_out, reps);
    run_kind<CNDMASK_SGPR>(d_out, reps);
    run_kind<CNDMASK_VCC_SET>(d_out, reps);
    run_kind<MIX_CMP_CNDMASK>(d_out, reps);
    run_kind<CMP_F64_SGPR>(d_out, reps);
    run_kind<MIX_F64_CNDMASK>(d_out, reps);
    run_kind<ADD_F64_DEP>(d_out, reps);
    run_kind<ADD_U32_DEP>(d_out, reps);
    run_kind<MAD_U64_SCONST>(d_out, reps);
    run_kind<DS_READ_B64_BCAST>(d_out, reps);
    run_kind<DS_READ_B64_LINEAR>(d_out, reps);
    run_kind<DS_READ_B128_BCAST>(d_out, reps);
    run_kind<DS_READ_B32_RANDOM>(d_out, reps);
    run_kind<DS_BPERMUTE>(d_out, reps);
    return 0;
}
