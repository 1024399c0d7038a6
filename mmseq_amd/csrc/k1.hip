// k1.hip -- device TU: the sample kernels (src/mmseq.cpp:857-891) and the builders of their streams
#include "gibbs_kernels.h"
#include "sell_kernels.h"
#include "sell_multi_kernels.h"
#include "bigk_kernels.h"
#include "mmg_launch.h"

namespace mmg {

const void *k1_sell_kernel(bool idx64, bool has_k, bool fixed_walk)
{
    if (!has_k && fixed_walk) return idx64 ? (const void *)k_sample_sell<uint64_t, false, 8, 1, false, true> : (const void *)k_sample_sell<uint32_t, false, 8, 1, false, true>;
    if (idx64) return has_k ? (const void *)k_sample_sell<uint64_t, true, 8> : (const void *)k_sample_sell<uint64_t, false, 8>;
    return has_k ? (const void *)k_sample_sell<uint32_t, true, 8> : (const void *)k_sample_sell<uint32_t, false, 8>;
}

const void *k1_sell_far_kernel(bool idx64)
{
    return idx64 ? (const void *)k_sample_sell<uint64_t, false, 8, 1, true> : (const void *)k_sample_sell<uint32_t, false, 8, 1, true>;
}

const void *k1_sell_multi_kernel(bool idx64, int nch)
{
    if (nch == 2) return idx64 ? (const void *)k_sample_sell_multi<uint64_t, 2> : (const void *)k_sample_sell_multi<uint32_t, 2>;
    if (nch == 4) return idx64 ? (const void *)k_sample_sell_multi<uint64_t, 4> : (const void *)k_sample_sell_multi<uint32_t, 4>;
    return nullptr;
}

const void *k1_bigk_kernel(bool idx64) { return idx64 ? (const void *)k_sample_bigk<uint64_t> : (const void *)k_sample_bigk<uint32_t>; }

#if defined(MMG_BIGK_STATS)
extern "C" int mmg_selftest_bigk_stats(unsigned long long *out) // reads and clears the counters (diagnostics build only)
{
    unsigned long long zero[16] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bigk_stats), sizeof(zero)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_bigk_stats), zero, sizeof(zero)) == hipSuccess ? 0 : 1;
}
#endif

const void *k1_csr_kernel(bool idx64, bool has_k)
{
    if (idx64) return has_k ? (const void *)k_sample<uint64_t, true, K1C_ELEMS, K1C_WIN, K1C_UNR, K1C_BS, K1C_ROWS>
                            : (const void *)k_sample<uint64_t, false, K1C_ELEMS, K1C_WIN, K1C_UNR, K1C_BS, K1C_ROWS>;
    return has_k ? (const void *)k_sample<uint32_t, true, K1C_ELEMS, K1C_WIN, K1C_UNR, K1C_BS, K1C_ROWS>
                 : (const void *)k_sample<uint32_t, false, K1C_ELEMS, K1C_WIN, K1C_UNR, K1C_BS, K1C_ROWS>;
}

void launch_tile_desc(bool idx64, const void *row_ptr, const uint32_t *col, const uint32_t *kmult, const uint64_t *tile_row, uint64_t n_tiles,
                      TileDesc *out, hipStream_t s)
{
    if (!n_tiles) return;
    if (idx64) hipLaunchKernelGGL(k_tile_desc<uint64_t>, dim3((unsigned)n_tiles), dim3(64), 0, s, (const uint64_t *)row_ptr, col, kmult, tile_row, n_tiles, out);
    else hipLaunchKernelGGL(k_tile_desc<uint32_t>, dim3((unsigned)n_tiles), dim3(64), 0, s, (const uint32_t *)row_ptr, col, kmult, tile_row, n_tiles, out);
}

void launch_tile_far(bool idx64, const void *row_ptr, const uint32_t *col, const uint64_t *key, const uint64_t *tile_row,
                     const uint32_t *cand, uint64_t n_cand, uint32_t *out /* [3][n_cand] */, hipStream_t s)
{
    if (!n_cand) return;
    if (idx64) hipLaunchKernelGGL(k_tile_far<uint64_t>, dim3((unsigned)n_cand), dim3(64), 0, s, (const uint64_t *)row_ptr, col, key, tile_row, cand, n_cand, out);
    else hipLaunchKernelGGL(k_tile_far<uint32_t>, dim3((unsigned)n_cand), dim3(64), 0, s, (const uint32_t *)row_ptr, col, key, tile_row, cand, n_cand, out);
}

void launch_encode_sell(bool idx64, const void *row_ptr, const uint32_t *col, const SellTile *tiles, uint64_t n_tiles,
                        uint8_t *stream, hipStream_t s)
{
    if (!n_tiles) return;
    if (idx64) hipLaunchKernelGGL(k_encode_sell<uint64_t>, dim3((unsigned)n_tiles), dim3(64), 0, s, (const uint64_t *)row_ptr, col, tiles, n_tiles, stream);
    else hipLaunchKernelGGL(k_encode_sell<uint32_t>, dim3((unsigned)n_tiles), dim3(64), 0, s, (const uint32_t *)row_ptr, col, tiles, n_tiles, stream);
}

} // namespace mmg
