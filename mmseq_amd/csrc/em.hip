// em.hip -- device TU: EM sweeps (src/mmseq.cpp:741-811), kernels in em_kernels.h
#include "gibbs_kernels.h"
#include "sell_kernels.h"
#include "em_kernels.h"
#include "mmg_launch.h"

namespace mmg {

// k_em_sell runs as 2 waves per workgroup sharing one window, 4 accumulator replicas (measured at cfg 3, round 3: 1.31 ms per
// sweep; 4 waves x 8 replicas 1.31, 4 x 4 1.38, 2 x 2 1.45, 2 x 8 1.67)
const void *em_sell_kernel(bool idx64, bool has_k, bool measure)
{
#define EMS_PICK(IDX, HK) (measure ? (const void *)k_em_sell<IDX, HK, true, 1, EM_SELL_W> : (const void *)k_em_sell<IDX, HK, false, EM_SELL_REP, EM_SELL_W>)
    if (idx64) return has_k ? EMS_PICK(uint64_t, true) : EMS_PICK(uint64_t, false);
    return has_k ? EMS_PICK(uint32_t, true) : EMS_PICK(uint32_t, false);
#undef EMS_PICK
}

void launch_em_rows_global(bool idx64, bool measure, const void *row_ptr, const uint32_t *col, const uint32_t *k, uint64_t m,
                           EmArgs a, hipStream_t s)
{
    if (!m) return;
    const unsigned gr = (unsigned)((m + 255) / 256);
    if (idx64) {
        if (measure) hipLaunchKernelGGL((k_em_rows_global<uint64_t, true>), dim3(gr), dim3(256), 0, s, (const uint64_t *)row_ptr, col, k, m, a);
        else hipLaunchKernelGGL((k_em_rows_global<uint64_t, false>), dim3(gr), dim3(256), 0, s, (const uint64_t *)row_ptr, col, k, m, a);
    } else {
        if (measure) hipLaunchKernelGGL((k_em_rows_global<uint32_t, true>), dim3(gr), dim3(256), 0, s, (const uint32_t *)row_ptr, col, k, m, a);
        else hipLaunchKernelGGL((k_em_rows_global<uint32_t, false>), dim3(gr), dim3(256), 0, s, (const uint32_t *)row_ptr, col, k, m, a);
    }
}

void launch_em_colcount(const uint32_t *col, uint64_t nnz, uint64_t *cnt, unsigned grid, hipStream_t s)
{
    if (nnz) hipLaunchKernelGGL(k_em_colcount, dim3(grid), dim3(256), 0, s, col, nnz, cnt);
}
// dst[i] op= src[i] (same device): the exchange of the EM self test that runs read shards side by side on ONE device
__global__ void k_combine_u64(uint64_t *dst, const uint64_t *src, size_t n) { const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] += src[i]; }
__global__ void k_combine_max_i32(int32_t *dst, const int32_t *src, size_t n) { const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = max(dst[i], src[i]); }
void launch_combine(void *dst, const void *src, size_t n, bool max_i32, hipStream_t s)
{
    if (!n) return;
    if (max_i32) hipLaunchKernelGGL(k_combine_max_i32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (int32_t *)dst, (const int32_t *)src, n);
    else hipLaunchKernelGGL(k_combine_u64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (uint64_t *)dst, (const uint64_t *)src, n);
}
void launch_fill_i32(int32_t *p, uint32_t n, int32_t v, hipStream_t s)
{
    hipLaunchKernelGGL(k_fill_i32, dim3((n + 255) / 256), dim3(256), 0, s, p, n, v);
}
void launch_em_prepare(uint32_t n, const double *mu, const double *l, const uint64_t *colcnt, const int32_t *ref, int measured,
                       uint32_t *word, uint64_t *hi, uint64_t *lo, double *partial, uint64_t *ll, const uint32_t *int_of_ext,
                       hipStream_t s)
{
    hipLaunchKernelGGL(k_em_prepare, dim3((n + 255) / 256), dim3(256), 0, s, n, mu, l, colcnt, ref, measured, word, hi, lo, partial, ll, int_of_ext);
}
void launch_em_check(uint32_t n, const uint32_t *word, const uint64_t *hi, uint64_t *ll, hipStream_t s)
{
    hipLaunchKernelGGL(k_em_check, dim3((n + 255) / 256), dim3(256), 0, s, n, word, hi, ll);
}
void launch_em_finish(const double *partial, uint32_t np, const uint64_t *ll, EmOut *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_em_finish, dim3(1), dim3(1), 0, s, partial, np, ll, out);
}
void launch_em_apply(uint32_t n, double *mu, const double *l, const uint32_t *word, const uint64_t *hi, const uint64_t *lo,
                     int32_t *sexp, hipStream_t s)
{
    hipLaunchKernelGGL(k_em_apply, dim3((n + 255) / 256), dim3(256), 0, s, n, mu, l, word, hi, lo, sexp);
}

} // namespace mmg
