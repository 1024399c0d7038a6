// microbenchmark: int32 atomicAdd throughput on MI355X for the count-scatter pattern of K1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

__global__ void k_atomic(const uint32_t* __restrict__ idx, size_t n, int* cnt, uint32_t T, int nrep)
{
    const uint32_t rep = nrep > 1 ? (blockIdx.x % nrep) : 0;
    int* c = cnt + (size_t)rep * T;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __hip_atomic_fetch_add(&c[idx[i]], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_atomic_wg(const uint32_t* __restrict__ idx, size_t n, int* cnt, uint32_t T, int nrep)
{
    const uint32_t rep = nrep > 1 ? (blockIdx.x % nrep) : 0;
    int* c = cnt + (size_t)rep * T;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __hip_atomic_fetch_add(&c[idx[i]], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// LDS histogram for hot (idx < H) columns, global for the rest
template<int H>
__global__ void k_atomic_lds(const uint32_t* __restrict__ idx, size_t n, int* cnt, uint32_t T)
{
    __shared__ int h[H];
    for (int i = threadIdx.x; i < H; i += blockDim.x) h[i] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t c = idx[i];
        if (c < H) atomicAdd(&h[c], 1); else __hip_atomic_fetch_add(&cnt[c], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < H; i += blockDim.x) { int v = h[i]; if (v) __hip_atomic_fetch_add(&cnt[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}
__global__ void k_read(const uint32_t* __restrict__ idx, size_t n, int* cnt)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += idx[i];
    if (acc == 0xdeadbeef) cnt[0] = acc;
}

int main()
{
    const size_t N = 50'000'000; const uint32_t T = 200000;
    std::mt19937_64 rng(1);
    std::vector<uint32_t> uni(N), skew(N), skew_sorted(N);
    std::uniform_int_distribution<uint32_t> U(0, T - 1);
    for (auto& v : uni) v = U(rng);
    // lognormal(0,2) abundance, 30% zero
    std::vector<double> w(T); std::normal_distribution<double> Z(0, 1); std::uniform_real_distribution<double> R(0, 1);
    for (auto& x : w) x = R(rng) < 0.3 ? 0.0 : std::exp(2 * Z(rng));
    std::vector<double> cdf(T); double run = 0; for (uint32_t t = 0; t < T; ++t) { run += w[t]; cdf[t] = run; }
    for (auto& v : skew) { double u = R(rng) * run; v = (uint32_t)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin()); if (v >= T) v = T - 1; }
    // relabelled so that hot columns have the lowest ids
    std::vector<uint32_t> order(T); for (uint32_t t = 0; t < T; ++t) order[t] = t;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return w[a] > w[b]; });
    std::vector<uint32_t> rank(T); for (uint32_t r = 0; r < T; ++r) rank[order[r]] = r;
    for (size_t i = 0; i < N; ++i) skew_sorted[i] = rank[skew[i]];
    uint32_t* d_idx; int* d_cnt; const int MAXREP = 64;
    CK(hipMalloc(&d_idx, N * 4)); CK(hipMalloc(&d_cnt, (size_t)T * 4 * MAXREP));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run_one = [&](const char* name, const std::vector<uint32_t>& h, int mode, int nrep, int grid) {
        CK(hipMemcpy(d_idx, h.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMemset(d_cnt, 0, (size_t)T * 4 * MAXREP));
        float best = 1e9;
        for (int it = 0; it < 4; ++it) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_atomic, dim3(grid), dim3(256), 0, 0, d_idx, N, d_cnt, T, nrep);
            else if (mode == 1) hipLaunchKernelGGL(k_atomic_wg, dim3(grid), dim3(256), 0, 0, d_idx, N, d_cnt, T, nrep);
            else if (mode == 2) hipLaunchKernelGGL(k_atomic_lds<4096>, dim3(grid), dim3(256), 0, 0, d_idx, N, d_cnt, T);
            else if (mode == 3) hipLaunchKernelGGL(k_atomic_lds<16384>, dim3(grid), dim3(256), 0, 0, d_idx, N, d_cnt, T);
            else hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, d_idx, N, d_cnt);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        }
        printf("%-34s nrep=%2d grid=%5d : %8.3f ms  %7.2f G atomics/s\n", name, nrep, grid, best, N / best / 1e6);
    };
    run_one("read-only stream", uni, 9, 1, 2048);
    for (int grid : {768, 2048, 8192}) run_one("uniform agent", uni, 0, 1, grid);
    run_one("uniform workgroup-scope", uni, 1, 1, 2048);
    run_one("skew agent", skew, 0, 1, 2048);
    for (int nrep : {2, 4, 8, 16, 32, 64}) run_one("skew agent replicas", skew, 0, nrep, 2048);
    for (int nrep : {8, 64}) run_one("uniform agent replicas", uni, 0, nrep, 2048);
    run_one("skew workgroup-scope", skew, 1, 1, 2048);
    run_one("skew-sorted LDS hot 4096", skew_sorted, 2, 1, 768);
    run_one("skew-sorted LDS hot 16384", skew_sorted, 3, 1, 768);
    run_one("skew-sorted LDS hot 16384", skew_sorted, 3, 1, 512);
    run_one("skew-sorted agent (no lds)", skew_sorted, 0, 1, 2048);
    return 0;
}
