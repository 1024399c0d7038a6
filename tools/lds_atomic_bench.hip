// LDS atomic / gather throughput on one window of 256 entries per wave-sized workgroup (what the EM and sample kernels do).
// Address patterns: uniform random, all lanes equal, skewed (a few hot entries).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

template <int MODE> // 0: ds_add_u64 x1, 1: ds_add_u64 x2 (two arrays), 2: ds_add_u32, 3: ds_read_b64 gather, 4: ds_max_i32,
                    // 5: ds_add_rtn_u64 + carry test, 6: the EM mix (b64 gather, b32 read, 2 adds), 7: the mix with ONE returned add
__global__ __launch_bounds__(64) void k(const uint16_t *__restrict__ idx, int per_lane, uint64_t *out)
{
    __shared__ uint64_t a[257], b[257];
    __shared__ uint32_t c[257];
    for (int i = threadIdx.x; i < 257; i += 64) { a[i] = i; b[i] = 0; c[i] = 0; }
    __syncthreads();
    const uint16_t *p = idx + ((size_t)blockIdx.x * 64 + threadIdx.x) * per_lane;
    uint64_t acc = 0;
    for (int j = 0; j < per_lane; j += 4) {
        const uint2 v = *(const uint2 *)(p + j);
        const uint32_t o[4] = {v.x & 0xffffu, v.x >> 16, v.y & 0xffffu, v.y >> 16};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MODE == 0) atomicAdd((unsigned long long *)&a[o[q]], 3ull);
            if (MODE == 1) { atomicAdd((unsigned long long *)&a[o[q]], 3ull); atomicAdd((unsigned long long *)&b[o[q]], 5ull); }
            if (MODE == 2) atomicAdd(&c[o[q]], 1u);
            if (MODE == 3) acc += a[o[q]];
            if (MODE == 4) atomicMax((int *)&c[o[q]], (int)j);
            if (MODE == 5 || MODE == 7) { // one returned add; the carry out of the 64 bits goes to the second array (rare)
                const uint64_t y = 0x0123456789abcdefull + o[q];
                const uint64_t old = atomicAdd((unsigned long long *)&a[o[q]], (unsigned long long)y);
                if (old + y < y) atomicAdd((unsigned long long *)&b[o[q]], 1ull);
            }
            if (MODE == 6 || MODE == 7) acc += a[o[q]] + c[o[q]];     // the gather and the scale word of an EM hit
            if (MODE == 6) { atomicAdd((unsigned long long *)&a[o[q]], 3ull); atomicAdd((unsigned long long *)&b[o[q]], 5ull); }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = a[1] + b[2] + c[3] + acc;
}

int main()
{
    const int per_lane = 512, grid = 256 * 20, lanes = grid * 64;
    const size_t n = (size_t)lanes * per_lane;
    std::vector<uint16_t> h(n);
    uint16_t *d; uint64_t *out;
    CK(hipMalloc(&d, n * 2)); CK(hipMalloc(&out, grid * 8));
    const char *pat[] = {"uniform random", "all lanes same address", "skewed (lognormal sigma 2 over 256)", "two hot entries 50/50"};
    for (int ptn = 0; ptn < 4; ++ptn) {
        std::vector<double> cdf(256); double run = 0; srand(1);
        for (int i = 0; i < 256; ++i) { double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
            run += exp(2.0 * sqrt(-2 * log(u1)) * cos(6.283185307 * u2)); cdf[i] = run; }
        for (size_t i = 0; i < n; ++i) {
            uint16_t v;
            if (ptn == 0) v = rand() & 255;
            else if (ptn == 1) v = (i / per_lane / 64 * 7 + (i % per_lane)) & 255; // same across the wave's lanes, varies in time
            else if (ptn == 2) { double u = (rand() / (RAND_MAX + 1.0)) * run; int lo = 0; while (lo < 255 && cdf[lo] < u) ++lo; v = lo; }
            else v = (rand() & 1) ? 17 : 200;
            h[i] = v;
        }
        if (ptn == 1) for (size_t i = 0; i < n; ++i) h[i] = (uint16_t)(((i % per_lane) * 13 + (i / ((size_t)per_lane * 64))) & 255);
        CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
        const char *names[] = {"ds_add_u64", "2 x ds_add_u64", "ds_add_u32", "ds_read_b64 gather", "ds_max_i32", "ds_add_rtn_u64+carry", "EM mix, 2 adds", "EM mix, 1 rtn add"};
        for (int m = 0; m < 8; ++m) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0));
                if (m == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, d, per_lane, out);
                if (m == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, d, per_lane, out);
                if (m == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), 0, 0, d, per_lane, out);
                if (m == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(64), 0, 0, d, per_lane, out);
                if (m == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(64), 0, 0, d, per_lane, out);
                if (m == 5) hipLaunchKernelGGL(k<5>, dim3(grid), dim3(64), 0, 0, d, per_lane, out);
                if (m == 6) hipLaunchKernelGGL(k<6>, dim3(grid), dim3(64), 0, 0, d, per_lane, out);
                if (m == 7) hipLaunchKernelGGL(k<7>, dim3(grid), dim3(64), 0, 0, d, per_lane, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double ops = (double)n * (m == 1 ? 2 : 1);
            printf("%-38s %-20s %7.3f ms  %8.1f G lane-ops/s  %6.2f lanes/clk/CU @2.4GHz\n", pat[ptn], names[m], best, ops / best / 1e6,
                   ops / (best * 1e-3) / 256 / 2.4e9);
        }
    }
    return 0;
}
