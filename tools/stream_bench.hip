// What HBM read bandwidth does K1's access pattern reach on MI355X?  (K1 measures 5.5 TB/s of counter traffic; a float4 copy 6.3.)
// Pattern A ("K1"): single-wave workgroups, each owning ONE contiguous range of `tiles` blocks of `blk` bytes; a block is read as
//   blk/256 wave-loads of 4 bytes per lane, two blocks in flight per wave (the register double buffer of k_sample_sell).
// Pattern B: the same ranges, 16 bytes per lane per load (1 KB per wave-instruction).
// Pattern C: grid-stride, every wave-load 1 KB, neighbouring waves read neighbouring kilobytes (a plain streaming read).
// usage: stream_bench [GB=1.2] [blk=1472] [tiles=18]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NL>
__global__ __launch_bounds__(64) void k_ranges_dword(const uint32_t *__restrict__ src, uint64_t range_dw, uint32_t tiles, uint32_t blk_dw, uint32_t *out)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t *p = src + (uint64_t)blockIdx.x * range_dw + lane;
    uint32_t acc = 0;
    uint32_t a[NL], b[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) a[i] = __builtin_nontemporal_load(p + i * 64);
#pragma unroll
    for (int i = 0; i < NL; ++i) b[i] = __builtin_nontemporal_load(p + blk_dw + i * 64);
    for (uint32_t t = 0; t < tiles; t += 2) {
        const uint32_t *q = p + (uint64_t)(t + 2 < tiles ? t + 2 : t) * blk_dw;
#pragma unroll
        for (int i = 0; i < NL; ++i) { acc += a[i]; a[i] = __builtin_nontemporal_load(q + i * 64); }
        const uint32_t *r = p + (uint64_t)(t + 3 < tiles ? t + 3 : t + 1 < tiles ? t + 1 : t) * blk_dw;
#pragma unroll
        for (int i = 0; i < NL; ++i) { acc += b[i]; b[i] = __builtin_nontemporal_load(r + i * 64); }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k_ranges_x4(const u32x4 *__restrict__ src, uint64_t range_q, uint32_t loads, uint32_t *out)
{
    const u32x4 *p = src + (uint64_t)blockIdx.x * range_q + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + 64);
    for (uint32_t t = 0; t < loads; t += 2) {
        acc += a; a = __builtin_nontemporal_load(p + (uint64_t)(t + 2 < loads ? t + 2 : t) * 64);
        acc += b; b = __builtin_nontemporal_load(p + (uint64_t)(t + 3 < loads ? t + 3 : t) * 64);
    }
    if (acc.x + acc.y + acc.z + acc.w == 0x12345678u) out[0] = acc.x;
}

__global__ __launch_bounds__(256) void k_stride_x4(const u32x4 *__restrict__ src, uint64_t n_q, uint32_t *out)
{
    u32x4 acc = {0, 0, 0, 0};
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n_q; i += 4 * stride) {
        const u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride), c = __builtin_nontemporal_load(src + i + 2 * stride),
                    d = __builtin_nontemporal_load(src + i + 3 * stride);
        acc += a; acc += b; acc += c; acc += d;
    }
    for (; i < n_q; i += stride) acc += __builtin_nontemporal_load(src + i);
    if (acc.x + acc.y + acc.z + acc.w == 0x12345678u) out[0] = acc.x;
}

template <typename F>
static double time_ms(F f, int reps = 20)
{
    f(); CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const double gb = argc > 1 ? atof(argv[1]) : 1.2;
    const uint32_t blk = argc > 2 ? atoi(argv[2]) : 1472, tiles = argc > 3 ? atoi(argv[3]) : 18; // 64 + 5.5 x 256
    const uint32_t blk_dw = (blk + 255) / 256 * 64;                     // whole 256-byte wave-loads
    const uint64_t range_dw = (uint64_t)blk_dw * tiles, n_ranges = (uint64_t)(gb * 1e9 / 4 / range_dw);
    const uint64_t total_dw = range_dw * n_ranges;
    uint32_t *d, *out;
    CK(hipMalloc(&d, total_dw * 4 + 4096)); CK(hipMalloc(&out, 64));
    CK(hipMemset(d, 1, total_dw * 4 + 4096));
    printf("# %.3f GB in %llu ranges of %u blocks x %u bytes; single-wave workgroups\n", total_dw * 4 / 1e9, (unsigned long long)n_ranges, tiles, blk_dw * 4);
    // warm the clocks
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_stride_x4, dim3(256 * 8), dim3(256), 0, 0, (const u32x4 *)d, total_dw / 4, out);
    CK(hipDeviceSynchronize());
    double ms;
    ms = time_ms([&] { hipLaunchKernelGGL(k_ranges_dword<6>, dim3((unsigned)n_ranges), dim3(64), 0, 0, d, range_dw, tiles, blk_dw, out); });
    printf("A  ranges, 4 B per lane per load, 2 blocks in flight : %.4f ms  %.2f TB/s\n", ms, total_dw * 4 / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_ranges_x4, dim3((unsigned)n_ranges), dim3(64), 0, 0, (const u32x4 *)d, range_dw / 4, (uint32_t)(range_dw / 256), out); });
    printf("B  ranges, 16 B per lane per load, 2 KB in flight    : %.4f ms  %.2f TB/s\n", ms, total_dw * 4 / ms / 1e9);
    for (int wg : {4, 8, 16}) {
        ms = time_ms([&] { hipLaunchKernelGGL(k_stride_x4, dim3(256 * wg), dim3(256), 0, 0, (const u32x4 *)d, total_dw / 4, out); });
        printf("C  grid-stride, 16 B per lane, %2d x 256-thread WGs/CU : %.4f ms  %.2f TB/s\n", wg, ms, total_dw * 4 / ms / 1e9);
    }
    return 0;
}
