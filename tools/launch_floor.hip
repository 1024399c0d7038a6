// What a launch of K1's geometry costs before it does anything: HIP-event time of (a) an empty kernel, (b) one that reads its range header
// with a scalar load and 255 doubles of a window, (c) the same plus one 1.5 KB block per wave -- 7168 single-wave workgroups (28 per CU),
// the grid of a config-2 launch.  K1 at config 2 takes 26 us for 10.9 tiles per workgroup; this is the part that is not tiles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(64) void k_empty(const uint64_t *, const double *, const uint32_t *, double *) {}
__global__ __launch_bounds__(64) void k_header(const uint64_t *hdr, const double *mu, const uint32_t *, double *out)
{
    __shared__ double s[256];
    const uint64_t b = hdr[(size_t)blockIdx.x * 8];
    for (int i = threadIdx.x; i < 255; i += 64) s[i] = mu[(b & 0xffff) + i];
    __syncthreads();
    if (s[threadIdx.x] == 12345.678) out[0] = 1.0;
}
__global__ __launch_bounds__(64) void k_block(const uint64_t *hdr, const double *mu, const uint32_t *stream, double *out)
{
    __shared__ double s[256];
    const uint64_t b = hdr[(size_t)blockIdx.x * 8];
    for (int i = threadIdx.x; i < 255; i += 64) s[i] = mu[(b & 0xffff) + i];
    uint32_t g[6];
    for (int i = 0; i < 6; ++i) g[i] = stream[((size_t)blockIdx.x * 6 + i) * 64 + threadIdx.x];
    __syncthreads();
    double t = 0;
    for (int i = 0; i < 6; ++i) t += s[g[i] & 255];
    if (t == 12345.678) out[0] = t;
}
int main()
{
    const int grid = 7168;
    uint64_t *hdr; double *mu, *out; uint32_t *stream;
    CK(hipMalloc(&hdr, grid * 64)); CK(hipMalloc(&mu, 1 << 20)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&stream, (size_t)grid * 6 * 256));
    CK(hipMemset(hdr, 0, grid * 64)); CK(hipMemset(mu, 0, 1 << 20)); CK(hipMemset(stream, 0, (size_t)grid * 6 * 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[3] = {"empty", "header + window", "header + window + one block"};
    for (int k = 0; k < 3; ++k) {
        for (int warm = 0; warm < 2000; ++warm) {
            if (k == 0) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(64), 0, 0, hdr, mu, stream, out);
            else if (k == 1) hipLaunchKernelGGL(k_header, dim3(grid), dim3(64), 0, 0, hdr, mu, stream, out);
            else hipLaunchKernelGGL(k_block, dim3(grid), dim3(64), 0, 0, hdr, mu, stream, out);
        }
        CK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0;
        const int n = 200;
        for (int i = 0; i < n; ++i) {
            CK(hipEventRecord(e0, 0));
            if (k == 0) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(64), 0, 0, hdr, mu, stream, out);
            else if (k == 1) hipLaunchKernelGGL(k_header, dim3(grid), dim3(64), 0, 0, hdr, mu, stream, out);
            else hipLaunchKernelGGL(k_block, dim3(grid), dim3(64), 0, 0, hdr, mu, stream, out);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best; sum += ms;
        }
        // back to back: 200 launches between one pair of events
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < n; ++i) {
            if (k == 0) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(64), 0, 0, hdr, mu, stream, out);
            else if (k == 1) hipLaunchKernelGGL(k_header, dim3(grid), dim3(64), 0, 0, hdr, mu, stream, out);
            else hipLaunchKernelGGL(k_block, dim3(grid), dim3(64), 0, 0, hdr, mu, stream, out);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float msb; CK(hipEventElapsedTime(&msb, e0, e1));
        printf("%-32s %d x 64: event pair around one launch %.2f us (best %.2f), back to back %.2f us per launch\n", names[k], grid, sum / n * 1e3, best * 1e3, msb / n * 1e3);
    }
    return 0;
}
