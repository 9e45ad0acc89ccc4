// copy_bw.hip -- what a plain device-to-device copy reaches on this box, for the same bytes as the bench's 64-frame
// 4K batch (531 MB in, 531 MB out): the ceiling the deblocking kernel's copy variant is compared with in DESIGN.md.
// Diagnostic only; not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

template <typename V, int UNROLL>
__global__ __launch_bounds__(256) void k_copy(const V *__restrict__ src, V *__restrict__ dst, size_t n)
{
    // each workgroup copies a contiguous span; UNROLL loads in flight per lane
    const size_t per_wg = (size_t)256 * UNROLL;
    for (size_t base = (size_t)blockIdx.x * per_wg; base < n; base += (size_t)gridDim.x * per_wg) {
        V r[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (i < n) r[u] = src[i];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (i < n) dst[i] = r[u];
        }
    }
}

template <typename V, int UNROLL>
static void run(const char *name, const void *src, void *dst, size_t bytes, int grid_mult)
{
    const size_t n = bytes / sizeof(V);
    const size_t per_wg = (size_t)256 * UNROLL;
    size_t full = (n + per_wg - 1) / per_wg;
    int grid = grid_mult > 0 ? 256 * grid_mult : (int)full;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<float> ms;
    for (int it = 0; it < 160; it++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_copy<V, UNROLL>), dim3(grid), dim3(256), 0, 0, (const V *)src, (V *)dst, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        if (it >= 100) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double med = ms[ms.size() / 2];
    printf("%-28s grid %7d  median %.4f ms  %.3f TB/s (read+write)\n", name, grid, med, 2.0 * bytes / med * 1e-9);
    fflush(stdout);
}

int main()
{
    const size_t bytes = (size_t)64 * 3840 * 2160;
    void *src, *dst;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess) return 1;
    hipMemset(src, 1, bytes);
    hipMemset(dst, 2, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    {
        std::vector<float> ms;
        for (int it = 0; it < 160; it++) {
            hipEventRecord(e0);
            hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float t;
            hipEventElapsedTime(&t, e0, e1);
            if (it >= 100) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-28s               median %.4f ms  %.3f TB/s (read+write)\n", "hipMemcpyDtoD", ms[ms.size() / 2],
               2.0 * bytes / ms[ms.size() / 2] * 1e-9);
    }
    run<uint2, 1>("8 B/lane, 1 in flight", src, dst, bytes, 0);
    run<uint2, 8>("8 B/lane, 8 in flight", src, dst, bytes, 0);
    run<uint4, 1>("16 B/lane, 1 in flight", src, dst, bytes, 0);
    run<uint4, 4>("16 B/lane, 4 in flight", src, dst, bytes, 0);
    run<uint4, 8>("16 B/lane, 8 in flight", src, dst, bytes, 0);
    run<uint4, 4>("16 B/lane, 4, persistent x8", src, dst, bytes, 8);
    run<uint4, 4>("16 B/lane, 4, persistent x16", src, dst, bytes, 16);
    run<uint2, 8>("8 B/lane, 8, persistent x8", src, dst, bytes, 8);
    hipFree(src);
    hipFree(dst);
    return 0;
}
