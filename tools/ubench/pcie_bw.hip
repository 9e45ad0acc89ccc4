// pcie_bw.hip -- what the host link of this box moves: page-locked host memory <-> HBM with hipMemcpyAsync, one direction
// at a time and both at once on two streams, for the transfer sizes the host pipeline uses (one 4K luma frame = 8.3 MB,
// strips of 2-4 MB, chunks of 64 MB).  The ceiling the streaming operator's end-to-end rate is compared with.
// Diagnostic only; not part of the product.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t sizes[] = {(size_t)2 << 20, (size_t)3840 * 2160, (size_t)32 << 20, (size_t)64 << 20};
    const size_t maxb = (size_t)64 << 20;
    void *h_in, *h_out, *d_in, *d_out;
    if (hipHostMalloc(&h_in, maxb, hipHostMallocDefault) != hipSuccess || hipHostMalloc(&h_out, maxb, hipHostMallocDefault) != hipSuccess ||
        hipMalloc(&d_in, maxb) != hipSuccess || hipMalloc(&d_out, maxb) != hipSuccess)
        return 1;
    hipStream_t s0, s1;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    for (size_t b : sizes) {
        const int reps = (int)(((size_t)2 << 30) / b);
        for (int w = 0; w < 3; w++) hipMemcpyAsync(d_in, h_in, b, hipMemcpyHostToDevice, s0);
        hipStreamSynchronize(s0);
        double t0 = now();
        for (int i = 0; i < reps; i++) hipMemcpyAsync(d_in, h_in, b, hipMemcpyHostToDevice, s0);
        hipStreamSynchronize(s0);
        const double h2d = (double)b * reps / (now() - t0) * 1e-9;
        t0 = now();
        for (int i = 0; i < reps; i++) hipMemcpyAsync(h_out, d_out, b, hipMemcpyDeviceToHost, s1);
        hipStreamSynchronize(s1);
        const double d2h = (double)b * reps / (now() - t0) * 1e-9;
        t0 = now();
        for (int i = 0; i < reps; i++) {
            hipMemcpyAsync(d_in, h_in, b, hipMemcpyHostToDevice, s0);
            hipMemcpyAsync(h_out, d_out, b, hipMemcpyDeviceToHost, s1);
        }
        hipStreamSynchronize(s0);
        hipStreamSynchronize(s1);
        const double both = 2.0 * (double)b * reps / (now() - t0) * 1e-9;
        printf("%9zu bytes per copy: H2D %.1f GB/s, D2H %.1f GB/s, both at once %.1f GB/s (sum of the two directions)\n", b, h2d, d2h, both);
        fflush(stdout);
    }
    return 0;
}
