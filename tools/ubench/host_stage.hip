// host_stage.hip -- what each step of the host-frame operator can cost on this box, one at a time (round 4: VERDICT r03 item 1):
//   1. pageable -> page-locked staging copies of one 4K luma frame (8.3 MB) with 1..16 host threads: glibc memcpy row by row
//      (3840-byte rows, what deblock_host.cpp did), memcpy of whole strips, and non-temporal stores;
//   2. hipHostRegister / hipHostUnregister of the caller's 8.3 MB (the alternative to staging for callers who reuse buffers);
//   3. one DMA, enqueue to completion, by size (the strip size trade-off), each direction;
//   4. hipMemcpy straight from / to pageable memory (the runtime's own staging);
//   5. a device kernel copying page-locked host memory to page-locked host memory (no DMA, the small-frame path's mechanism).
// Diagnostic only; not part of the product.  Output: one JSON object on stdout.
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void nt_copy(uint8_t *dst, const uint8_t *src, size_t n)
{
    // dst 32-byte aligned by construction below
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        __m256i a = _mm256_loadu_si256((const __m256i *)(src + i)), b = _mm256_loadu_si256((const __m256i *)(src + i + 32));
        __m256i c = _mm256_loadu_si256((const __m256i *)(src + i + 64)), d = _mm256_loadu_si256((const __m256i *)(src + i + 96));
        _mm256_stream_si256((__m256i *)(dst + i), a);
        _mm256_stream_si256((__m256i *)(dst + i + 32), b);
        _mm256_stream_si256((__m256i *)(dst + i + 64), c);
        _mm256_stream_si256((__m256i *)(dst + i + 96), d);
    }
    if (i < n) std::memcpy(dst + i, src + i, n - i);
    _mm_sfence();
}

enum Mode { ROWS, BLOCK, NT };

static double staged_copy_gbps(uint8_t *dst, const uint8_t *src, size_t rows, size_t rb, int threads, Mode m, int reps)
{
    double best = 0;
    for (int r = 0; r < reps; r++) {
        std::atomic<int> go{0};
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++)
            th.emplace_back([&, t] {
                while (!go.load(std::memory_order_acquire)) {}
                const size_t r0 = rows * t / threads, r1 = rows * (t + 1) / threads;
                if (m == ROWS)
                    for (size_t y = r0; y < r1; y++) std::memcpy(dst + y * rb, src + y * rb, rb);
                else if (m == BLOCK)
                    std::memcpy(dst + r0 * rb, src + r0 * rb, (r1 - r0) * rb);
                else
                    nt_copy(dst + r0 * rb, src + r0 * rb, (r1 - r0) * rb);
            });
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        const double t0 = now();
        go.store(1, std::memory_order_release);
        for (auto &t : th) t.join();
        const double gb = (double)rows * rb / (now() - t0) * 1e-9;
        if (gb > best) best = gb;
    }
    return best;
}

__global__ void copy16(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

int main()
{
    const size_t W = 3840, H = 2160, B = W * H;
    uint8_t *page_in = (uint8_t *)aligned_alloc(4096, B), *page_out = (uint8_t *)aligned_alloc(4096, B);
    std::memset(page_in, 7, B);
    std::memset(page_out, 9, B);
    uint8_t *pin_a, *pin_b, *dev;
    if (hipHostMalloc((void **)&pin_a, B, hipHostMallocDefault) != hipSuccess || hipHostMalloc((void **)&pin_b, B, hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void **)&dev, B) != hipSuccess)
        return 1;
    std::memset(pin_a, 1, B);
    std::memset(pin_b, 2, B);
    hipStream_t s0;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    printf("{\"frame_bytes\": %zu, \"host_threads_available\": %u,\n", B, std::thread::hardware_concurrency());

    // 1. staging copies
    const char *mn[3] = {"memcpy_rows_3840B", "memcpy_block", "nt_store_block"};
    printf(" \"stage_in_pageable_to_pinned_GBps\": {");
    for (int m = 0; m < 3; m++) {
        printf("%s\"%s\": {", m ? ", " : "", mn[m]);
        const int ts[] = {1, 2, 3, 4, 6, 8, 12, 16};
        for (int i = 0; i < 8; i++) printf("%s\"%d\": %.1f", i ? ", " : "", ts[i], staged_copy_gbps(pin_a, page_in, H, W, ts[i], (Mode)m, 7));
        printf("}");
    }
    printf("},\n \"stage_out_pinned_to_pageable_GBps\": {");
    for (int m = 0; m < 3; m++) {
        printf("%s\"%s\": {", m ? ", " : "", mn[m]);
        const int ts[] = {1, 2, 3, 4, 6, 8, 12, 16};
        for (int i = 0; i < 8; i++) printf("%s\"%d\": %.1f", i ? ", " : "", ts[i], staged_copy_gbps(page_out, pin_b, H, W, ts[i], (Mode)m, 7));
        printf("}");
    }
    printf("},\n");
    fflush(stdout);

    // 2. registration of caller memory
    {
        double reg = 1e9, unreg = 1e9, reg_first = 0;
        for (int r = 0; r < 6; r++) {
            double t0 = now();
            if (hipHostRegister(page_in, B, hipHostRegisterDefault) != hipSuccess) { reg = -1; break; }
            double t1 = now();
            hipHostUnregister(page_in);
            double t2 = now();
            if (r == 0) reg_first = t1 - t0;
            if (t1 - t0 < reg) reg = t1 - t0;
            if (t2 - t1 < unreg) unreg = t2 - t1;
        }
        printf(" \"host_register_8MB_ms\": {\"first\": %.3f, \"best\": %.3f, \"unregister_best\": %.3f},\n", reg_first * 1e3, reg * 1e3, unreg * 1e3);
        // DMA in place from registered memory
        if (hipHostRegister(page_in, B, hipHostRegisterDefault) == hipSuccess) {
            double best = 1e9;
            for (int r = 0; r < 8; r++) {
                double t0 = now();
                hipMemcpyAsync(dev, page_in, B, hipMemcpyHostToDevice, s0);
                hipStreamSynchronize(s0);
                double t = now() - t0;
                if (t < best) best = t;
            }
            printf(" \"dma_h2d_from_registered_8MB_ms\": %.3f,\n", best * 1e3);
            hipHostUnregister(page_in);
        }
    }
    fflush(stdout);

    // 3. one DMA by size, enqueue -> completion observed by the host
    printf(" \"one_dma_us_by_bytes\": {");
    const size_t sz[] = {64 << 10, 256 << 10, 512 << 10, 1 << 20, 2 << 20, 4 << 20, B};
    for (int i = 0; i < 7; i++) {
        double h2d = 1e9, d2h = 1e9;
        for (int r = 0; r < 12; r++) {
            double t0 = now();
            hipMemcpyAsync(dev, pin_a, sz[i], hipMemcpyHostToDevice, s0);
            hipStreamSynchronize(s0);
            double t1 = now();
            hipMemcpyAsync(pin_b, dev, sz[i], hipMemcpyDeviceToHost, s0);
            hipStreamSynchronize(s0);
            double t2 = now();
            if (t1 - t0 < h2d) h2d = t1 - t0;
            if (t2 - t1 < d2h) d2h = t2 - t1;
        }
        printf("%s\"%zu\": {\"h2d\": %.1f, \"d2h\": %.1f}", i ? ", " : "", sz[i], h2d * 1e6, d2h * 1e6);
    }
    printf("},\n");
    fflush(stdout);

    // 4. the runtime's own path for pageable memory
    {
        double h2d = 1e9, d2h = 1e9;
        for (int r = 0; r < 6; r++) {
            double t0 = now();
            hipMemcpy(dev, page_in, B, hipMemcpyHostToDevice);
            double t1 = now();
            hipMemcpy(page_out, dev, B, hipMemcpyDeviceToHost);
            double t2 = now();
            if (t1 - t0 < h2d) h2d = t1 - t0;
            if (t2 - t1 < d2h) d2h = t2 - t1;
        }
        printf(" \"hipMemcpy_pageable_8MB_ms\": {\"h2d\": %.3f, \"d2h\": %.3f},\n", h2d * 1e3, d2h * 1e3);
    }
    fflush(stdout);

    // 5. a kernel moving page-locked host memory itself
    {
        const size_t n = B / 16;
        double best = 1e9, best_h2d = 1e9, best_d2h = 1e9;
        for (int r = 0; r < 8; r++) {
            double t0 = now();
            copy16<<<2048, 256, 0, s0>>>((const uint4 *)pin_a, (uint4 *)pin_b, n);
            hipStreamSynchronize(s0);
            double t1 = now();
            copy16<<<2048, 256, 0, s0>>>((const uint4 *)pin_a, (uint4 *)dev, n);
            hipStreamSynchronize(s0);
            double t2 = now();
            copy16<<<2048, 256, 0, s0>>>((const uint4 *)dev, (uint4 *)pin_b, n);
            hipStreamSynchronize(s0);
            double t3 = now();
            if (t1 - t0 < best) best = t1 - t0;
            if (t2 - t1 < best_h2d) best_h2d = t2 - t1;
            if (t3 - t2 < best_d2h) best_d2h = t3 - t2;
        }
        printf(" \"kernel_copy_8MB_ms\": {\"host_to_host\": %.3f, \"host_to_hbm\": %.3f, \"hbm_to_host\": %.3f}\n}\n", best * 1e3, best_h2d * 1e3,
               best_d2h * 1e3);
    }
    return 0;
}
