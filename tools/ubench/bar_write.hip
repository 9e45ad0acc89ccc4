// bar_write.hip -- can the host's cores write a frame straight into HBM through the PCIe BAR (posted writes, no staging ring and no
// H2D DMA), and how fast?  hipMalloc memory and fine-grained device memory (hipExtMallocWithFlags), 1..8 threads, non-temporal
// stores of 8.3 MB (one 4K luma frame) from ordinary pageable memory; the result is read back with a DMA and compared.  Each
// allocation kind runs in a child process: where the memory is not host-accessible the child dies of SIGSEGV and the parent says so.
// Diagnostic only; not part of the product.
#include <hip/hip_runtime.h>
#include <emmintrin.h>
#include <sys/wait.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void nt_copy(uint8_t *d, const uint8_t *s, size_t n)
{
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        __m128i a = _mm_loadu_si128((const __m128i *)(s + i)), b = _mm_loadu_si128((const __m128i *)(s + i + 16));
        __m128i c = _mm_loadu_si128((const __m128i *)(s + i + 32)), e = _mm_loadu_si128((const __m128i *)(s + i + 48));
        _mm_stream_si128((__m128i *)(d + i), a);
        _mm_stream_si128((__m128i *)(d + i + 16), b);
        _mm_stream_si128((__m128i *)(d + i + 32), c);
        _mm_stream_si128((__m128i *)(d + i + 48), e);
    }
    _mm_sfence();
}

static int child(int kind)
{
    const size_t B = (size_t)3840 * 2160;
    uint8_t *src = (uint8_t *)aligned_alloc(4096, B), *back = (uint8_t *)aligned_alloc(4096, B);
    for (size_t i = 0; i < B; i++) src[i] = (uint8_t)(i * 2654435761u >> 24);
    uint8_t *dev = nullptr;
    hipError_t e = kind == 0 ? hipMalloc((void **)&dev, B) : hipExtMallocWithFlags((void **)&dev, B, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { printf("  \"alloc_error\": \"%s\"\n", hipGetErrorString(e)); return 0; }
    hipMemset(dev, 0, B);
    hipDeviceSynchronize();
    const int ts[] = {1, 2, 3, 4, 6, 8};
    printf("  \"GBps_by_threads\": {");
    for (int ti = 0; ti < 6; ti++) {
        const int T = ts[ti];
        double best = 0;
        for (int r = 0; r < 6; r++) {
            std::atomic<int> go{0};
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++)
                th.emplace_back([&, t] {
                    while (!go.load(std::memory_order_acquire)) {}
                    const size_t a = (B / 64 * t / T) * 64, b = t + 1 == T ? B : (B / 64 * (t + 1) / T) * 64;
                    nt_copy(dev + a, src + a, b - a);
                });
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
            const double t0 = now();
            go.store(1, std::memory_order_release);
            for (auto &t : th) t.join();
            const double gb = (double)B / (now() - t0) * 1e-9;
            if (gb > best) best = gb;
        }
        printf("%s\"%d\": %.1f", ti ? ", " : "", T, best);
        fflush(stdout);
    }
    printf("},\n");
    hipMemcpy(back, dev, B, hipMemcpyDeviceToHost);
    printf("  \"read_back_equal\": %s\n", std::memcmp(back, src, B) == 0 ? "true" : "false");
    fflush(stdout);
    return 0;
}

int main()
{
    const char *names[2] = {"hipMalloc", "hipExtMallocWithFlags_finegrained"};
    printf("{\n");
    for (int kind = 0; kind < 2; kind++) {
        printf(" \"%s\": {\n", names[kind]);
        fflush(stdout);
        pid_t p = fork(); /* before any HIP call in this process */
        if (p == 0) _exit(child(kind));
        int st = 0;
        waitpid(p, &st, 0);
        if (WIFSIGNALED(st)) printf("  \"host_access\": \"no: child ended with signal %d\"\n", WTERMSIG(st));
        printf(" }%s\n", kind == 0 ? "," : "");
        fflush(stdout);
    }
    printf("}\n");
    return 0;
}
