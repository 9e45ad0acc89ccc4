// valu_rate.hip -- measures the issue rate of integer VALU instructions on gfx950 as a function of
// waves per SIMD.  Diagnostic only (evidence for DESIGN.md's VALU budget); not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int KIND>
__global__ void k_valu(unsigned *out, int iters, unsigned long long *cycles)
{
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned b = blockIdx.x | 1;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if constexpr (KIND == 0) {
                asm volatile("v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n"
                             "v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if constexpr (KIND == 1) {
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if constexpr (KIND == 2) {
                asm volatile("v_perm_b32 %0, %0, %8, %8\n v_perm_b32 %1, %1, %8, %8\n v_perm_b32 %2, %2, %8, %8\n v_perm_b32 %3, %3, %8, %8\n"
                             "v_perm_b32 %4, %4, %8, %8\n v_perm_b32 %5, %5, %8, %8\n v_perm_b32 %6, %6, %8, %8\n v_perm_b32 %7, %7, %8, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if constexpr (KIND == 3) {
                asm volatile("v_pk_max_i16 %0, %0, %8\n v_pk_min_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_min_i16 %3, %3, %8\n"
                             "v_pk_mad_i16 %4, %4, %8, %8\n v_pk_mad_i16 %5, %5, %8, %8\n v_pk_ashrrev_i16 %6, 1, %6\n v_pk_ashrrev_i16 %7, 1, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else {
                asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                             "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int KIND>
void run(const char *name)
{
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {
        const int threads = 64 * 4 * wps > 1024 ? 1024 : 64 * 4 * wps; // waves per CU = 4 SIMDs * wps
        const int blocks_per_cu = (64 * 4 * wps) / threads;
        const int blocks = 256 * blocks_per_cu;
        unsigned *out; unsigned long long *cyc;
        hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k_valu<KIND><<<blocks, threads>>>(out, 10, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k_valu<KIND><<<blocks, threads>>>(out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double insts_per_wave = (double)iters * 64;
        double wave_instr_total = insts_per_wave * (blocks * (threads / 64));
        double per_simd_per_s = wave_instr_total / 1024.0 / (ms * 1e-3);
        // s_memtime ticks at 100 MHz on gfx9-family? report both
        printf("%-10s waves/SIMD %d: %.3f ms  %.3f G wave-instr/s/SIMD  => %.2f cycles/instr @2.4GHz  (memtime ticks %llu)\n",
               name, wps, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, h[0]);
        hipFree(out); hipFree(cyc);
    }
}

int main()
{
    run<0>("pk_add_u16");
    run<1>("add_u32");
    run<2>("perm_b32");
    run<3>("pk_mix");
    run<4>("fma_f32");
    return 0;
}
