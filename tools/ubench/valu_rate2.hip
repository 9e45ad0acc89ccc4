// valu_rate2.hip -- issue cost (cycles per wave64 instruction at 4 waves/SIMD) of the integer VALU
// encodings the deblocking kernel can choose between.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define KERNEL(NAME, ASM)                                                                            \
    __global__ void NAME(unsigned *out, int iters)                                                   \
    {                                                                                                \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,   \
                 a6 = a0 + 6, a7 = a0 + 7, b = blockIdx.x | 1;                                        \
        for (int i = 0; i < iters; i++) {                                                            \
            _Pragma("unroll") for (int j = 0; j < 8; j++) {                                          \
                asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),       \
                             "+v"(a6), "+v"(a7) : "v"(b));                                           \
            }                                                                                        \
        }                                                                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;          \
    }

#define L8(fmt) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)
KERNEL(k_add_u32, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
KERNEL(k_add_u32_lit, "v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x12345, %1\n v_add_u32 %2, 0x12345, %2\n v_add_u32 %3, 0x12345, %3\n v_add_u32 %4, 0x12345, %4\n v_add_u32 %5, 0x12345, %5\n v_add_u32 %6, 0x12345, %6\n v_add_u32 %7, 0x12345, %7\n")
KERNEL(k_add_u32_e64, "v_add_u32_e64 %0, %0, %8\n v_add_u32_e64 %1, %1, %8\n v_add_u32_e64 %2, %2, %8\n v_add_u32_e64 %3, %3, %8\n v_add_u32_e64 %4, %4, %8\n v_add_u32_e64 %5, %5, %8\n v_add_u32_e64 %6, %6, %8\n v_add_u32_e64 %7, %7, %8\n")
KERNEL(k_add_sdwa, "v_add_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n")
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_dot4c, "v_dot4c_i32_i8 %0, %8, %8\n v_dot4c_i32_i8 %1, %8, %8\n v_dot4c_i32_i8 %2, %8, %8\n v_dot4c_i32_i8 %3, %8, %8\n v_dot4c_i32_i8 %4, %8, %8\n v_dot4c_i32_i8 %5, %8, %8\n v_dot4c_i32_i8 %6, %8, %8\n v_dot4c_i32_i8 %7, %8, %8\n")
KERNEL(k_dot4, "v_dot4_i32_i8 %0, %0, %8, %0\n v_dot4_i32_i8 %1, %1, %8, %1\n v_dot4_i32_i8 %2, %2, %8, %2\n v_dot4_i32_i8 %3, %3, %8, %3\n v_dot4_i32_i8 %4, %4, %8, %4\n v_dot4_i32_i8 %5, %5, %8, %5\n v_dot4_i32_i8 %6, %6, %8, %6\n v_dot4_i32_i8 %7, %7, %8, %7\n")
KERNEL(k_sat_pk, "v_sat_pk_u8_i16 %0, %0\n v_sat_pk_u8_i16 %1, %1\n v_sat_pk_u8_i16 %2, %2\n v_sat_pk_u8_i16 %3, %3\n v_sat_pk_u8_i16 %4, %4\n v_sat_pk_u8_i16 %5, %5\n v_sat_pk_u8_i16 %6, %6\n v_sat_pk_u8_i16 %7, %7\n")
KERNEL(k_max_i32, "v_max_i32 %0, %0, %8\n v_max_i32 %1, %1, %8\n v_max_i32 %2, %2, %8\n v_max_i32 %3, %3, %8\n v_max_i32 %4, %4, %8\n v_max_i32 %5, %5, %8\n v_max_i32 %6, %6, %8\n v_max_i32 %7, %7, %8\n")
KERNEL(k_med3, "v_med3_i32 %0, %0, %8, 7\n v_med3_i32 %1, %1, %8, 7\n v_med3_i32 %2, %2, %8, 7\n v_med3_i32 %3, %3, %8, 7\n v_med3_i32 %4, %4, %8, 7\n v_med3_i32 %5, %5, %8, 7\n v_med3_i32 %6, %6, %8, 7\n v_med3_i32 %7, %7, %8, 7\n")
KERNEL(k_add3, "v_add3_u32 %0, %0, %8, 7\n v_add3_u32 %1, %1, %8, 7\n v_add3_u32 %2, %2, %8, 7\n v_add3_u32 %3, %3, %8, 7\n v_add3_u32 %4, %4, %8, 7\n v_add3_u32 %5, %5, %8, 7\n v_add3_u32 %6, %6, %8, 7\n v_add3_u32 %7, %7, %8, 7\n")
KERNEL(k_pk_mad, "v_pk_mad_i16 %0, %0, %8, %8\n v_pk_mad_i16 %1, %1, %8, %8\n v_pk_mad_i16 %2, %2, %8, %8\n v_pk_mad_i16 %3, %3, %8, %8\n v_pk_mad_i16 %4, %4, %8, %8\n v_pk_mad_i16 %5, %5, %8, %8\n v_pk_mad_i16 %6, %6, %8, %8\n v_pk_mad_i16 %7, %7, %8, %8\n")
KERNEL(k_mul_i24, "v_mul_i32_i24 %0, %0, %8\n v_mul_i32_i24 %1, %1, %8\n v_mul_i32_i24 %2, %2, %8\n v_mul_i32_i24 %3, %3, %8\n v_mul_i32_i24 %4, %4, %8\n v_mul_i32_i24 %5, %5, %8\n v_mul_i32_i24 %6, %6, %8\n v_mul_i32_i24 %7, %7, %8\n")
KERNEL(k_add_u16, "v_add_u16 %0, %0, %8\n v_add_u16 %1, %1, %8\n v_add_u16 %2, %2, %8\n v_add_u16 %3, %3, %8\n v_add_u16 %4, %4, %8\n v_add_u16 %5, %5, %8\n v_add_u16 %6, %6, %8\n v_add_u16 %7, %7, %8\n")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
KERNEL(k_ashr, "v_ashrrev_i32 %0, 1, %0\n v_ashrrev_i32 %1, 1, %1\n v_ashrrev_i32 %2, 1, %2\n v_ashrrev_i32 %3, 1, %3\n v_ashrrev_i32 %4, 1, %4\n v_ashrrev_i32 %5, 1, %5\n v_ashrrev_i32 %6, 1, %6\n v_ashrrev_i32 %7, 1, %7\n")
KERNEL(k_sad_u8, "v_sad_u8 %0, %0, %8, %0\n v_sad_u8 %1, %1, %8, %1\n v_sad_u8 %2, %2, %8, %2\n v_sad_u8 %3, %3, %8, %3\n v_sad_u8 %4, %4, %8, %4\n v_sad_u8 %5, %5, %8, %5\n v_sad_u8 %6, %6, %8, %6\n v_sad_u8 %7, %7, %8, %7\n")
KERNEL(k_dot2c, "v_dot2c_i32_i16 %0, %8, %8\n v_dot2c_i32_i16 %1, %8, %8\n v_dot2c_i32_i16 %2, %8, %8\n v_dot2c_i32_i16 %3, %8, %8\n v_dot2c_i32_i16 %4, %8, %8\n v_dot2c_i32_i16 %5, %8, %8\n v_dot2c_i32_i16 %6, %8, %8\n v_dot2c_i32_i16 %7, %8, %8\n")

typedef void (*kfn)(unsigned *, int);
static void run(const char *name, kfn k)
{
    const int iters = 4000, wps = 4, threads = 1024, blocks = 256;
    unsigned *out;
    (void)hipMalloc(&out, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, blocks, threads, 0, 0, out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, blocks, threads, 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)iters * 64 * wps; // wave-instructions per SIMD
    printf("%-16s %.3f ms  %.2f ns per wave-instr per SIMD => %.2f cycles @2.4GHz\n", name, ms, ms * 1e6 / per_simd, ms * 1e-3 / per_simd * 2.4e9);
    (void)hipFree(out);
}
int main()
{
    run("add_u32(VOP2)", k_add_u32); run("add_u32+literal", k_add_u32_lit); run("add_u32_e64", k_add_u32_e64);
    run("add_u32_sdwa", k_add_sdwa); run("mov_dpp", k_mov_dpp); run("dot4c_i8(VOP2)", k_dot4c); run("dot4_i8(VOP3P)", k_dot4);
    run("dot2c_i16(VOP2)", k_dot2c); run("sat_pk_u8_i16", k_sat_pk); run("max_i32(VOP2)", k_max_i32); run("med3_i32", k_med3);
    run("add3_u32", k_add3); run("pk_mad_i16", k_pk_mad); run("mul_i32_i24", k_mul_i24); run("add_u16(VOP2)", k_add_u16);
    run("cndmask(VOP2)", k_cndmask); run("ashrrev(VOP2)", k_ashr); run("sad_u8", k_sad_u8);
    return 0;
}
