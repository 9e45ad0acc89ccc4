// store_hazard_probe.hip -- COMPILE-ONLY probe of the toolchain's hazard table for stores of more than 8 bytes whose data registers the
// next VALU instruction overwrites (profiles/r04/store_hazard.md).  k_imm: the buffer store's soffset is an immediate -> hipcc inserts
// `s_nop 1` (two wait states) for gfx942 / gfx950 and `s_nop 0` (one) for gfx90a.  k_sgpr: soffset is an SGPR -> nothing is inserted;
// that is the form whose data MI355X was seen to corrupt, and the one the kernels guard by hand and tools/check_store_hazard.py scans
// for.  tests/test_abi_cpu.py compiles this file and checks both facts, so a toolchain that changes its table is noticed.
#include <hip/hip_runtime.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k_imm(uint8_t *p, int n, const uint32_t *q)
{
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(p, 0, (uint32_t)n, 0x00020000);
    u32x4 w; w.x = q[threadIdx.x]; w.y = q[threadIdx.x + 64]; w.z = q[threadIdx.x + 128]; w.w = q[threadIdx.x + 192];
    __builtin_amdgcn_raw_buffer_store_b128(w, rd, threadIdx.x * 16, 0, 0);     // soffset = 0 (not a register)
    w.x += 7; w.y ^= w.x; w.z += w.y; w.w -= w.z;
    __builtin_amdgcn_raw_buffer_store_b128(w, rd, threadIdx.x * 16 + 4096, 0, 0);
}
__global__ void k_sgpr(uint8_t *p, int n, const uint32_t *q, int soff)
{
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(p, 0, (uint32_t)n, 0x00020000);
    u32x4 w; w.x = q[threadIdx.x]; w.y = q[threadIdx.x + 64]; w.z = q[threadIdx.x + 128]; w.w = q[threadIdx.x + 192];
    __builtin_amdgcn_raw_buffer_store_b128(w, rd, threadIdx.x * 16, soff, 0);  // soffset = SGPR
    w.x += 7; w.y ^= w.x; w.z += w.y; w.w -= w.z;
    __builtin_amdgcn_raw_buffer_store_b128(w, rd, threadIdx.x * 16 + 4096, soff, 0);
}
