/*
 * hevcdbk_diag.h -- the DIAGNOSTIC build of the library (libhevcdbk_diag.so = the same sources compiled with
 * -DHEVCDBK_DIAG).  It carries what must never ship: the copy variant of the packed kernels (their loads and stores with
 * no arithmetic), the timing-only ablations of the luma filter (WRONG pixels), the measured-and-rejected LDS-queue kernel,
 * launch-geometry knobs.  Tools under tools/ and the ablation tests load it; libhevcdbk.so has none of these symbols,
 * none of these kernels, and reads no environment variable.
 */
#pragma once
#include "../../include/hevc_deblock.h"

#ifdef __cplusplus
extern "C" {
#endif

/* kernel selector of hevc_deblocking_filter_device / hevcdbk_device_run_timed, diagnostic library only: the packed
 * kernel's loads and stores with no arithmetic (dst = src) -- the memory-path ceiling of the access pattern */
#define HEVCDBK_DIAG_KERNEL_COPY 100

/* block map selector (OR-ed into the kernel selector like HEVCDBK_MAP_*), diagnostic library only: persistent waves
 * walking down stripes of the batch, the next tile and its bS bytes prefetched into LDS by buffer_load ... lds
 * (8-bit luma, scalar QP, width a multiple of 128; other operands take the geometry's own map).  Bit-exact; measured
 * SLOWER than the plain maps on MI355X, which is why it is not in the product. */
#define HEVCDBK_DIAG_MAP_STRIPE 0x300
/* likewise: whole block rows staged in LDS by a workgroup (naturally aligned 16-byte-per-lane LDS-DMA in, aligned 16-byte
 * stores out, blocks filtered out of LDS between two barriers, column bx = 0 by a second tiny launch); 8-bit luma, scalar
 * QP, width a multiple of 128.  Bit-exact; measured SLOWER than the plain maps. */
#define HEVCDBK_DIAG_MAP_TILES 0x400
/* likewise: the row map with N block rows per workgroup (knob rows=N), a wave walking down its 64 columns with the next
 * row's tile prefetched in registers while the current one is filtered */
#define HEVCDBK_DIAG_MAP_PIPE 0x500
/* likewise: k whole block rows (minus the column bx = 0) per workgroup, row-major inside the group so that no lane idles,
 * plain one-shot waves; the frame border by extra workgroups of the same launch */
#define HEVCDBK_DIAG_MAP_GROUP 0x600

/* comma-separated knobs, process-wide, replacing the previous set (NULL or "" = defaults):
 *   wg=N      workgroup width cap of the packed kernels (64..1024, default 512)
 *   noswz     row-major map without the per-XCD workgroup renumbering
 *   nofuse    Y, U, V as three launches even where the fused launch applies
 *   dmacopy   small frames through DMA copies instead of host-direct kernels
 *   queue     8-bit luma through the LDS-queue kernel (bit-exact, slower)
 *   align     copy variant: row spans shifted onto their natural alignment
 *   mode3     8-bit luma through the instrumented instantiation of the kernel with no knob active (A/B baseline)
 *   prio=N    wave priority experiment: bit 0 = s_setprio 3 until the row loads are issued, bit 1 = from the final pack on (8-bit
 *             fused deblocking + SAO kernel: bit 1 = s_setprio 2 behind the barrier, for the whole of stage 2)
 *   dummy=N   N extra VALU instructions per wave (how the kernel time responds to VALU work); with the stripe map
 *             a bit set of timing experiments instead: 1 no bS DMA, 2 no stores, 4 no tile DMA, 8 no border,
 *             16 a workgroup barrier before the stores; with the plain maps' copy variant: 32 = lane pairs move their
 *             two blocks as dwordx4 rows (the access pattern of a 16-byte-per-lane kernel)
 *   rows=N    pipe map: block rows per workgroup (default 4)
 *   lds=N     N bytes of unused dynamic LDS per workgroup of the plain 8-bit luma kernel and its copy variant: limits the
 *             workgroups per CU (160 KiB / N) for occupancy A/B runs
 *   xpad=N    row-major map of the packed kernels: N padding workgroups behind every XCD's range, so that the eight XCDs do not
 *             walk through the batch at offsets that are equal modulo a power of two (placement experiments)
 *   nostrong | nonormal | barriers   luma ablations -- WRONG PIXELS, timing only
 * returns HEVCDBK_OK or HEVCDBK_ERR_ARG (unknown knob; nothing changed) */
HEVCDBK_API int hevcdbk_diag_set(const char *spec);

#ifdef __cplusplus
}
#endif
