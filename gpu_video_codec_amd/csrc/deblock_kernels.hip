/*
 * deblock_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4, wave64).
 *
 * Mapping shared by all kernels: one lane owns one offset 8x8 block (SURVEY 8 layout), a wave
 * owns 64 consecutive blocks of one block row, so every row load of a wave is one contiguous
 * 64 x 8 B (8-bit) / 64 x 16 B (16-bit) span of the un-padded plane.  The reference's 4-sample
 * zero padding (cpu.h:55-71) is virtual: out-of-image halves/rows are materialised as zeros in
 * registers and never stored.  Blocks are mutually independent, so src == dst
 * is race-free.
 *
 * No MFMA: this is byte/int16 arithmetic on a streaming in-place filter; the bound is HBM.
 */
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "deblock_core.h"
#include "deblock_kernels.h"
#include "deblock_packed.h"
#include "deblock_packed_h265.h"
#include "deblock_packed16.h"
#include "sao_packed.h"

/*
 * Every launch goes through hipExtLaunchKernelGGL so that a caller can ask for the NEXT launch to stamp its own start and
 * stop into two events (the dispatch's begin / end timestamps -- what a profiler's kernel trace reports -- with no barrier
 * packets around the kernel).  One-shot and per host thread; without a request the events are NULL = an ordinary launch.
 */
static thread_local hipEvent_t t_next_start = nullptr, t_next_stop = nullptr;
void dbk_set_next_launch_events(hipEvent_t start, hipEvent_t stop)
{
    t_next_start = start;
    t_next_stop = stop;
}
#define DBK_LAUNCH_LDS(kernel, grid, block, lds_bytes, stream, ...)                               \
    do {                                                                                           \
        hipEvent_t s_ = t_next_start, e_ = t_next_stop;                                            \
        t_next_start = t_next_stop = nullptr;                                                      \
        hipExtLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, s_, e_, 0, __VA_ARGS__);     \
    } while (0)
#define DBK_LAUNCH(kernel, grid, block, stream, ...) DBK_LAUNCH_LDS(kernel, grid, block, 0, stream, __VA_ARGS__)

namespace {

/* ------------------------------------------------------------------------------------------ */
/* generic kernel: 32-bit arithmetic, uint8 / uint16 samples, luma / chroma, scalar QP / map   */

template <typename T>
struct Quad; /* 4 consecutive samples as one memory word */
template <>
struct Quad<uint8_t> {
    using W = uint32_t;
    static __device__ __forceinline__ void unpack(W w, int &a, int &b, int &c, int &d)
    {
        a = w & 0xff; b = (w >> 8) & 0xff; c = (w >> 16) & 0xff; d = w >> 24;
    }
    static __device__ __forceinline__ W pack(int a, int b, int c, int d)
    {
        return (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24);
    }
    static __device__ __forceinline__ W zero() { return 0u; }
};
template <>
struct Quad<uint16_t> {
    using W = uint2;
    static __device__ __forceinline__ void unpack(W w, int &a, int &b, int &c, int &d)
    {
        a = w.x & 0xffff; b = w.x >> 16; c = w.y & 0xffff; d = w.y >> 16;
    }
    static __device__ __forceinline__ W pack(int a, int b, int c, int d)
    {
        return make_uint2((uint32_t)a | ((uint32_t)b << 16), (uint32_t)c | ((uint32_t)d << 16));
    }
    static __device__ __forceinline__ W zero() { return make_uint2(0u, 0u); }
};

template <typename T, bool CHROMA, bool QPMAP>
__global__ __launch_bounds__(256) void dbk_generic_kernel(const DbkArgs a)
{
    using Q = Quad<T>;
    using W = typename Q::W;
    const int bx = blockIdx.x * 64 + threadIdx.x;
    const int by = blockIdx.y * 4 + threadIdx.y + a.by_begin;
    const int f = blockIdx.z;
    if (bx >= a.nbx || by >= (a.by_count ? a.by_begin + a.by_count : a.nby)) return;

    const uint8_t *src = a.src + (long long)f * a.frame_stride;
    uint8_t *dst = a.dst + (long long)f * a.frame_stride;
    const int x0 = bx * 8 - 4, y0 = by * 8 - 4; /* image coords of the block's (0,0) */
    const bool lv = bx > 0;                     /* cols 0..3 inside the image */
    const bool rv = bx < a.nbx - 1;             /* cols 4..7 inside the image */

    int v[8][8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int y = y0 + r;
        const bool rowv = (unsigned)y < (unsigned)a.plane_h;
        const uint8_t *row = src + (long long)y * a.pitch + (long long)x0 * (int)sizeof(T);
        W l = Q::zero(), rr = Q::zero();
        if (rowv && lv) l = *reinterpret_cast<const W *>(row);
        if (rowv && rv) rr = *reinterpret_cast<const W *>(row + 4 * sizeof(T));
        Q::unpack(l, v[r][0], v[r][1], v[r][2], v[r][3]);
        Q::unpack(rr, v[r][4], v[r][5], v[r][6], v[r][7]);
    }

    const dbk::BlockBs bs =
        dbk::load_block_bs(a.vert_bs + (long long)f * a.vert_bs_stride, a.hor_bs + (long long)f * a.hor_bs_stride,
                           bx, by, a.vstride, a.hstride, a.limit_bx, a.limit_by, a.n_vert, a.n_hor);

    dbk::BlockQp q;
    if constexpr (QPMAP) {
        const uint8_t *map = a.qp_map + (long long)f * a.map_frame_stride;
        const int sc = CHROMA ? 2 : 1, lw = a.plane_w * sc, lh = a.plane_h * sc;
        /* P0 / Q0 of line 0 of each segment, plane coords (same positions as oracle seg_tc_beta) */
        const int qp0 = dbk::seg_qp_from_map(map, a.map_stride, a.ctu_log2, sc, lw, lh, x0 + 3, y0 + 0, x0 + 4, y0 + 0);
        const int qp1 = dbk::seg_qp_from_map(map, a.map_stride, a.ctu_log2, sc, lw, lh, x0 + 3, y0 + 4, x0 + 4, y0 + 4);
        const int qp2 = dbk::seg_qp_from_map(map, a.map_stride, a.ctu_log2, sc, lw, lh, x0 + 0, y0 + 3, x0 + 0, y0 + 4);
        const int qp3 = dbk::seg_qp_from_map(map, a.map_stride, a.ctu_log2, sc, lw, lh, x0 + 4, y0 + 3, x0 + 0, y0 + 4);
        q.tc[0] = a.tc_tab[qp0] << a.shift; q.beta[0] = a.beta_tab[qp0] << a.shift;
        q.tc[1] = a.tc_tab[qp1] << a.shift; q.beta[1] = a.beta_tab[qp1] << a.shift;
        q.tc[2] = a.tc_tab[qp2] << a.shift; q.beta[2] = a.beta_tab[qp2] << a.shift;
        q.tc[3] = a.tc_tab[qp3] << a.shift; q.beta[3] = a.beta_tab[qp3] << a.shift;
    } else {
#pragma unroll
        for (int s = 0; s < 4; s++) { q.tc[s] = a.tc; q.beta[s] = a.beta; }
    }

    dbk::filter_block_generic<CHROMA>(v, bs, q, a.max_v);

#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int y = y0 + r;
        const bool rowv = (unsigned)y < (unsigned)a.plane_h;
        uint8_t *row = dst + (long long)y * a.pitch + (long long)x0 * (int)sizeof(T);
        if (rowv && lv) *reinterpret_cast<W *>(row) = Q::pack(v[r][0], v[r][1], v[r][2], v[r][3]);
        if (rowv && rv) *reinterpret_cast<W *>(row + 4 * sizeof(T)) = Q::pack(v[r][4], v[r][5], v[r][6], v[r][7]);
    }
}

template <typename T, bool CHROMA>
hipError_t launch_generic_t(const DbkArgs &a, hipStream_t stream)
{
    dim3 block(64, 4, 1);
    dim3 grid((a.nbx + 63) / 64, ((a.by_count ? a.by_count : a.nby) + 3) / 4, a.n_frames);
    if (a.qp_map)
        DBK_LAUNCH((dbk_generic_kernel<T, CHROMA, true>), grid, block, stream, a);
    else
        DBK_LAUNCH((dbk_generic_kernel<T, CHROMA, false>), grid, block, stream, a);
    return hipGetLastError();
}

} /* namespace */

hipError_t dbk_launch_generic(const DbkArgs &a, int sample_bytes, bool chroma, hipStream_t stream)
{
    if (a.n_frames <= 0 || a.nbx <= 0 || a.nby <= 0) return hipSuccess;
    if (sample_bytes == 1)
        return chroma ? launch_generic_t<uint8_t, true>(a, stream) : launch_generic_t<uint8_t, false>(a, stream);
    return chroma ? launch_generic_t<uint16_t, true>(a, stream) : launch_generic_t<uint16_t, false>(a, stream);
}

/* ------------------------------------------------------------------------------------------ */
/* packed kernel: 8-bit samples, scalar QP, v_pk_*_i16 arithmetic (deblock_packed.h)            */

namespace {

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

/* cache-policy bits of the raw buffer builtins on gfx950: bit 1 = nt (streamed, evict first) */
#ifndef DBK_AUX_LD /* experiment builds only (tools/exp): cache-policy bits of the pixel loads / stores */
#define DBK_AUX_LD 0
#endif
#ifndef DBK_AUX_ST
#define DBK_AUX_ST 0
#endif
template <bool NT> constexpr int aux_ld() { return NT ? 2 : DBK_AUX_LD; }
template <bool NT> constexpr int aux_st() { return NT ? 2 : DBK_AUX_ST; }
constexpr uint32_t kOob = 0xfffffff0u; /* voffset >= num_records: load returns 0, store is dropped */

/*
 * The four bS bytes of a lane's offset block (cpu.h:159-163, 223-227, 287-291, 368-372 and the chroma twins),
 * as four back-to-back buffer_load_ubyte with NO wait in between.  The guards of the reference fold into
 * the buffer range check wherever the failing index is out of the array: ver1 at by == 0 has a negative
 * index, ver2 at by == nby-1 (and the chroma over-read) has index >= n_vert, hor2 past the array
 * likewise -- the hardware returns 0 there.  Only "bx > 0" (hor1) and "bx < limit_bx" (hor2) can fail on an
 * in-range index and need a select; interior waves (EDGE == false) satisfy both by construction.
 */
template <bool EDGE>
__device__ __forceinline__ dbk::BlockBs load_bs_buffer(const DbkArgs &a, int f, int by, int bx, bool active)
{
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(a.vert_bs) + (long long)f * a.vert_bs_stride, 0, (uint32_t)a.n_vert, 0x00020000);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(a.hor_bs) + (long long)f * a.hor_bs_stride, 0, (uint32_t)a.n_hor, 0x00020000);
    dbk::BlockBs b;
    if constexpr (!EDGE) { /* by wave-uniform: scalar row offsets */
        b.ver1 = __builtin_amdgcn_raw_buffer_load_b8(rv, (uint32_t)bx, (by - 1) * a.vstride, 0);
        b.ver2 = __builtin_amdgcn_raw_buffer_load_b8(rv, (uint32_t)bx, by * a.vstride, 0);
        b.hor1 = __builtin_amdgcn_raw_buffer_load_b8(rh, (uint32_t)(bx - 1), by * a.hstride, 0);
        b.hor2 = __builtin_amdgcn_raw_buffer_load_b8(rh, (uint32_t)bx, by * a.hstride, 0);
    } else {
        const uint32_t iv = (uint32_t)(by * a.vstride + bx), ih = (uint32_t)(by * a.hstride + bx);
        b.ver1 = __builtin_amdgcn_raw_buffer_load_b8(rv, active ? iv - (uint32_t)a.vstride : kOob, 0, 0);
        b.ver2 = __builtin_amdgcn_raw_buffer_load_b8(rv, active ? iv : kOob, 0, 0);
        b.hor1 = __builtin_amdgcn_raw_buffer_load_b8(rh, (active && bx > 0) ? ih - 1u : kOob, 0, 0);
        b.hor2 = __builtin_amdgcn_raw_buffer_load_b8(rh, (active && bx < a.limit_bx) ? ih : kOob, 0, 0);
    }
    return b;
}

/* PATH 3 of packed_body (first / last wave of a block row in the row map, 1 <= by <= nby-2 wave-uniform, every lane
 * active): the scalar-row form of the interior waves; "bx > 0" (hor1) folds into the range check (index -1), "bx <
 * limit_bx" (hor2) is the one select left */
__device__ __forceinline__ dbk::BlockBs load_bs_buffer_rowedge(const DbkArgs &a, int f, int by, int bx)
{
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(a.vert_bs) + (long long)f * a.vert_bs_stride, 0, (uint32_t)a.n_vert, 0x00020000);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(a.hor_bs) + (long long)f * a.hor_bs_stride, 0, (uint32_t)a.n_hor, 0x00020000);
    dbk::BlockBs b;
    b.ver1 = __builtin_amdgcn_raw_buffer_load_b8(rv, (uint32_t)bx, (by - 1) * a.vstride, 0);
    b.ver2 = __builtin_amdgcn_raw_buffer_load_b8(rv, (uint32_t)bx, by * a.vstride, 0);
    b.hor1 = __builtin_amdgcn_raw_buffer_load_b8(rh, (uint32_t)(bx - 1), by * a.hstride, 0); /* bx == 0: 0xffffffff, out of range */
    b.hor2 = __builtin_amdgcn_raw_buffer_load_b8(rh, bx < a.limit_bx ? (uint32_t)bx : kOob, by * a.hstride, 0);
    return b;
}

/* spec-exact mode: the four 4-sample-granular bS bytes of a lane's block (deblock_h265.h load_block_bs_h265); edges on
 * the picture boundary and halves outside the picture read 0 through an out-of-range offset */
template <int PATH>
__device__ __forceinline__ void load_bs_buffer_h265(const DbkArgs &a, int f, int by, int bx, bool active, int (&entry)[4])
{
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(a.vert_bs) + (long long)f * a.vert_bs_stride, 0, (uint32_t)a.n_vert, 0x00020000);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(a.hor_bs) + (long long)f * a.hor_bs_stride, 0, (uint32_t)a.n_hor, 0x00020000);
    if constexpr (PATH == 0) { /* interior wave: by scalar, 1 <= by <= nby-2, 1 <= bx <= nbx-2 */
        entry[0] = __builtin_amdgcn_raw_buffer_load_b8(rv, (uint32_t)bx, (2 * by - 1) * a.vstride, 0);
        entry[1] = __builtin_amdgcn_raw_buffer_load_b8(rv, (uint32_t)bx, 2 * by * a.vstride, 0);
        entry[2] = __builtin_amdgcn_raw_buffer_load_b8(rh, (uint32_t)(2 * bx - 1), by * a.hstride, 0);
        entry[3] = __builtin_amdgcn_raw_buffer_load_b8(rh, (uint32_t)(2 * bx), by * a.hstride, 0);
    } else {
        const bool vedge = active && bx > 0 && bx < a.nbx - 1, hedge = active && by > 0 && by < a.nby - 1;
        const uint32_t iv = (uint32_t)(2 * by * a.vstride + bx), ih = (uint32_t)(by * a.hstride + 2 * bx);
        entry[0] = __builtin_amdgcn_raw_buffer_load_b8(rv, (vedge && by > 0) ? iv - (uint32_t)a.vstride : kOob, 0, 0);
        entry[1] = __builtin_amdgcn_raw_buffer_load_b8(rv, (vedge && by < a.nby - 1) ? iv : kOob, 0, 0);
        entry[2] = __builtin_amdgcn_raw_buffer_load_b8(rh, (hedge && bx > 0) ? ih - 1u : kOob, 0, 0);
        entry[3] = __builtin_amdgcn_raw_buffer_load_b8(rh, (hedge && bx < a.nbx - 1) ? ih : kOob, 0, 0);
    }
}

/* per-segment tc / beta of a lane's block: the scalar-QP values, or -- QPMAP -- looked up from the per-CTU
 * map exactly as the generic kernel and the oracle do (QP = (QpP + QpQ + 1) >> 1 of the CTUs holding P0 / Q0
 * of the segment's first line) */
template <bool CHROMA>
__device__ __forceinline__ void block_unit_qps_dev(const DbkArgs &a, int f, int by, int bx, int (&u)[4]);
template <bool QPMAP, bool CHROMA>
__device__ __forceinline__ dbk::BlockQp block_qp(const DbkArgs &a, int f, int by, int bx)
{
    dbk::BlockQp q;
    if constexpr (QPMAP) {
        /* the generic kernel's eight look-ups (seg_qp_from_map per segment) are four map units: above-left, above-right,
         * below-left, below-right of the block's centre (block_unit_qps); hor2 pairs above-right with below-LEFT (cpu.h:383-414) */
        int u[4];
        block_unit_qps_dev<CHROMA>(a, f, by, bx, u);
        const int qp0 = dbk::seg_qp_avg(u[0], u[1]), qp1 = dbk::seg_qp_avg(u[2], u[3]);
        const int qp2 = dbk::seg_qp_avg(u[0], u[2]), qp3 = dbk::seg_qp_avg(u[1], u[2]);
        q.tc[0] = a.tc_tab[qp0] << a.shift; q.beta[0] = a.beta_tab[qp0] << a.shift;
        q.tc[1] = a.tc_tab[qp1] << a.shift; q.beta[1] = a.beta_tab[qp1] << a.shift;
        q.tc[2] = a.tc_tab[qp2] << a.shift; q.beta[2] = a.beta_tab[qp2] << a.shift;
        q.tc[3] = a.tc_tab[qp3] << a.shift; q.beta[3] = a.beta_tab[qp3] << a.shift;
    } else {
#pragma unroll
        for (int s = 0; s < 4; s++) { q.tc[s] = a.tc; q.beta[s] = a.beta; }
    }
    return q;
}

/* QP-map launches, luma (round 4): the segments' QPs as INDICES into the workgroup's operand table instead of tc / beta values
 * (reference-exact mode: one index serves both halves of the table) */
/* The four map units of a lane's block (dbk::block_unit_qps, deblock_core.h: above-left, above-right, below-left, below-right of
 * the block's centre) through a buffer resource: 32-bit offsets from ONE 24-bit multiply-add (map rows and strides are far below
 * 2^24: the entry points check the stride) instead of two 32 x 32-bit multiplies -- quarter rate on this hardware -- and four
 * 64-bit address sums.  Same clamping, same values. */
template <bool CHROMA>
__device__ __forceinline__ void block_unit_qps_dev(const DbkArgs &a, int f, int by, int bx, int (&u)[4])
{
    const int sc = CHROMA ? 2 : 1, lw = a.plane_w * sc, lh = a.plane_h * sc;
    const int x0 = bx * 8 - 4, y0 = by * 8 - 4;
    const int xl = dbk::clampi((x0 + 3) * sc, 0, lw - 1) >> a.ctu_log2, xr = dbk::clampi((x0 + 4) * sc, 0, lw - 1) >> a.ctu_log2;
    const int yt = dbk::clampi((y0 + 3) * sc, 0, lh - 1) >> a.ctu_log2, yb = dbk::clampi((y0 + 4) * sc, 0, lh - 1) >> a.ctu_log2;
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(a.qp_map) + (long long)f * a.map_frame_stride, 0, 0x80000000u, 0x00020000);
    const uint32_t ot = __umul24((uint32_t)yt, (uint32_t)a.map_stride);
    const uint32_t ob = yb != yt ? ot + (uint32_t)a.map_stride : ot; /* the lower unit row is the same one or the next */
    u[0] = __builtin_amdgcn_raw_buffer_load_b8(rm, ot + (uint32_t)xl, 0, 0);
    u[1] = __builtin_amdgcn_raw_buffer_load_b8(rm, ot + (uint32_t)xr, 0, 0);
    u[2] = __builtin_amdgcn_raw_buffer_load_b8(rm, ob + (uint32_t)xl, 0, 0);
    u[3] = __builtin_amdgcn_raw_buffer_load_b8(rm, ob + (uint32_t)xr, 0, 0);
}
/* the spec-exact mode's four segment QPs from them (dbk::h265_block_qpl4) */
template <bool CHROMA>
__device__ __forceinline__ void block_qpl4_dev(const DbkArgs &a, int f, int by, int bx, int (&qpl)[4])
{
    int q[4];
    block_unit_qps_dev<CHROMA>(a, f, by, bx, q);
    qpl[0] = dbk::seg_qp_avg(q[0], q[1]); /* ver1: above-left | above-right */
    qpl[1] = dbk::seg_qp_avg(q[2], q[3]); /* ver2: below-left | below-right */
    qpl[2] = dbk::seg_qp_avg(q[0], q[2]); /* hor1: above-left / below-left */
    qpl[3] = dbk::seg_qp_avg(q[1], q[3]); /* hor2: above-right / below-right */
}

template <bool CHROMA>
__device__ __forceinline__ dbk::QsTable block_qp_tab(const DbkArgs &a, int f, int by, int bx, const DBK_LDS uint32_t *tab)
{
    int u[4];
    block_unit_qps_dev<CHROMA>(a, f, by, bx, u);
    dbk::QsTable q;
    q.tab = tab;
    q.ib[0] = q.it[0] = dbk::seg_qp_avg(u[0], u[1]);
    q.ib[1] = q.it[1] = dbk::seg_qp_avg(u[2], u[3]);
    q.ib[2] = q.it[2] = dbk::seg_qp_avg(u[0], u[2]);
    q.ib[3] = q.it[3] = dbk::seg_qp_avg(u[1], u[2]); /* hor2 pairs above-right with below-LEFT (cpu.h:383-414) */
    return q;
}

/* The operand table of a reference-exact QP-map luma launch (deblock_packed.h, ktab_build): every lane of the workgroup fills
 * its share of the 54 rows from the launch's tc / beta tables, one barrier, and the table is there for all waves.  Called at
 * the very top of a kernel, before any wave can leave. */
__device__ __forceinline__ const DBK_LDS uint32_t *ktab_setup(uint32_t *lds, const DbkArgs &a)
{
    dbk::ktab_build<false>(lds, (int)threadIdx.x, (int)blockDim.x,
                           [&](int i) { return i < 52 ? (int)a.beta_tab[i] << a.shift : 0; },
                           [&](int i) { return i < 52 ? (int)a.tc_tab[i] << a.shift : 0; });
    __syncthreads();
    return (const DBK_LDS uint32_t *)lds;
}

/* the same for the spec-exact mode: rows by Table 8-12's two indices (beta: 0..51, tc: 0..53), the standard's thresholds */
__device__ __forceinline__ const DBK_LDS uint32_t *ktab_setup_h265(uint32_t *lds, int shift)
{
    dbk::ktab_build<true>(lds, (int)threadIdx.x, (int)blockDim.x,
                          [&](int i) { return dbk::h265_beta(i < 52 ? i : 51) << shift; },
                          [&](int i) { return dbk::h265_tc(i) << shift; });
    __syncthreads();
    return (const DBK_LDS uint32_t *)lds;
}

/*
 * Body of the packed kernel for one lane (= one offset block).
 *
 * EDGE == false: interior wave -- all 64 lanes own both halves of all 8 rows.  The 8 row loads are
 *   8 back-to-back buffer_load_dwordx2 with a scalar row offset (no per-row branch, no VALU address
 *   math), so a wave has its whole 4 KB tile in flight before the first s_waitcnt.
 * EDGE == true: frame-edge wave (first/last block row, first/last wave of a row).  Out-of-image
 *   halves use an out-of-range buffer offset: the hardware returns 0 for the load (= the
 *   reference's zero padding, cpu.h:55-71) and drops the store.  Out-of-image rows are skipped
 *   with wave-uniform branches.
 * MODE 0 = filter, MODE 1 = diagnostic copy (same loads/stores, no arithmetic).
 */
template <bool CHROMA, int MODE, bool NT, int PATH, bool QPMAP>
__device__ __forceinline__ void packed_body(const DbkArgs &a, int by, int f, int bx, bool active, int by0,
                                            const DbkH265Args *hx = nullptr /* MODE 2 only */,
                                            uint32_t *ktab_lds = nullptr /* QPMAP luma: LDS for the workgroup's operand table */)
{
    /* QP-map luma launches: every wave of the workgroup passes ktab_setup() exactly once -- AFTER its row, bS and map loads have
     * been issued, so that the table's construction (the first 54 lanes of the workgroup; the others wait at the barrier)
     * overlaps the latency of loads the wave has to wait for anyway.  PATH 3 lets idle lanes leave first thing, and a lane that
     * has left builds no row: there the table comes first. */
    constexpr bool KT = QPMAP && !CHROMA && (MODE == 0 || MODE == 2 || MODE == 3);
    const DBK_LDS uint32_t *ktab = nullptr;
    if constexpr (KT && PATH == 3) ktab = MODE == 2 ? ktab_setup_h265(ktab_lds, 0) : ktab_setup(ktab_lds, a);
    /* PATH 0: interior wave, by == by0 wave-uniform, every lane owns both halves of all 8 rows.
     * PATH 1: every lane's 8 rows are inside the image, but lanes may sit in different block rows (row-major
     *         map) and the wave may hold frame-edge blocks (bx == 0 / nbx-1) or idle lanes: still ONE 8-byte
     *         access per row with a scalar row offset -- the lane's extra rows go into its vector offset, an
     *         out-of-image half is zeroed after the load (the bytes fetched there belong to the neighbouring
     *         image row) and is stored by nobody (split store: full lanes 8 bytes, edge lanes 4).
     * PATH 2: first / last block row: per-lane 4-byte accesses, out-of-image rows and halves via
     *         out-of-range offsets.
     * PATH 3: (round 3) the first / last wave of a block row in the row map: by wave-uniform like PATH 0, but the wave
     *         holds the bx == 0 block (left half outside the image), the bx == nbx-1 block (right half outside) and / or
     *         idle lanes.  Idle lanes leave at once; everybody else runs PATH 0's memory code against a buffer resource
     *         whose base lies 4 bytes earlier, so that the bx == 0 lane's 8-byte access has offset 0 instead of -4 (it
     *         fetches the last 4 bytes of the row above and its own right half); the one or two edge lanes of the wave get
     *         their outside half zeroed under a wave-uniform guard and store their inside half with a 4-byte access, all
     *         other lanes with the 8-byte one.  About 25 instructions more than PATH 0 where PATH 1 costs 60: two of the
     *         eight waves of a 4K block row are such waves. */
    if constexpr (PATH == 3) {
        if (!active) return;
    }
#ifdef HEVCDBK_DIAG
    /* copy variant, knob "align": every lane moves the naturally aligned 8 bytes 8*bx .. 8*bx+7 (the last block of a row has
     * none), so no 128-byte line is shared between two waves: the seam-free ceiling of the row map (timing only) */
    const bool shifted = MODE == 1 && a.diag_xshift != 0;
    if (shifted) active = active && bx < a.nbx - 1;
    const bool lv = shifted ? active : (active && bx > 0);
    const bool rv = shifted ? active : (active && bx < a.nbx - 1);
#else
    const bool lv = active && bx > 0;            /* cols 0..3 inside the image */
    const bool rv = active && bx < a.nbx - 1;    /* cols 4..7 inside the image */
#endif
    const int y0 = by * 8 - 4;
#ifdef HEVCDBK_DIAG
    const uint32_t xoff = (uint32_t)(bx * 8 - 4) + (MODE == 1 ? (uint32_t)a.diag_xshift : 0u);
#else
    const uint32_t xoff = (uint32_t)(bx * 8 - 4);
#endif
    const uint32_t plane_bytes = (uint32_t)a.pitch * (uint32_t)a.plane_h;

    constexpr int kBias = PATH == 3 ? 4 : 0; /* PATH 3: the resource starts 4 bytes before the plane, offsets grow by 4 */
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(a.src) + (long long)f * a.frame_stride - kBias, 0, plane_bytes + kBias, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        a.dst + (long long)f * a.frame_stride - kBias, 0, plane_bytes + kBias, 0x00020000);

    uint32_t L[8], R[8];
#ifdef HEVCDBK_DIAG
    if constexpr (MODE == 3) {
        if (a.diag_prio & 1) __builtin_amdgcn_s_setprio(3);
    }
#endif
#ifdef HEVCDBK_DIAG
    if constexpr (PATH == 0 && MODE == 1) {
        if (a.diag_dummy & 32) {
            /* copy diagnostic, knob dummy=32: the memory side of "lane pairs move 16 bytes" -- lanes 2k / 2k+1 load and store
             * rows 0-3 / 4-7 of their two adjacent blocks as dwordx4 (no exchange: it is a copy); prices the access pattern
             * a 16-byte-per-lane filter kernel would have, whose lane exchange would cost about 64 VALU instructions */
            const uint32_t odd = (uint32_t)bx & 1u; /* interior waves start at an even bx: pairs (2k, 2k+1) */
            const uint32_t voff = (uint32_t)((bx & ~1) * 8 - 4) + (odd ? 4u * (uint32_t)a.pitch : 0u);
            u32x4 w[4];
#pragma unroll
            for (int i = 0; i < 4; i++) w[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (y0 + i) * (int)a.pitch, 0);
#pragma unroll
            for (int i = 0; i < 4; i++) __builtin_amdgcn_raw_buffer_store_b128(w[i], rd, voff, (y0 + i) * (int)a.pitch, 0);
            return;
        }
    }
#endif
    if constexpr (PATH == 0) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, xoff, (y0 + r) * (int)a.pitch, aux_ld<NT>());
            L[r] = w.x;
            R[r] = w.y;
        }
    } else if constexpr (PATH == 3) {
        const uint32_t xoff3 = (uint32_t)(bx * 8); /* = xoff + kBias */
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, xoff3, (y0 + r) * (int)a.pitch, aux_ld<NT>());
            L[r] = w.x;
            R[r] = w.y;
        }
        /* the reference's zero padding (cpu.h:55-71) for the one lane of the wave whose half lies outside the image */
        if (__builtin_amdgcn_ballot_w64(!lv) != 0ull) {
#pragma unroll
            for (int r = 0; r < 8; r++) L[r] = lv ? L[r] : 0u;
        }
        if (__builtin_amdgcn_ballot_w64(!rv) != 0ull) {
#pragma unroll
            for (int r = 0; r < 8; r++) R[r] = rv ? R[r] : 0u;
        }
    } else if constexpr (PATH == 1) {
        /* the buffer range check looks at the vector offset alone, so it must not go negative: a bx == 0 lane
         * (xoff = -4) fetches its 8 bytes one half further right and takes its right half from the first dword */
        const uint32_t vrow = (uint32_t)((by - by0) * 8) * (uint32_t)a.pitch;
        const uint32_t voff = active ? (lv ? xoff + vrow : xoff + 4u + vrow) : kOob;
        const int y0s = by0 * 8 - 4; /* scalar */
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, (y0s + r) * (int)a.pitch, aux_ld<NT>());
            L[r] = lv ? w.x : 0u;
            R[r] = rv ? (lv ? w.y : w.x) : 0u;
        }
    } else {
        /* straight-line, per-lane offsets: an out-of-image half or row gets an out-of-range offset (load -> 0).  The offset is
         * pushed out of range by OR-ing kOob into it -- a per-lane word for the halves, a scalar word per row -- instead of
         * selecting under sixteen 64-bit lane masks: those masks made this path, which runs in two block rows of a frame, the
         * SGPR peak of the whole kernel */
        const uint32_t base = (uint32_t)(y0 * (int)a.pitch) + xoff;
        const uint32_t lbits = lv ? 0u : kOob, rbits = rv ? 0u : kOob;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t ob = (unsigned)(y0 + r) < (unsigned)a.plane_h ? 0u : kOob; /* scalar */
            const uint32_t off = base + (uint32_t)r * (uint32_t)a.pitch;
            L[r] = __builtin_amdgcn_raw_buffer_load_b32(rs, off | lbits | ob, 0, aux_ld<NT>());
            R[r] = __builtin_amdgcn_raw_buffer_load_b32(rs, (off + 4u) | rbits | ob, 0, aux_ld<NT>());
        }
    }

    if constexpr (MODE == 0 || MODE == 3) { /* 3 = the filter with the timing-only ablation switches compiled in */
        const dbk::BlockBs bs = PATH == 3 ? load_bs_buffer_rowedge(a, f, by, bx) : load_bs_buffer<(PATH != 0)>(a, f, by, bx, active);
        if constexpr (QPMAP && !CHROMA) {
            dbk::QsTable qs = block_qp_tab<CHROMA>(a, f, by, active ? bx : 0, ktab);
            if constexpr (PATH != 3) qs.tab = ktab_setup(ktab_lds, a);
#ifdef HEVCDBK_DIAG
            dbk::packed_filter_luma_block_src(L, R, bs, qs, MODE == 3 ? a.diag_ablate : 0);
#else
            dbk::packed_filter_luma_block_src(L, R, bs, qs);
#endif
        } else {
        const dbk::BlockQp q = block_qp<QPMAP, CHROMA>(a, f, by, active ? bx : 0);
#ifdef HEVCDBK_DIAG
        if constexpr (MODE == 3) {
            if (a.diag_prio & 1) __builtin_amdgcn_s_setprio(0);
            uint32_t sink = L[0];
            for (int i = 0; i < a.diag_dummy; i++) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(sink) : "v"(R[0]));
            if (a.diag_dummy && sink == 0x12345678u) L[0] ^= 1u; /* keeps the chain alive; never true for packed samples */
        }
        dbk::packed_filter_block<CHROMA, !QPMAP>(L, R, bs, q, MODE == 3 ? a.diag_ablate : 0);
        if constexpr (MODE == 3) {
            if (a.diag_prio & 2) __builtin_amdgcn_s_setprio(3);
        }
#else
        dbk::packed_filter_block<CHROMA, !QPMAP>(L, R, bs, q);
#endif
        }
    }
#ifdef HEVCDBK_DIAG
    else if constexpr (MODE == 1) {
        /* copy variant, knob dummy=128: the four bS byte loads of the filter ride along (what they cost the memory path) */
        if (a.diag_dummy & 128) {
            const dbk::BlockBs bs = load_bs_buffer<PATH != 0>(a, f, by, bx, active);
            if (bs.ver1 + bs.ver2 + bs.hor1 + bs.hor2 == 0x7fffffff) L[0] ^= 1u; /* never true: keeps the loads alive */
        }
    }
#endif
    else if constexpr (MODE == 2) { /* spec-exact mode, H.265 8.7.2 */
        int entry[4];
        load_bs_buffer_h265<PATH>(a, f, by, bx, active, entry);
        dbk::H265Seg sg;
        if constexpr (QPMAP) {
            int qpl[4];
            block_qpl4_dev<CHROMA>(a, f, by, active ? bx : 0, qpl);
            const dbk::H265Prm prm = {hx->tc_off, hx->beta_off, hx->c_qp_offset, 0, 255};
            if constexpr (KT) {
                if constexpr (PATH != 3) ktab = ktab_setup_h265(ktab_lds, 0);
                dbk::h265_seg_rows(entry, qpl, prm, ktab, sg);
            } else {
                dbk::h265_seg_params<CHROMA>(entry, qpl, prm, sg);
            }
        } else { /* one QP: beta is a scalar and tc one of two scalars picked by the bS */
#pragma unroll
            for (int i = 0; i < 4; i++) {
                sg.entry[i] = entry[i];
                sg.beta[i] = hx->beta_s;
                sg.tc[i] = (entry[i] & dbk::kH265BsMask) == 2 ? hx->tc_bs2 : hx->tc_bs1;
            }
        }
        if constexpr (!QPMAP && !CHROMA) { /* the segment constants of both bS values, once per block, on the scalar unit */
            const dbk::H265Uni u = dbk::h265_uni(hx->beta_s, hx->tc_bs1, hx->tc_bs2);
            dbk::packed_filter_block_h265<CHROMA>(L, R, sg, &u);
        } else {
            dbk::packed_filter_block_h265<CHROMA, KT>(L, R, sg);
        }
    }

    /* the eight scalar row offsets are built AGAIN for the stores (the empty asm hides from the compiler that they equal the
     * loads' offsets): carried across the filter they cost 8 SGPRs, and the kernel's SGPR allocation decides how many waves
     * really fit a SIMD -- 800 SGPRs per SIMD in blocks of 16: at 80 per wave ten waves' worth fit and eight always find room,
     * at 96 only 8.3 do and a finished wave's block is not always where the next wave's fits (round 3: the same
     * instructions at 93 instead of 78 SGPRs ran 1 % slower with 17 % fewer waves resident on average) */
    int spitch = __builtin_amdgcn_readfirstlane((int)a.pitch); /* wave-uniform by construction; says so to the compiler */
    asm volatile("" : "+s"(spitch));
    if constexpr (PATH == 0) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            u32x2 w;
            w.x = L[r];
            w.y = R[r];
            __builtin_amdgcn_raw_buffer_store_b64(w, rd, xoff, (y0 + r) * spitch, aux_st<NT>());
        }
    } else if constexpr (PATH == 3) {
        const uint32_t xoff3 = (uint32_t)(bx * 8);
        const uint32_t vfull = (lv && rv) ? xoff3 : kOob; /* both halves in the image: the 8-byte store */
#pragma unroll
        for (int r = 0; r < 8; r++) {
            u32x2 w;
            w.x = L[r];
            w.y = R[r];
            __builtin_amdgcn_raw_buffer_store_b64(w, rd, vfull, (y0 + r) * spitch, aux_st<NT>());
        }
        if (__builtin_amdgcn_ballot_w64(!lv) != 0ull) { /* the bx == 0 lane: its right half */
            const uint32_t vh = (!lv && rv) ? xoff3 + 4u : kOob;
#pragma unroll
            for (int r = 0; r < 8; r++) __builtin_amdgcn_raw_buffer_store_b32(R[r], rd, vh, (y0 + r) * spitch, aux_st<NT>());
        }
        if (__builtin_amdgcn_ballot_w64(!rv) != 0ull) { /* the bx == nbx-1 lane: its left half */
            const uint32_t vh = (lv && !rv) ? xoff3 : kOob;
#pragma unroll
            for (int r = 0; r < 8; r++) __builtin_amdgcn_raw_buffer_store_b32(L[r], rd, vh, (y0 + r) * spitch, aux_st<NT>());
        }
    } else if constexpr (PATH == 1) {
        const uint32_t voff = xoff + (uint32_t)((by - by0) * 8) * (uint32_t)a.pitch;
        const uint32_t vfull = (lv && rv) ? voff : kOob;                               /* both halves in the image */
        const uint32_t vhalf = (lv && !rv) ? voff : ((rv && !lv) ? voff + 4u : kOob);  /* exactly one half */
        const int y0s = by0 * 8 - 4;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            u32x2 w;
            w.x = L[r];
            w.y = R[r];
            __builtin_amdgcn_raw_buffer_store_b64(w, rd, vfull, (y0s + r) * (int)a.pitch, aux_st<NT>());
            __builtin_amdgcn_raw_buffer_store_b32(lv ? L[r] : R[r], rd, vhalf, (y0s + r) * (int)a.pitch, aux_st<NT>());
        }
    } else {
        /* the 16 per-lane offsets are built AGAIN here (the empty asm keeps the compiler from carrying the load offsets across
         * the whole filter): this path runs in the first and last block row only, but its 16 live address registers set the
         * VGPR count -- and with it the waves per SIMD -- of the entire kernel */
        uint32_t base = (uint32_t)(y0 * (int)a.pitch) + xoff;
        asm volatile("" : "+v"(base));
        const uint32_t lbits = lv ? 0u : kOob, rbits = rv ? 0u : kOob;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t ob = (unsigned)(y0 + r) < (unsigned)a.plane_h ? 0u : kOob;
            const uint32_t off = base + (uint32_t)r * (uint32_t)a.pitch;
            __builtin_amdgcn_raw_buffer_store_b32(L[r], rd, off | lbits | ob, 0, aux_st<NT>());
            __builtin_amdgcn_raw_buffer_store_b32(R[r], rd, (off + 4u) | rbits | ob, 0, aux_st<NT>());
        }
    }
}


/*
 * 16-bit containers (bit depth 8..16), luma, scalar QP: same mapping, one lane = one offset block =
 * 8 rows x 16 bytes.  Interior waves issue 8 back-to-back buffer_load_dwordx4 (a wave's row span is
 * 1 KiB contiguous, starting 8 bytes before a 16-byte boundary); edge waves use two dwordx2 halves with
 * out-of-range offsets for out-of-image halves / rows.  Arithmetic = deblock_packed.h on the samples as
 * they are (no widening needed), so twice the bytes per pixel at the same instruction count: this is
 * the variant that runs into the HBM roof (BASELINE config 5).
 */
template <int MODE, bool NT, bool EDGE, bool QPMAP, bool CHROMA = false, bool WIDE = false>
__device__ __forceinline__ void packed16_body(const DbkArgs &a, int by, int f, int bx, bool active,
                                              const DbkH265Args *hx = nullptr /* MODE 2 (spec-exact) only */,
                                              uint32_t *ktab_lds = nullptr /* QPMAP luma: LDS for the workgroup's operand table */)
{
    constexpr bool KT = QPMAP && !CHROMA && (MODE == 0 || MODE == 2); /* packed_body: built after the loads have been issued */
    const bool lv = active && bx > 0;
    const bool rv = active && bx < a.nbx - 1;
    const int y0 = by * 8 - 4;
#ifdef HEVCDBK_DIAG
    const uint32_t xoff = (uint32_t)(bx * 16 - 8) + (MODE == 1 ? (uint32_t)a.diag_xshift : 0u);
#else
    const uint32_t xoff = (uint32_t)(bx * 16 - 8);
#endif
    const uint32_t plane_bytes = (uint32_t)a.pitch * (uint32_t)a.plane_h;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(a.src) + (long long)f * a.frame_stride, 0, plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        a.dst + (long long)f * a.frame_stride, 0, plane_bytes, 0x00020000);

    uint32_t W[8][4];
    if constexpr (!EDGE) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, xoff, (y0 + r) * (int)a.pitch, aux_ld<NT>());
            W[r][0] = w.x; W[r][1] = w.y; W[r][2] = w.z; W[r][3] = w.w;
        }
    } else {
        /* out-of-image halves / rows: kOob OR-ed into the offset (packed_body, PATH 2) */
        const uint32_t base = (uint32_t)(y0 * (int)a.pitch) + xoff;
        const uint32_t lbits = lv ? 0u : kOob, rbits = rv ? 0u : kOob;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t ob = (unsigned)(y0 + r) < (unsigned)a.plane_h ? 0u : kOob;
            const uint32_t off = base + (uint32_t)r * (uint32_t)a.pitch;
            const u32x2 l = __builtin_amdgcn_raw_buffer_load_b64(rs, off | lbits | ob, 0, aux_ld<NT>());
            const u32x2 rr = __builtin_amdgcn_raw_buffer_load_b64(rs, (off + 8u) | rbits | ob, 0, aux_ld<NT>());
            W[r][0] = l.x; W[r][1] = l.y; W[r][2] = rr.x; W[r][3] = rr.y;
        }
    }

    if constexpr (MODE == 0) {
        const dbk::BlockBs bs = load_bs_buffer<EDGE>(a, f, by, bx, active);
        if constexpr (KT) {
            dbk::QsTable qs = block_qp_tab<false>(a, f, by, active ? bx : 0, nullptr);
            qs.tab = ktab_setup(ktab_lds, a);
            dbk::packed_filter_luma_block16_src<WIDE>(W, bs, qs, a.max_v);
        } else {
            const dbk::BlockQp q = block_qp<QPMAP, CHROMA>(a, f, by, active ? bx : 0);
            if constexpr (CHROMA) dbk::packed_filter_chroma_block16(W, bs, q, a.max_v);
            else dbk::packed_filter_luma_block16<WIDE, !QPMAP>(W, bs, q, a.max_v);
        }
    } else if constexpr (MODE == 2) { /* spec-exact mode, H.265 8.7.2 */
        int entry[4];
        load_bs_buffer_h265<EDGE ? 2 : 0>(a, f, by, bx, active, entry);
        dbk::H265Seg sg;
        if constexpr (QPMAP) {
            int qpl[4];
            block_qpl4_dev<CHROMA>(a, f, by, active ? bx : 0, qpl);
            const dbk::H265Prm prm = {hx->tc_off, hx->beta_off, hx->c_qp_offset, a.shift, a.max_v};
            if constexpr (KT) dbk::h265_seg_rows(entry, qpl, prm, ktab_setup_h265(ktab_lds, a.shift), sg);
            else dbk::h265_seg_params<CHROMA>(entry, qpl, prm, sg);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                sg.entry[i] = entry[i];
                sg.beta[i] = hx->beta_s;
                sg.tc[i] = (entry[i] & dbk::kH265BsMask) == 2 ? hx->tc_bs2 : hx->tc_bs1;
            }
        }
        if constexpr (!QPMAP && !CHROMA) {
            const dbk::H265Uni u = dbk::h265_uni(hx->beta_s, hx->tc_bs1, hx->tc_bs2);
            dbk::packed_filter_block16_h265<CHROMA, WIDE>(W, sg, a.max_v, &u);
        } else {
            dbk::packed_filter_block16_h265<CHROMA, WIDE, KT>(W, sg, a.max_v);
        }
    }

    if constexpr (!EDGE) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            u32x4 w;
            w.x = W[r][0]; w.y = W[r][1]; w.z = W[r][2]; w.w = W[r][3];
            __builtin_amdgcn_raw_buffer_store_b128(w, rd, xoff, (y0 + r) * (int)a.pitch, aux_st<NT>());
        }
    } else {
        uint32_t base = (uint32_t)(y0 * (int)a.pitch) + xoff;
        asm volatile("" : "+v"(base)); /* rebuild the 16 offsets here instead of carrying them across the filter (packed_body, PATH 2) */
        const uint32_t lbits = lv ? 0u : kOob, rbits = rv ? 0u : kOob;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t ob = (unsigned)(y0 + r) < (unsigned)a.plane_h ? 0u : kOob;
            const uint32_t off = base + (uint32_t)r * (uint32_t)a.pitch;
            u32x2 l, rr;
            l.x = W[r][0]; l.y = W[r][1]; rr.x = W[r][2]; rr.y = W[r][3];
            __builtin_amdgcn_raw_buffer_store_b64(l, rd, off | lbits | ob, 0, aux_st<NT>());
            __builtin_amdgcn_raw_buffer_store_b64(rr, rd, (off + 8u) | rbits | ob, 0, aux_st<NT>());
        }
    }
}

/*
 * Work mapping of the packed kernels (see WaveCoords below).
 */
struct WaveCoords {
    int f;          /* frame (scalar) */
    int by, bx;     /* offset block of this lane */
    bool active;    /* lane owns a block */
    bool interior;  /* wave-uniform: all 64 lanes in one block row, none touches the frame border */
    int by0;        /* wave-uniform: block row of the wave's first lane */
    bool rows_in;   /* wave-uniform: every lane's block row is in 1..nby-2 (all 8 pixel rows inside the image) */
};

/*
 * LINEAR == false: one workgroup per block row (blockIdx.x = by, .y = frame, .z = 512-lane chunk).
 * LINEAR == true : the offset blocks of a frame are numbered row-major, t = by*nbx + bx, and dealt to
 *   lanes in that order, so no lane idles at a row end (481 blocks per row at 4K leave 31 of 512 lanes
 *   idle otherwise) and only the ~1-in-7 waves that contain a row end / frame border take the per-lane
 *   path.  Workgroups are renumbered so that the 8 XCDs (round-robin over blockIdx, an observation used
 *   for speed only) each get a CONTIGUOUS range of workgroups: neighbouring workgroups share the cache
 *   lines at their common boundary, and their partial writes then merge in one XCD's L2.
 *   Divisions are exact multiply-high by host-computed reciprocals (dividend < 2^32 / divisor).
 */
template <bool LINEAR>
__device__ __forceinline__ bool wave_coords(const DbkArgs &a, WaveCoords &c)
{
    const int lane = (int)(threadIdx.x & 63u);
    if constexpr (!LINEAR) {
        c.by = (int)blockIdx.x + a.by_begin;
        c.f = blockIdx.y;
        c.bx = blockIdx.z * (int)blockDim.x + (int)threadIdx.x;
        const int wave_bx0 = __builtin_amdgcn_readfirstlane(c.bx) & ~63;
        c.active = c.bx < a.nbx;
        c.by0 = c.by;
        c.rows_in = c.by > 0 && c.by < a.nby - 1;
        c.interior = wave_bx0 > 0 && wave_bx0 + 64 <= a.nbx - 1 && c.rows_in;
        return true;
    } else {
        const uint32_t id = blockIdx.x, per_xcd = gridDim.x >> 3;
        const uint32_t logical = a.xcd_swizzle ? (id & 7u) * per_xcd + (id >> 3) : id;
        const uint32_t f = __umulhi(logical, a.magic_wpf);
        if (f >= (uint32_t)a.n_frames) return false; /* padding workgroup */
        c.f = (int)f;
        const uint32_t wg = logical - f * (uint32_t)a.wpf;
        const uint32_t t0 = wg * blockDim.x + (__builtin_amdgcn_readfirstlane(threadIdx.x) & ~63u); /* scalar */
        const uint32_t by0 = __umulhi(t0, a.magic_nbx), bx0 = t0 - by0 * (uint32_t)a.nbx;
        c.interior = bx0 >= 1u && bx0 + 63u <= (uint32_t)a.nbx - 2u && by0 >= 1u && by0 + 2u <= (uint32_t)a.nby;
        c.by0 = (int)by0;
        c.rows_in = by0 >= 1u && __umulhi(t0 + 63u, a.magic_nbx) + 2u <= (uint32_t)a.nby; /* last lane's row <= nby-2 */
        if (c.interior) {
            c.by = (int)by0;
            c.bx = (int)bx0 + lane;
            c.active = true;
        } else {
            const uint32_t t = t0 + (uint32_t)lane;
            const uint32_t by = __umulhi(t, a.magic_nbx);
            c.by = (int)by;
            c.bx = (int)(t - by * (uint32_t)a.nbx);
            c.active = t < (uint32_t)a.nb_total;
        }
        return true;
    }
}

template <int MODE, bool NT, bool LINEAR, bool QPMAP, bool WIDE = false>
__global__ __launch_bounds__(1024) void dbk_packed16_kernel(const DbkArgs a)
{
    uint32_t *kt = nullptr;
    if constexpr (QPMAP && MODE == 0) {
        __shared__ uint32_t ktab[dbk::kKTabDwords];
        kt = ktab;
    }
    WaveCoords c;
    if (!wave_coords<LINEAR>(a, c)) return;
    if (c.interior) packed16_body<MODE, NT, false, QPMAP, false, WIDE>(a, c.by, c.f, c.bx, true, nullptr, kt);
    else packed16_body<MODE, NT, true, QPMAP, false, WIDE>(a, c.by, c.f, c.bx, c.active, nullptr, kt);
}

/* 16-bit containers: reference-exact chroma, and the spec-exact mode's luma / chroma (deblock_packed16.h) */
template <bool LINEAR, bool QPMAP>
__global__ __launch_bounds__(1024) void dbk_packed16c_kernel(const DbkArgs a)
{
    WaveCoords c;
    if (!wave_coords<LINEAR>(a, c)) return;
    if (c.interior) packed16_body<0, false, false, QPMAP, true>(a, c.by, c.f, c.bx, true);
    else packed16_body<0, false, true, QPMAP, true>(a, c.by, c.f, c.bx, c.active);
}
template <bool CHROMA, bool LINEAR, bool QPMAP, bool WIDE = false>
__global__ __launch_bounds__(1024) void dbk_packed16_h265_kernel(const DbkH265Args h)
{
    const DbkArgs &a = h.base;
    uint32_t *kt = nullptr;
    if constexpr (QPMAP && !CHROMA) {
        __shared__ uint32_t ktab[dbk::kKTabDwords];
        kt = ktab;
    }
    WaveCoords c;
    if (!wave_coords<LINEAR>(a, c)) return;
    if (c.interior) packed16_body<2, false, false, QPMAP, CHROMA, WIDE>(a, c.by, c.f, c.bx, true, &h, kt);
    else packed16_body<2, false, true, QPMAP, CHROMA, WIDE>(a, c.by, c.f, c.bx, c.active, &h, kt);
}

/*
 * Work mapping: ONE WORKGROUP OWNS ONE BLOCK ROW of one frame (up to 1024 offset blocks per
 * workgroup; wider planes take blockIdx.z chunks).  A wave is 64 consecutive bx, so each of its
 * row accesses is one contiguous 512-byte span that starts 4 bytes before an 8-byte boundary
 * (offset blocks start at image x = 8*bx - 4).  The 128-byte lines at both ends of a wave's span
 * are shared with the neighbouring wave; keeping all waves of a block row in one workgroup keeps
 * both halves of every such line on one CU / one XCD L2, where the partial writes merge before
 * write-back.
 */
/* (round 4: forcing the QP-map instantiations to 8 waves per SIMD -- amdgpu_waves_per_eu(8, 8): 64 VGPRs and 32-64 bytes of
 * scratch in a rarely taken path -- measured the same as the compiler's own 69-72 VGPRs / 7 waves / no scratch in three
 * alternating runs each, 0.658-0.662 against 0.661-0.664 and 0.639-0.642 against 0.639-0.646: not kept) */
template <bool CHROMA, int MODE, bool NT, bool LINEAR, bool QPMAP>
__global__ __launch_bounds__(1024) void dbk_packed_kernel(const DbkArgs a)
{
    uint32_t *kt = nullptr;
    if constexpr (QPMAP && !CHROMA && (MODE == 0 || MODE == 3)) {
        __shared__ uint32_t ktab[dbk::kKTabDwords];
        kt = ktab;
    }
    WaveCoords c;
    if (!wave_coords<LINEAR>(a, c)) return; /* a padding workgroup: all of its waves leave here */
    if (c.interior) packed_body<CHROMA, MODE, NT, 0, QPMAP>(a, c.by, c.f, c.bx, true, c.by0, nullptr, kt);
    else if (c.rows_in) packed_body<CHROMA, MODE, NT, LINEAR ? 1 : 3, QPMAP>(a, c.by, c.f, c.bx, c.active, c.by0, nullptr, kt);
    else packed_body<CHROMA, MODE, NT, 2, QPMAP>(a, c.by, c.f, c.bx, c.active, c.by0, nullptr, kt);
}

/* spec-exact mode (H.265 8.7.2), 8-bit samples, packed-int16 arithmetic: same mapping and memory path */
template <bool CHROMA, bool LINEAR, bool QPMAP>
__device__ __forceinline__ void packed_h265_dispatch(const DbkH265Args &h)
{
    const DbkArgs &a = h.base;
    uint32_t *kt = nullptr;
    if constexpr (QPMAP && !CHROMA) {
        __shared__ uint32_t ktab[dbk::kKTabDwords];
        kt = ktab;
    }
    WaveCoords c;
    if (!wave_coords<LINEAR>(a, c)) return;
    if (c.interior) packed_body<CHROMA, 2, false, 0, QPMAP>(a, c.by, c.f, c.bx, true, c.by0, &h, kt);
    else if (c.rows_in) packed_body<CHROMA, 2, false, LINEAR ? 1 : 3, QPMAP>(a, c.by, c.f, c.bx, c.active, c.by0, &h, kt);
    else packed_body<CHROMA, 2, false, 2, QPMAP>(a, c.by, c.f, c.bx, c.active, c.by0, &h, kt);
}
template <bool CHROMA, bool LINEAR, bool QPMAP>
__global__ __launch_bounds__(1024) void dbk_packed_h265_kernel(const DbkH265Args h)
{
    packed_h265_dispatch<CHROMA, LINEAR, QPMAP>(h);
}
/* ------------------------------------------------------------------------------------------ */
/* one launch for the planes of a 4:2:0 frame                                */
/*
 * The reference launches three kernels per frame (gpu.cu:1269-1285).  For small frames the launches cost more
 * than the filtering, so the 8-bit scalar-QP planes of a frame go into ONE grid: blockIdx.x runs over the block
 * rows of Y, then U, then V (row map), a workgroup is as wide as the luma row, and the waves of a chroma row that
 * lie beyond its width exit at once.
 */
template <bool NT>
__global__ __launch_bounds__(1024) void dbk_packed_multi_kernel(const DbkMultiArgs m)
{
    const int row = blockIdx.x;
    const int pl = row >= m.row_end[0] ? (row >= m.row_end[1] ? 2 : 1) : 0; /* scalar */
    const DbkArgs &a = m.p[pl];
    WaveCoords c;
    c.by = row - (pl ? m.row_end[pl - 1] : 0);
    c.f = blockIdx.y;
    c.bx = blockIdx.z * (int)blockDim.x + (int)threadIdx.x;
    const int wave_bx0 = __builtin_amdgcn_readfirstlane(c.bx) & ~63;
    if (wave_bx0 >= a.nbx) return; /* chroma rows are narrower than the workgroup */
    c.active = c.bx < a.nbx;
    c.by0 = c.by;
    c.rows_in = c.by > 0 && c.by < a.nby - 1;
    c.interior = wave_bx0 > 0 && wave_bx0 + 64 <= a.nbx - 1 && c.rows_in;
    if (pl == 0) {
        if (c.interior) packed_body<false, 0, NT, 0, false>(a, c.by, c.f, c.bx, true, c.by0);
        else if (c.rows_in) packed_body<false, 0, NT, 3, false>(a, c.by, c.f, c.bx, c.active, c.by0);
        else packed_body<false, 0, NT, 2, false>(a, c.by, c.f, c.bx, c.active, c.by0);
    } else {
        if (c.interior) packed_body<true, 0, NT, 0, false>(a, c.by, c.f, c.bx, true, c.by0);
        else if (c.rows_in) packed_body<true, 0, NT, 3, false>(a, c.by, c.f, c.bx, c.active, c.by0);
        else packed_body<true, 0, NT, 2, false>(a, c.by, c.f, c.bx, c.active, c.by0);
    }
}


/* the same for 16-bit containers (bit depth 8..12; WIDE = 12-bit luma, deblock_packed.h): Y, U, V of a 10-bit 4:2:0 batch in
 * one launch */
template <bool WIDE>
__global__ __launch_bounds__(1024) void dbk_packed16_multi_kernel(const DbkMultiArgs m)
{
    const int row = blockIdx.x;
    const int pl = row >= m.row_end[0] ? (row >= m.row_end[1] ? 2 : 1) : 0; /* scalar */
    const DbkArgs &a = m.p[pl];
    const int by = row - (pl ? m.row_end[pl - 1] : 0), f = blockIdx.y;
    const int bx = blockIdx.z * (int)blockDim.x + (int)threadIdx.x;
    const int wave_bx0 = __builtin_amdgcn_readfirstlane(bx) & ~63;
    if (wave_bx0 >= a.nbx) return; /* chroma rows are narrower than the workgroup */
    const bool active = bx < a.nbx;
    const bool interior = wave_bx0 > 0 && wave_bx0 + 64 <= a.nbx - 1 && by > 0 && by < a.nby - 1;
    if (pl == 0) {
        if (interior) packed16_body<0, false, false, false, false, WIDE>(a, by, f, bx, true);
        else packed16_body<0, false, true, false, false, WIDE>(a, by, f, bx, active);
    } else {
        if (interior) packed16_body<0, false, false, false, true>(a, by, f, bx, true);
        else packed16_body<0, false, true, false, true>(a, by, f, bx, active);
    }
}

#include "deblock_sao_fused.inc"

#ifdef HEVCDBK_DIAG
/* the measured-and-rejected kernels (LDS queue, stripe map, tile map): diagnostic build only, DESIGN.md 4.1 */
#include "deblock_diag_kernels.inc"
#endif

} /* namespace */

/* largest tc (scaled to the bit depth) a lane of this launch can be handed: the scalar-QP value, or with a QP map any
 * entry of the table */
static int max_scaled_tc(const DbkArgs &a)
{
    if (!a.qp_map) return a.tc;
    int m = 0;
    for (int i = 0; i < 52; i++) m = a.tc_tab[i] > m ? a.tc_tab[i] : m;
    return m << a.shift;
}

bool dbk_packed_supports(const DbkArgs &a, int sample_bytes, bool chroma)
{
    /* the packed kernels address a plane through a buffer resource with 32-bit offsets */
    if ((unsigned long long)a.pitch * (unsigned long long)a.plane_h >= (1ull << 31)) return false;
    /* caller-supplied tc tables may exceed what the 16-bit fields of the luma core hold (deblock_packed.h) */
    if (!chroma && !dbk::packed_luma_tc_fits(a.max_v, max_scaled_tc(a))) return false;
    if (sample_bytes == 1) return a.max_v == 255;                /* 8-bit: luma and chroma */
    /* 16-bit containers up to 12 bit.  Luma: up to 11 bit every intermediate fits int16 (the normal filter's
     * 9*(q0-p0) - 3*(q1-p1) + 8 needs 12*max_v + 8 <= 32767); 12 bit runs the WIDE variant of the core.
     * Chroma: 4*(p0-q0) + p1 - q1 + 4 needs 5*max_v + 4 <= 32767 */
    return a.max_v <= 4095 && a.pitch % 8 == 0 && a.frame_stride % 8 == 0 &&
           ((uintptr_t)a.src % 8) == 0 && ((uintptr_t)a.dst % 8) == 0;
}

#ifdef HEVCDBK_DIAG
DbkDiag g_dbk_diag; /* the defaults of the struct */
static int wg_cap() { return g_dbk_diag.wg_cap; }
#else
/* workgroup width cap of the packed kernels (measured best on MI355X; the diagnostic build can vary it) */
static constexpr int wg_cap() { return 512; }
#endif

template <bool NT, bool LINEAR>
static void launch_packed_t(const DbkArgs &a, int sample_bytes, bool chroma, int mode, dim3 grid, dim3 block, hipStream_t stream)
{
    const bool qm = a.qp_map != nullptr;
#ifdef HEVCDBK_DIAG
    if (mode == 1) { /* copy: the kernel's loads and stores, no arithmetic */
        if (sample_bytes == 2) DBK_LAUNCH((dbk_packed16_kernel<1, NT, LINEAR, false>), grid, block, stream, a);
        else DBK_LAUNCH_LDS((dbk_packed_kernel<false, 1, NT, LINEAR, false>), grid, block, g_dbk_diag.lds, stream, a);
        return;
    }
    if (g_dbk_diag.lds && sample_bytes == 1 && !chroma && !qm) { /* occupancy experiment: unused dynamic LDS per workgroup */
        DBK_LAUNCH_LDS((dbk_packed_kernel<false, 0, NT, LINEAR, false>), grid, block, g_dbk_diag.lds, stream, a);
        return;
    }
    if (sample_bytes == 1 && !chroma && !qm && a.use_queue && block.x <= 512) {
        if (block.x <= 128) DBK_LAUNCH((dbk_packed_q_kernel<NT, LINEAR, 128>), grid, block, stream, a);
        else if (block.x <= 256) DBK_LAUNCH((dbk_packed_q_kernel<NT, LINEAR, 256>), grid, block, stream, a);
        else DBK_LAUNCH((dbk_packed_q_kernel<NT, LINEAR, 512>), grid, block, stream, a);
        return;
    }
    if (sample_bytes == 1 && !chroma && !qm && (a.diag_ablate || a.diag_prio || a.diag_dummy || g_dbk_diag.mode3)) { /* timing only */
        DBK_LAUNCH((dbk_packed_kernel<false, 3, NT, LINEAR, false>), grid, block, stream, a);
        return;
    }
#endif
    (void)mode;
    if (sample_bytes == 2 && chroma) {
        if (qm) DBK_LAUNCH((dbk_packed16c_kernel<LINEAR, true>), grid, block, stream, a);
        else DBK_LAUNCH((dbk_packed16c_kernel<LINEAR, false>), grid, block, stream, a);
    } else if (sample_bytes == 2) {
        if (a.max_v > 2047) { /* 12 bit: the WIDE variant of the core (deblock_packed.h) */
            if (qm) DBK_LAUNCH((dbk_packed16_kernel<0, NT, LINEAR, true, true>), grid, block, stream, a);
            else DBK_LAUNCH((dbk_packed16_kernel<0, NT, LINEAR, false, true>), grid, block, stream, a);
        } else if (qm) DBK_LAUNCH((dbk_packed16_kernel<0, NT, LINEAR, true>), grid, block, stream, a);
        else DBK_LAUNCH((dbk_packed16_kernel<0, NT, LINEAR, false>), grid, block, stream, a);
    } else if (chroma) {
        if (qm) DBK_LAUNCH((dbk_packed_kernel<true, 0, NT, LINEAR, true>), grid, block, stream, a);
        else DBK_LAUNCH((dbk_packed_kernel<true, 0, NT, LINEAR, false>), grid, block, stream, a);
    } else {
        if (qm) DBK_LAUNCH((dbk_packed_kernel<false, 0, NT, LINEAR, true>), grid, block, stream, a);
        else DBK_LAUNCH((dbk_packed_kernel<false, 0, NT, LINEAR, false>), grid, block, stream, a);
    }
}

/* work mapping of the packed kernels for plane geometry `a`: fills the row-major numbering fields of `b` and the launch
 * dimensions; returns true for the row-major (LINEAR) map, false for one workgroup per block row */
static bool plan_packed(const DbkArgs &a, DbkArgs &b, dim3 &grid, dim3 &block)
{
    const int cap = wg_cap();
    const long long nb = (long long)a.nbx * a.nby;
    /* row-major block numbering needs exact 32-bit reciprocal division: dividends < 2^32 / divisor */
    const int wg = (int)(nb < cap ? (nb + 63) / 64 * 64 : cap);
    const long long wpf = (nb + wg - 1) / wg;
    const long long total = (wpf * a.n_frames + 7) / 8 * 8; /* multiple of 8: one contiguous range per XCD */
    /* reciprocal division floor(2^32/d)+1 is exact for dividends < 2^32/d and needs d >= 2 */
    /* measured on MI355X: when a whole block row fits one workgroup (4K: 481 blocks) the row mapping wins
     * (3.96 vs 3.69 TB/s at 4K 8-bit, idle lanes included); wider rows (8K: 961 blocks) do better row-major
     * (5.06 vs 4.89 TB/s at 8K 10-bit).  a.map_override (HEVCDBK_MAP_ROWS / HEVCDBK_MAP_LINEAR of the C ABI) forces
     * either; both give the same bytes. */
    /* a block-row range (strip launches of the host pipeline) always takes the row map */
    const bool want_linear = a.by_count == 0 && (a.map_override == 2 || (a.map_override != 1 && a.nbx > cap));
    const bool linear = want_linear && wpf >= 2 && a.nbx >= 2 && (nb + 1024) * a.nbx < (1ll << 32) &&
                        total * wpf < (1ll << 32) && total < (1ll << 31);
    if (linear) {
        b.nb_total = (int)nb;
        b.wpf = (int)wpf;
#ifdef HEVCDBK_DIAG
        b.xcd_swizzle = !g_dbk_diag.noswz;
#else
        b.xcd_swizzle = 1;
#endif
        b.magic_wpf = (uint32_t)((1ull << 32) / (unsigned long long)wpf + 1ull);
        b.magic_nbx = (uint32_t)((1ull << 32) / (unsigned long long)a.nbx + 1ull);
        block = dim3(wg, 1, 1);
        grid = dim3((unsigned)total, 1, 1);
#ifdef HEVCDBK_DIAG
        if (g_dbk_diag.xpad > 0 && b.xcd_swizzle) grid = dim3((unsigned)(total + 8ll * g_dbk_diag.xpad), 1, 1); /* experiment: see DbkDiag::xpad */
#endif
        return true;
    }
    /* small planes / degenerate divisors: one workgroup per block row, wider rows split into cap-lane chunks */
    const int per_wg = a.nbx < cap ? a.nbx : cap;
    block = dim3((per_wg + 63) / 64 * 64, 1, 1);
    grid = dim3(a.by_count ? a.by_count : a.nby, a.n_frames, (a.nbx + (int)block.x - 1) / (int)block.x);
    return false;
}

#ifdef HEVCDBK_DIAG
/* the stripe map (dbk_stripe_kernel): 8-bit luma, scalar QP, whole planes, an even number of blocks per row beside the
 * left border column; fills the st_* fields of `b` and the launch size */
static bool plan_stripe(const DbkArgs &a, DbkArgs &b, int sample_bytes, bool chroma, unsigned &grid)
{
    if (sample_bytes != 1 || chroma || a.qp_map || a.by_count != 0 || a.max_v != 255) return false;
    const long long M = a.nbx - 1, rows = a.nby - 2;
    if (M < 16 || (M % 16) || rows < 1) return false; /* pairs / 16-byte bS pieces never straddle a row end */
    /* row group: k rows whose k*M blocks fill whole waves as nearly as possible (4K: M = 480, k = 2 -> 15 waves, none idle) */
    int k = 1;
    double best = 2.0;
    for (int c = 1; c <= 8 && c <= rows; c++) {
        const long long waves = (c * M + 63) / 64;
        const double waste = (double)(waves * 64 - c * M) / (double)(waves * 64);
        if (waste + 1e-9 < best) { best = waste; k = c; }
    }
    const long long W = (k * M + 63) / 64, groups = (rows + k - 1) / k, items = groups * a.n_frames;
    /* exact reciprocal division: dividend < 2^32 / divisor */
    if (items * groups >= (1ll << 32) || (W * 64) * M >= (1ll << 32)) return false;
    if (W > 16) return false;
    /* one buffer resource per array for the whole batch, frames addressed through the 32-bit scalar offset */
    if ((unsigned long long)a.frame_stride * (unsigned long long)a.n_frames >= (1ull << 32) ||
        (unsigned long long)a.vert_bs_stride * (unsigned long long)a.n_frames + (unsigned long long)a.n_vert >= (1ull << 32) ||
        (unsigned long long)a.hor_bs_stride * (unsigned long long)a.n_frames + (unsigned long long)a.n_hor >= (1ull << 32))
        return false;
    /* persistent workgroups: as many as are resident at once (W waves, W * kStripeLdsPerWave bytes of LDS each) */
    const size_t lds = (size_t)W * kStripeLdsPerWave;
    static int wgs_per_cu_cache[17] = {0};
    if (wgs_per_cu_cache[W] == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dbk_stripe_kernel<0>, (int)(64 * W), lds) != hipSuccess || n <= 0)
            n = (int)(28 / W) > 0 ? (int)(28 / W) : 1;
        wgs_per_cu_cache[W] = n;
    }
    const int wgs_per_cu = wgs_per_cu_cache[W];
    long long Q = (long long)(a.n_cus > 0 ? a.n_cus : 256) * wgs_per_cu;
    if (Q > items) Q = items;
    const long long bwpf = (2ll * a.nbx + a.nby - 2 + 63) / 64;
    const long long border_wgs = (bwpf * a.n_frames + W - 1) / W, total = border_wgs + Q;
    if (total >= (1ll << 31) || (border_wgs * W + 16) * bwpf >= (1ll << 32)) return false;
    auto magic = [](long long d) { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / (unsigned long long)d + 1ull); };
    b.st_M = (int)M;
    b.st_k = k;
    b.st_W = (int)W;
    b.st_Q = (int)Q;
    b.st_groups = (int)groups;
    b.st_items = (int)items;
    b.st_border_wgs = (int)border_wgs;
    b.st_lds_bytes = (int)lds;
    b.st_border_wpf = (int)bwpf;
    b.st_magic_M = magic(M);
    b.st_magic_groups = magic(groups);
    b.st_magic_bwpf = magic(bwpf);
    grid = (unsigned)total;
#ifdef HEVCDBK_DIAG
    {
        static bool said = false;
        if (!said) {
            said = true;
            fprintf(stderr, "[diag] stripe plan: M %lld k %d W %lld Q %lld groups %lld items %lld WGs/CU %d CUs %d border waves/frame %lld border WGs %lld grid %u\n",
                    M, k, W, Q, groups, items, wgs_per_cu, a.n_cus, bwpf, border_wgs, grid);
        }
    }
#endif
    return true;
}

#endif /* HEVCDBK_DIAG */

#ifdef HEVCDBK_DIAG
/* the tile map (dbk_tile_kernel + dbk_col0_kernel): 8-bit luma, scalar QP, whole planes whose rows are whole 128-byte
 * lines; fills the tl_* fields of `b`, the grid's x size and the dynamic LDS size */
static bool plan_tile(const DbkArgs &a, DbkArgs &b, int sample_bytes, bool chroma, unsigned &grid_x, size_t &lds)
{
    if (sample_bytes != 1 || chroma || a.qp_map || a.by_count != 0 || a.max_v != 255) return false;
    if (a.plane_w % 128 != 0 || a.pitch % 16 != 0 || a.frame_stride % 16 != 0 || ((uintptr_t)a.src % 16) != 0 ||
        ((uintptr_t)a.dst % 16) != 0 || a.nby < 2)
        return false;
    const long long M = a.nbx - 1;
    /* k block rows per workgroup: at most 16 waves and 64 KiB of LDS (two workgroups per CU), fewest idle lanes */
    int k = 0;
    double best = 2.0;
    for (int c = 1; c <= 8; c++) {
        const long long waves = (c * M + 63) / 64;
        if (waves > 16 || 8ll * c * a.plane_w + 16 > 65536) break;
        const double waste = (double)(waves * 64 - c * M) / (double)(waves * 64);
        if (waste + 1e-9 < best) { best = waste; k = c; }
    }
    if (k == 0) return false;
    const long long nw = (k * M + 63) / 64, cw = a.plane_w / 16, ndma = 8ll * k * a.plane_w / 1024;
    if ((nw * 64) * M >= (1ll << 32) || (ndma * 64 + 1024) * cw >= (1ll << 32)) return false;
    auto magic = [](long long d) { return (uint32_t)((1ull << 32) / (unsigned long long)d + 1ull); }; /* d >= 2 */
    b.tl_k = k;
    b.tl_M = (int)M;
    b.tl_nw = (int)nw;
    b.tl_cw = (int)cw;
    b.tl_ndma = (int)ndma;
    b.tl_magic_M = magic(M);
    b.tl_magic_cw = magic(cw);
    grid_x = (unsigned)((a.nby + k - 1) / k);
    lds = (size_t)(8ll * k * a.plane_w + 16);
    return true;
}

#endif /* HEVCDBK_DIAG */

hipError_t dbk_launch_packed(const DbkArgs &a, int sample_bytes, bool chroma, int mode, hipStream_t stream)
{
    if (a.n_frames <= 0 || a.nbx <= 0 || a.nby <= 0) return hipSuccess;
    DbkArgs b = a;
#ifdef HEVCDBK_DIAG
    b.diag_ablate = g_dbk_diag.ablate;
    b.diag_prio = g_dbk_diag.prio;
    b.diag_dummy = g_dbk_diag.dummy;
    b.use_queue = g_dbk_diag.queue;
    b.diag_xshift = (mode == 1 && g_dbk_diag.align) ? 4 * sample_bytes : 0;
#endif
#ifdef HEVCDBK_DIAG
    if (a.map_override == 4) { /* HEVCDBK_DIAG_MAP_TILES */
        unsigned gx = 0;
        size_t lds = 0;
        const DbkArgs b0 = b;
        if (plan_tile(b0, b, sample_bytes, chroma, gx, lds)) {
            const dim3 grid(gx, (unsigned)b.n_frames, 1), block(64u * (unsigned)b.tl_nw, 1, 1), cgrid(((unsigned)b.nby + 63u) / 64u, (unsigned)b.n_frames, 1);
            hipEvent_t s_ = t_next_start, e_ = t_next_stop; /* two launches: the first stamps the start, the second the stop */
            t_next_start = t_next_stop = nullptr;
            if (mode == 1) {
                hipExtLaunchKernelGGL((dbk_tile_kernel<1>), grid, block, lds, stream, s_, nullptr, 0, b);
                hipExtLaunchKernelGGL((dbk_col0_kernel<1>), cgrid, dim3(64), 0, stream, nullptr, e_, 0, b);
            } else {
                hipExtLaunchKernelGGL((dbk_tile_kernel<0>), grid, block, lds, stream, s_, nullptr, 0, b);
                hipExtLaunchKernelGGL((dbk_col0_kernel<0>), cgrid, dim3(64), 0, stream, nullptr, e_, 0, b);
            }
            return hipGetLastError();
        }
        b = b0;
    }
#endif
#ifdef HEVCDBK_DIAG
    if (a.map_override == 5 && sample_bytes == 1 && !chroma && !a.qp_map && a.by_count == 0 && a.max_v == 255) { /* HEVCDBK_DIAG_MAP_PIPE */
        const int cap = wg_cap();
        const int per_wg = a.nbx < cap ? a.nbx : cap;
        const int n = g_dbk_diag.rows > 0 ? g_dbk_diag.rows : 4;
        const dim3 block((per_wg + 63) / 64 * 64, 1, 1);
        const dim3 grid((a.nby + n - 1) / n, a.n_frames, (a.nbx + (int)block.x - 1) / (int)block.x);
        if (mode == 1) DBK_LAUNCH((dbk_pipe_kernel<1>), grid, block, stream, b, n);
        else DBK_LAUNCH((dbk_pipe_kernel<0>), grid, block, stream, b, n);
        return hipGetLastError();
    }
#endif
#ifdef HEVCDBK_DIAG
    if (a.map_override == 6 && sample_bytes == 1 && !chroma && !a.qp_map && a.by_count == 0 && a.max_v == 255 && a.nbx >= 3 && a.nby >= 3) { /* HEVCDBK_DIAG_MAP_GROUP */
        const long long M = a.nbx - 1, rows = a.nby - 2;
        int k = 1;
        double best = 2.0;
        for (int c = 1; c <= 8 && c <= rows; c++) {
            const long long waves = (c * M + 63) / 64;
            if (waves > 16) break;
            const double waste = (double)(waves * 64 - c * M) / (double)(waves * 64);
            if (waste + 1e-9 < best) { best = waste; k = c; }
        }
        const long long Wv = (k * M + 63) / 64, groups = (rows + k - 1) / k;
        if (Wv <= 16 && (Wv * 64 + 64) * M < (1ll << 32)) {
            const long long bwaves = (2ll * a.nbx + a.nby - 2 + 63) / 64, bwgs = (bwaves + Wv - 1) / Wv;
            b.st_M = (int)M; b.st_k = k; b.st_W = (int)Wv; b.st_groups = (int)groups; b.st_border_wgs = (int)bwgs;
            b.st_magic_M = M <= 1 ? 0u : (uint32_t)((1ull << 32) / (unsigned long long)M + 1ull);
            const dim3 grid((unsigned)(bwgs + groups), (unsigned)a.n_frames, 1), block((unsigned)(64 * Wv), 1, 1);
            if (mode == 1) DBK_LAUNCH((dbk_group_kernel<1>), grid, block, stream, b);
            else DBK_LAUNCH((dbk_group_kernel<0>), grid, block, stream, b);
            return hipGetLastError();
        }
    }
#endif
#ifdef HEVCDBK_DIAG
    if (a.map_override == 3) { /* HEVCDBK_DIAG_MAP_STRIPE */
        unsigned sgrid = 0;
        const DbkArgs b0 = b;
        if (plan_stripe(b0, b, sample_bytes, chroma, sgrid)) {
            if (mode == 1) DBK_LAUNCH_LDS((dbk_stripe_kernel<1>), dim3(sgrid), dim3(64 * b.st_W), b.st_lds_bytes, stream, b);
            else DBK_LAUNCH_LDS((dbk_stripe_kernel<0>), dim3(sgrid), dim3(64 * b.st_W), b.st_lds_bytes, stream, b);
            return hipGetLastError();
        }
        b = b0; /* operands the stripes do not take: the geometry's own map */
    }
#endif
#ifndef HEVCDBK_DIAG
    if (mode != 0) return hipErrorInvalidValue; /* the copy diagnostic exists in libhevcdbk_diag.so only */
#endif
    dim3 grid, block;
    const bool linear = plan_packed(a, b, grid, block);
    if (linear) {
        launch_packed_t<false, true>(b, sample_bytes, chroma, mode, grid, block, stream);
    } else {
        launch_packed_t<false, false>(b, sample_bytes, chroma, mode, grid, block, stream);
    }
    return hipGetLastError();
}

/* all planes of one sample width and bit depth that the packed kernels take (8-bit, or 16-bit containers up to 12 bit),
 * scalar QP, same frame count; plane 0 luma, the others chroma */
bool dbk_multi_supports(const DbkArgs *planes, int n, const int *sample_bytes)
{
#ifdef HEVCDBK_DIAG
    if (g_dbk_diag.nofuse) return false;
#endif
    if (n < 2 || n > 3) return false;
    for (int i = 0; i < n; i++)
        if (sample_bytes[i] != sample_bytes[0] || planes[i].qp_map || planes[i].max_v != planes[0].max_v ||
            planes[i].n_frames != planes[0].n_frames || planes[i].nbx > planes[0].nbx ||
            !dbk_packed_supports(planes[i], sample_bytes[i], i > 0))
            return false;
    return planes[0].nbx <= 1024;
}

hipError_t dbk_launch_packed_multi(const DbkArgs *planes, int n, int sample_bytes, hipStream_t stream)
{
    DbkMultiArgs m;
    std::memset(&m, 0, sizeof(m));
    int rows = 0;
    for (int i = 0; i < 3; i++) {
        if (i < n) {
            m.p[i] = planes[i];
            rows += planes[i].nby;
        }
        m.row_end[i] = rows;
    }
    if (rows <= 0 || planes[0].n_frames <= 0) return hipSuccess;
    const int cap = wg_cap();
    const int per_wg = planes[0].nbx < cap ? planes[0].nbx : cap;
    dim3 block((per_wg + 63) / 64 * 64, 1, 1);
    dim3 grid(rows, planes[0].n_frames, (planes[0].nbx + (int)block.x - 1) / (int)block.x);
    if (sample_bytes == 1) DBK_LAUNCH((dbk_packed_multi_kernel<false>), grid, block, stream, m);
    else if (planes[0].max_v > 2047) DBK_LAUNCH((dbk_packed16_multi_kernel<true>), grid, block, stream, m);
    else DBK_LAUNCH((dbk_packed16_multi_kernel<false>), grid, block, stream, m);
    return hipGetLastError();
}

/* spec-exact mode, packed kernels: 8-bit samples (luma or chroma), scalar QP or QP map */
bool dbk_packed_h265_supports(const DbkH265Args &h, int sample_bytes, bool chroma)
{
    const DbkArgs &a = h.base;
    if ((unsigned long long)a.pitch * (unsigned long long)a.plane_h >= (1ull << 31)) return false;
    /* tc comes from Table 8-12 (<= 24), scaled to the bit depth: always inside the packed core's range, checked anyway */
    if (!chroma && !dbk::packed_luma_tc_fits(a.max_v, 24 << a.shift)) return false;
    if (sample_bytes == 1) return a.max_v == 255;
    /* 16-bit containers up to 12 bit (12-bit luma: the WIDE variant), see deblock_packed16.h */
    return a.max_v <= 4095 && a.pitch % 8 == 0 && a.frame_stride % 8 == 0 &&
           ((uintptr_t)a.src % 8) == 0 && ((uintptr_t)a.dst % 8) == 0;
}

hipError_t dbk_launch_packed_h265(const DbkH265Args &h, int sample_bytes, bool chroma, hipStream_t stream)
{
    if (h.base.n_frames <= 0 || h.base.nbx <= 0 || h.base.nby <= 0) return hipSuccess;
    DbkH265Args g = h;
    dim3 grid, block;
    const bool linear = plan_packed(h.base, g.base, grid, block);
    {
        const int qp = h.qp;
        auto cl = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
        const int sh = h.base.shift; /* tc' and beta' scale with the bit depth, 8.7.2.5.3 */
        g.beta_s = dbk::h265_beta(cl(qp + h.beta_off, 0, 51)) << sh;
        if (chroma) {
            g.tc_bs1 = 0;
            g.tc_bs2 = dbk::h265_tc(cl(dbk::h265_chroma_qp(qp + h.c_qp_offset) + 2 + h.tc_off, 0, 53)) << sh;
        } else {
            g.tc_bs1 = dbk::h265_tc(cl(qp + h.tc_off, 0, 53)) << sh;
            g.tc_bs2 = dbk::h265_tc(cl(qp + 2 + h.tc_off, 0, 53)) << sh;
        }
    }
#define DBK_H265_LAUNCH(C, LIN)                                                                                       \
    do {                                                                                                              \
        if (sample_bytes == 2 && !C && g.base.max_v > 2047) { /* 12-bit luma */                                       \
            if (g.base.qp_map) DBK_LAUNCH((dbk_packed16_h265_kernel<false, LIN, true, true>), grid, block, stream, g); \
            else DBK_LAUNCH((dbk_packed16_h265_kernel<false, LIN, false, true>), grid, block, stream, g);  \
        } else if (sample_bytes == 2) {                                                                               \
            if (g.base.qp_map) DBK_LAUNCH((dbk_packed16_h265_kernel<C, LIN, true>), grid, block, stream, g); \
            else DBK_LAUNCH((dbk_packed16_h265_kernel<C, LIN, false>), grid, block, stream, g);            \
        } else if (g.base.qp_map) DBK_LAUNCH((dbk_packed_h265_kernel<C, LIN, true>), grid, block, stream, g); \
        else DBK_LAUNCH((dbk_packed_h265_kernel<C, LIN, false>), grid, block, stream, g);                  \
    } while (0)
    if (linear) {
        if (chroma) DBK_H265_LAUNCH(true, true);
        else DBK_H265_LAUNCH(false, true);
    } else {
        if (chroma) DBK_H265_LAUNCH(true, false);
        else DBK_H265_LAUNCH(false, false);
    }
#undef DBK_H265_LAUNCH
    return hipGetLastError();
}

/* ---- deblocking + SAO in one kernel ---- */
bool dbk_deblock_sao_supports(const DbkArgs &d, const DbkSaoArgs &s, int sample_bytes, bool chroma)
{
    if (d.by_count != 0 || d.max_v != s.max_v) return false;
    if (sample_bytes == 1) {
        if (d.max_v != 255 || s.band_shift != 3) return false;
    } else {
        /* 16-bit containers up to 12 bit; the SAO stage's output rows are 16-byte stores */
        if (sample_bytes != 2 || d.max_v > 4095 || s.band_shift < 3 || (1 << (s.band_shift + 5)) - 1 != s.max_v) return false;
    }
    if (!dbk_packed_supports(d, sample_bytes, chroma)) return false;
    if (d.plane_w != s.plane_w || d.plane_h != s.plane_h || d.n_frames != s.n_frames || d.n_frames > 65535) return false;
    const long long al = 4 * sample_bytes;
    if ((unsigned long long)s.pitch * (unsigned long long)s.plane_h >= (1ull << 31) || s.pitch % al != 0 || s.frame_stride % al != 0 ||
        ((uintptr_t)s.dst % al) != 0)
        return false;
    const int tw = sample_bytes == 1 ? kFusedTileW : kFused16Tile, th = sample_bytes == 1 ? kFusedTile : kFused16Tile;
    const unsigned long long tiles = (unsigned long long)((d.plane_w + tw - 1) / tw) * ((d.plane_h + th - 1) / th);
    return tiles * (unsigned long long)d.n_frames + 8 < (1ull << 31) && tiles * tiles < (1ull << 32) &&
           (tiles * d.n_frames + 8) * tiles < (1ull << 32); /* exact reciprocal divisions: dividend < 2^32 / divisor */
}

/* 1-D grid of the fused kernel, a multiple of 8 workgroups */
static unsigned fused_grid(int plane_w, int plane_h, int n_frames, int sample_bytes, DbkFusedGrid &g)
{
    const int tw = sample_bytes == 1 ? kFusedTileW : kFused16Tile, th = sample_bytes == 1 ? kFusedTile : kFused16Tile;
    const unsigned long long tx = (plane_w + tw - 1) / tw, ty = (plane_h + th - 1) / th, tpf = tx * ty;
    g.tiles_x = (uint32_t)tx;
    g.tiles_per_frame = (uint32_t)tpf;
    g.total = (uint32_t)(tpf * n_frames);
    g.magic_tpf = tpf <= 1 ? 0u : (uint32_t)((1ull << 32) / tpf + 1ull);
    g.magic_tx = tx <= 1 ? 0u : (uint32_t)((1ull << 32) / tx + 1ull);
    const unsigned grid = (g.total + 7u) / 8u * 8u;
    g.per_xcd = grid / 8u;
    return grid;
}

/* the scalar-QP operands of the packed spec-exact kernels, as dbk_launch_packed_h265 derives them */
static void fused_h265_scalars(DbkH265Args &d, bool chroma)
{
    auto cl = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    const int sh = d.base.shift, qp = d.qp;
    d.beta_s = dbk::h265_beta(cl(qp + d.beta_off, 0, 51)) << sh;
    if (chroma) {
        d.tc_bs1 = 0;
        d.tc_bs2 = dbk::h265_tc(cl(dbk::h265_chroma_qp(qp + d.c_qp_offset) + 2 + d.tc_off, 0, 53)) << sh;
    } else {
        d.tc_bs1 = dbk::h265_tc(cl(qp + d.tc_off, 0, 53)) << sh;
        d.tc_bs2 = dbk::h265_tc(cl(qp + 2 + d.tc_off, 0, 53)) << sh;
    }
}

/* launch K<..., QPMAP> with QPMAP = the plane carries a QP map */
#define DBK_FUSED_LAUNCH(qm, K, threads, lds, ...)                                                 \
    do {                                                                                           \
        if (qm) DBK_LAUNCH_LDS((K<__VA_ARGS__, true>), grid, dim3(threads), lds, stream, fa);      \
        else DBK_LAUNCH_LDS((K<__VA_ARGS__, false>), grid, dim3(threads), lds, stream, fa);        \
    } while (0)

hipError_t dbk_launch_deblock_sao(const DbkArgs &d, const DbkSaoArgs &s, int sample_bytes, bool chroma, hipStream_t stream)
{
    if (d.n_frames <= 0 || d.nbx <= 0 || d.nby <= 0) return hipSuccess;
    DbkFusedArgs fa;
    fa.d = d;
#ifdef HEVCDBK_DIAG
    fa.d.diag_prio = g_dbk_diag.prio; /* wave priority experiment of the fused kernel (deblock_sao_fused.inc) */
#endif
    fa.s = s;
    const bool qm = d.qp_map != nullptr;
    const dim3 grid(fused_grid(d.plane_w, d.plane_h, d.n_frames, sample_bytes, fa.g), 1, 1);
    if (sample_bytes == 1) {
        if (chroma) DBK_FUSED_LAUNCH(qm, dbk_sao_fused_kernel, kFusedThreads, kFusedLds, true);
        else DBK_FUSED_LAUNCH(qm, dbk_sao_fused_kernel, kFusedThreads, kFusedLds, false);
    } else {
        if (chroma) DBK_FUSED_LAUNCH(qm, dbk_sao_fused16_kernel, kFused16Threads, kFused16Lds, true, false);
        else if (d.max_v > 2047) DBK_FUSED_LAUNCH(qm, dbk_sao_fused16_kernel, kFused16Threads, kFused16Lds, false, true);
        else DBK_FUSED_LAUNCH(qm, dbk_sao_fused16_kernel, kFused16Threads, kFused16Lds, false, false);
    }
    return hipGetLastError();
}

hipError_t dbk_launch_deblock_sao_h265(const DbkH265Args &h, const DbkSaoArgs &s, int sample_bytes, bool chroma, hipStream_t stream)
{
    if (h.base.n_frames <= 0 || h.base.nbx <= 0 || h.base.nby <= 0) return hipSuccess;
    DbkFusedH265Args fa;
    fa.d = h;
    fa.s = s;
    fused_h265_scalars(fa.d, chroma);
    const bool qm = h.base.qp_map != nullptr;
    const dim3 grid(fused_grid(h.base.plane_w, h.base.plane_h, h.base.n_frames, sample_bytes, fa.g), 1, 1);
    if (sample_bytes == 1) {
        if (chroma) DBK_FUSED_LAUNCH(qm, dbk_sao_fused_h265_kernel, kFusedThreads, kFusedLds, true);
        else DBK_FUSED_LAUNCH(qm, dbk_sao_fused_h265_kernel, kFusedThreads, kFusedLds, false);
    } else {
        if (chroma) DBK_FUSED_LAUNCH(qm, dbk_sao_fused16_h265_kernel, kFused16Threads, kFused16Lds, true, false);
        else if (h.base.max_v > 2047) DBK_FUSED_LAUNCH(qm, dbk_sao_fused16_h265_kernel, kFused16Threads, kFused16Lds, false, true);
        else DBK_FUSED_LAUNCH(qm, dbk_sao_fused16_h265_kernel, kFused16Threads, kFused16Lds, false, false);
    }
    return hipGetLastError();
}

hipError_t dbk_launch_deblock_sao_multi(const DbkArgs *d, const DbkSaoArgs *s, int n, int sample_bytes, hipStream_t stream)
{
    if (n < 2 || n > 3) return hipErrorInvalidValue;
    if (d[0].n_frames <= 0) return hipSuccess;
    DbkFusedMultiArgs m;
    std::memset(&m, 0, sizeof(m));
    unsigned total = 0;
    for (int i = 0; i < 3; i++) {
        if (i < n) {
            m.pl[i].d = d[i];
            m.pl[i].s = s[i];
            total += fused_grid(d[i].plane_w, d[i].plane_h, d[i].n_frames, sample_bytes, m.pl[i].g);
        }
        m.wg_end[i] = total;
    }
    const dim3 grid(total, 1, 1);
    const DbkFusedMultiArgs &fa = m;
    const bool qm = d[0].qp_map != nullptr; /* the caller checked: all planes with a map, or none */
    if (sample_bytes == 1) DBK_FUSED_LAUNCH(qm, dbk_sao_fused_multi_kernel, kFusedThreads, kFusedLds, 1, false);
    else if (d[0].max_v > 2047) DBK_FUSED_LAUNCH(qm, dbk_sao_fused_multi_kernel, kFused16Threads, kFused16Lds, 2, true);
    else DBK_FUSED_LAUNCH(qm, dbk_sao_fused_multi_kernel, kFused16Threads, kFused16Lds, 2, false);
    return hipGetLastError();
}

hipError_t dbk_launch_deblock_sao_multi_h265(const DbkH265Args *h, const DbkSaoArgs *s, int n, int sample_bytes, hipStream_t stream)
{
    if (n < 2 || n > 3) return hipErrorInvalidValue;
    if (h[0].base.n_frames <= 0) return hipSuccess;
    DbkFusedMultiH265Args m;
    std::memset(&m, 0, sizeof(m));
    unsigned total = 0;
    for (int i = 0; i < 3; i++) {
        if (i < n) {
            m.pl[i].d = h[i];
            m.pl[i].s = s[i];
            fused_h265_scalars(m.pl[i].d, i > 0);
            total += fused_grid(h[i].base.plane_w, h[i].base.plane_h, h[i].base.n_frames, sample_bytes, m.pl[i].g);
        }
        m.wg_end[i] = total;
    }
    const dim3 grid(total, 1, 1);
    const DbkFusedMultiH265Args &fa = m;
    const bool qm = h[0].base.qp_map != nullptr; /* the caller checked: all planes with a map, or none */
    if (sample_bytes == 1) DBK_FUSED_LAUNCH(qm, dbk_sao_fused_multi_h265_kernel, kFusedThreads, kFusedLds, 1, false);
    else if (h[0].base.max_v > 2047) DBK_FUSED_LAUNCH(qm, dbk_sao_fused_multi_h265_kernel, kFused16Threads, kFused16Lds, 2, true);
    else DBK_FUSED_LAUNCH(qm, dbk_sao_fused_multi_h265_kernel, kFused16Threads, kFused16Lds, 2, false);
    return hipGetLastError();
}
