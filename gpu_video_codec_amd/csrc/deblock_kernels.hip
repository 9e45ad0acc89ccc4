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

#include <cstdlib>
#include <cstring>

#include "deblock_core.h"
#include "deblock_kernels.h"
#include "deblock_packed.h"

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
    const int by = blockIdx.y * 4 + threadIdx.y;
    const int f = blockIdx.z;
    if (bx >= a.nbx || by >= a.nby) return;

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
    dim3 grid((a.nbx + 63) / 64, (a.nby + 3) / 4, a.n_frames);
    if (a.qp_map)
        hipLaunchKernelGGL((dbk_generic_kernel<T, CHROMA, true>), grid, block, 0, stream, a);
    else
        hipLaunchKernelGGL((dbk_generic_kernel<T, CHROMA, false>), grid, block, 0, stream, a);
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

/* cache-policy bits of the raw buffer builtins on gfx950: bit 1 = nt (streamed, evict first) */
template <bool NT> constexpr int aux_bits() { return NT ? 2 : 0; }
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
template <bool CHROMA, int MODE, bool NT, bool EDGE>
__device__ __forceinline__ void packed_body(const DbkArgs &a, int by, int f, int bx, bool active)
{
    /* EDGE == false: by is wave-uniform (scalar row offsets); EDGE == true: by and bx may differ per lane */
    const bool lv = active && bx > 0;            /* cols 0..3 inside the image */
    const bool rv = active && bx < a.nbx - 1;    /* cols 4..7 inside the image */
    const int y0 = by * 8 - 4;
    const uint32_t xoff = (uint32_t)(bx * 8 - 4) + (MODE == 1 ? (uint32_t)a.diag_xshift : 0u);
    const uint32_t plane_bytes = (uint32_t)a.pitch * (uint32_t)a.plane_h;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(a.src) + (long long)f * a.frame_stride, 0, plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        a.dst + (long long)f * a.frame_stride, 0, plane_bytes, 0x00020000);

    uint32_t L[8], R[8];
    if constexpr (!EDGE) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, xoff, (y0 + r) * (int)a.pitch, aux_bits<NT>());
            L[r] = w.x;
            R[r] = w.y;
        }
    } else {
        /* straight-line, per-lane offsets: an out-of-image half or row gets an out-of-range offset (load -> 0) */
        const uint32_t base = (uint32_t)(y0 * (int)a.pitch) + xoff;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const bool yv = (unsigned)(y0 + r) < (unsigned)a.plane_h;
            const uint32_t off = base + (uint32_t)r * (uint32_t)a.pitch;
            L[r] = __builtin_amdgcn_raw_buffer_load_b32(rs, (yv && lv) ? off : kOob, 0, aux_bits<NT>());
            R[r] = __builtin_amdgcn_raw_buffer_load_b32(rs, (yv && rv) ? off + 4u : kOob, 0, aux_bits<NT>());
        }
    }

    if constexpr (MODE == 0) {
        const dbk::BlockBs bs = load_bs_buffer<EDGE>(a, f, by, bx, active);
        dbk::packed_filter_block<CHROMA>(L, R, bs, a.tc, a.beta, a.diag_ablate);
    }

    if constexpr (!EDGE) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            u32x2 w;
            w.x = L[r];
            w.y = R[r];
            __builtin_amdgcn_raw_buffer_store_b64(w, rd, xoff, (y0 + r) * (int)a.pitch, aux_bits<NT>());
        }
    } else {
        const uint32_t base = (uint32_t)(y0 * (int)a.pitch) + xoff;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const bool yv = (unsigned)(y0 + r) < (unsigned)a.plane_h;
            const uint32_t off = base + (uint32_t)r * (uint32_t)a.pitch;
            __builtin_amdgcn_raw_buffer_store_b32(L[r], rd, (yv && lv) ? off : kOob, 0, aux_bits<NT>());
            __builtin_amdgcn_raw_buffer_store_b32(R[r], rd, (yv && rv) ? off + 4u : kOob, 0, aux_bits<NT>());
        }
    }
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

/*
 * 16-bit containers (bit depth 8..16), luma, scalar QP: same mapping, one lane = one offset block =
 * 8 rows x 16 bytes.  Interior waves issue 8 back-to-back buffer_load_dwordx4 (a wave's row span is
 * 1 KiB contiguous, starting 8 bytes before a 16-byte boundary); edge waves use two dwordx2 halves with
 * out-of-range offsets for out-of-image halves / rows.  Arithmetic = deblock_packed.h on the samples as
 * they are (no widening needed), so twice the bytes per pixel at the same instruction count: this is
 * the variant that runs into the HBM roof (BASELINE config 5).
 */
template <int MODE, bool NT, bool EDGE>
__device__ __forceinline__ void packed16_body(const DbkArgs &a, int by, int f, int bx, bool active)
{
    const bool lv = active && bx > 0;
    const bool rv = active && bx < a.nbx - 1;
    const int y0 = by * 8 - 4;
    const uint32_t xoff = (uint32_t)(bx * 16 - 8) + (MODE == 1 ? (uint32_t)a.diag_xshift : 0u);
    const uint32_t plane_bytes = (uint32_t)a.pitch * (uint32_t)a.plane_h;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(a.src) + (long long)f * a.frame_stride, 0, plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        a.dst + (long long)f * a.frame_stride, 0, plane_bytes, 0x00020000);

    uint32_t W[8][4];
    if constexpr (!EDGE) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, xoff, (y0 + r) * (int)a.pitch, aux_bits<NT>());
            W[r][0] = w.x; W[r][1] = w.y; W[r][2] = w.z; W[r][3] = w.w;
        }
    } else {
        const uint32_t base = (uint32_t)(y0 * (int)a.pitch) + xoff;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const bool yv = (unsigned)(y0 + r) < (unsigned)a.plane_h;
            const uint32_t off = base + (uint32_t)r * (uint32_t)a.pitch;
            const u32x2 l = __builtin_amdgcn_raw_buffer_load_b64(rs, (yv && lv) ? off : kOob, 0, aux_bits<NT>());
            const u32x2 rr = __builtin_amdgcn_raw_buffer_load_b64(rs, (yv && rv) ? off + 8u : kOob, 0, aux_bits<NT>());
            W[r][0] = l.x; W[r][1] = l.y; W[r][2] = rr.x; W[r][3] = rr.y;
        }
    }

    if constexpr (MODE == 0) {
        const dbk::BlockBs bs = load_bs_buffer<EDGE>(a, f, by, bx, active);
        dbk::packed_filter_luma_block16(W, bs, a.tc, a.beta, a.max_v);
    }

    if constexpr (!EDGE) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            u32x4 w;
            w.x = W[r][0]; w.y = W[r][1]; w.z = W[r][2]; w.w = W[r][3];
            __builtin_amdgcn_raw_buffer_store_b128(w, rd, xoff, (y0 + r) * (int)a.pitch, aux_bits<NT>());
        }
    } else {
        const uint32_t base = (uint32_t)(y0 * (int)a.pitch) + xoff;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const bool yv = (unsigned)(y0 + r) < (unsigned)a.plane_h;
            const uint32_t off = base + (uint32_t)r * (uint32_t)a.pitch;
            u32x2 l, rr;
            l.x = W[r][0]; l.y = W[r][1]; rr.x = W[r][2]; rr.y = W[r][3];
            __builtin_amdgcn_raw_buffer_store_b64(l, rd, (yv && lv) ? off : kOob, 0, aux_bits<NT>());
            __builtin_amdgcn_raw_buffer_store_b64(rr, rd, (yv && rv) ? off + 8u : kOob, 0, aux_bits<NT>());
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
        c.by = blockIdx.x;
        c.f = blockIdx.y;
        c.bx = blockIdx.z * (int)blockDim.x + (int)threadIdx.x;
        const int wave_bx0 = __builtin_amdgcn_readfirstlane(c.bx) & ~63;
        c.active = c.bx < a.nbx;
        c.interior = wave_bx0 > 0 && wave_bx0 + 64 <= a.nbx - 1 && c.by > 0 && c.by < a.nby - 1;
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

template <int MODE, bool NT, bool LINEAR>
__global__ __launch_bounds__(1024) void dbk_packed16_kernel(const DbkArgs a)
{
    WaveCoords c;
    if (!wave_coords<LINEAR>(a, c)) return;
    if (c.interior) packed16_body<MODE, NT, false>(a, c.by, c.f, c.bx, true);
    else packed16_body<MODE, NT, true>(a, c.by, c.f, c.bx, c.active);
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
template <bool CHROMA, int MODE, bool NT, bool LINEAR>
__global__ __launch_bounds__(1024) void dbk_packed_kernel(const DbkArgs a)
{
    WaveCoords c;
    if (!wave_coords<LINEAR>(a, c)) return;
    if (c.interior) packed_body<CHROMA, MODE, NT, false>(a, c.by, c.f, c.bx, true);
    else packed_body<CHROMA, MODE, NT, true>(a, c.by, c.f, c.bx, c.active);
}

} /* namespace */

bool dbk_packed_supports(const DbkArgs &a, int sample_bytes, bool chroma)
{
    if (a.qp_map != nullptr) return false;                       /* scalar QP only */
    if (sample_bytes == 1) return a.max_v == 255;                /* 8-bit: luma and chroma */
    /* 16-bit containers: luma, and only while every intermediate fits int16: the normal filter's
     * 9*(q0-p0) - 3*(q1-p1) + 8 needs 12*max_v + 8 <= 32767, i.e. bit depth <= 11 */
    return !chroma && a.max_v <= 2047 && a.pitch % 8 == 0 && a.frame_stride % 8 == 0 &&
           ((uintptr_t)a.src % 8) == 0 && ((uintptr_t)a.dst % 8) == 0;
}

/* development knobs: HEVCDBK_TUNE=nt selects the non-temporal variant, HEVCDBK_WG caps the workgroup width */
static bool tune_nt()
{
    static const bool v = [] { const char *e = getenv("HEVCDBK_TUNE"); return e && strstr(e, "nt") != nullptr; }();
    return v;
}
static int tune_wg_cap()
{
    static const int v = [] {
        const char *e = getenv("HEVCDBK_WG");
        int c = e ? atoi(e) / 64 * 64 : 512;
        return c < 64 ? 64 : (c > 1024 ? 1024 : c);
    }();
    return v;
}

static bool tune_rowmap()
{
    static const bool v = [] { const char *e = getenv("HEVCDBK_TUNE"); return e && strstr(e, "rowmap") != nullptr; }();
    return v;
}

template <bool NT, bool LINEAR>
static void launch_packed_t(const DbkArgs &a, int sample_bytes, bool chroma, int mode, dim3 grid, dim3 block, hipStream_t stream)
{
    if (sample_bytes == 2) {
        if (mode == 1) hipLaunchKernelGGL((dbk_packed16_kernel<1, NT, LINEAR>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((dbk_packed16_kernel<0, NT, LINEAR>), grid, block, 0, stream, a);
    } else if (mode == 1)
        hipLaunchKernelGGL((dbk_packed_kernel<false, 1, NT, LINEAR>), grid, block, 0, stream, a);
    else if (chroma)
        hipLaunchKernelGGL((dbk_packed_kernel<true, 0, NT, LINEAR>), grid, block, 0, stream, a);
    else
        hipLaunchKernelGGL((dbk_packed_kernel<false, 0, NT, LINEAR>), grid, block, 0, stream, a);
}

hipError_t dbk_launch_packed(const DbkArgs &a, int sample_bytes, bool chroma, int mode, hipStream_t stream)
{
    if (a.n_frames <= 0 || a.nbx <= 0 || a.nby <= 0) return hipSuccess;
    DbkArgs b = a;
    { const char *e = getenv("HEVCDBK_TUNE"); b.diag_ablate = e && strstr(e, "nostrong") ? 1 : (e && strstr(e, "nonormal") ? 2 : 0); }
    if (mode == 1) { /* diagnostic copy only: HEVCDBK_TUNE=align shifts the spans onto their natural alignment */
        const char *e = getenv("HEVCDBK_TUNE");
        b.diag_xshift = (e && strstr(e, "align")) ? 4 * sample_bytes : 0;
    }
    const int cap = tune_wg_cap();
    const long long nb = (long long)a.nbx * a.nby;
    /* row-major block numbering needs exact 32-bit reciprocal division: dividends < 2^32 / divisor */
    const int wg = (int)(nb < cap ? (nb + 63) / 64 * 64 : cap);
    const long long wpf = (nb + wg - 1) / wg;
    const long long total = (wpf * a.n_frames + 7) / 8 * 8; /* multiple of 8: one contiguous range per XCD */
    /* reciprocal division floor(2^32/d)+1 is exact for dividends < 2^32/d and needs d >= 2 */
    /* measured on MI355X: when a whole block row fits one workgroup (4K: 481 blocks) the row mapping wins
     * (3.96 vs 3.69 TB/s at 4K 8-bit, idle lanes included); wider rows (8K: 961 blocks) do better row-major
     * (5.06 vs 4.89 TB/s at 8K 10-bit).  HEVCDBK_TUNE=rowmap / linear force either for A/B runs. */
    const char *tune = getenv("HEVCDBK_TUNE");
    const bool want_linear = (tune && strstr(tune, "linear")) || (!tune_rowmap() && a.nbx > cap);
    const bool linear = want_linear && wpf >= 2 && a.nbx >= 2 && (nb + 1024) * a.nbx < (1ll << 32) &&
                        total * wpf < (1ll << 32) && total < (1ll << 31);
    if (linear) {
        b.nb_total = (int)nb;
        b.wpf = (int)wpf;
        { const char *e = getenv("HEVCDBK_TUNE"); b.xcd_swizzle = !(e && strstr(e, "noswz")); }
        b.magic_wpf = (uint32_t)((1ull << 32) / (unsigned long long)wpf + 1ull);
        b.magic_nbx = (uint32_t)((1ull << 32) / (unsigned long long)a.nbx + 1ull);
        dim3 block(wg, 1, 1), grid((unsigned)total, 1, 1);
        if (tune_nt()) launch_packed_t<true, true>(b, sample_bytes, chroma, mode, grid, block, stream);
        else launch_packed_t<false, true>(b, sample_bytes, chroma, mode, grid, block, stream);
        return hipGetLastError();
    }
    /* small planes / degenerate divisors: one workgroup per block row, wider rows split into cap-lane chunks */
    const int per_wg = a.nbx < cap ? a.nbx : cap;
    dim3 block((per_wg + 63) / 64 * 64, 1, 1);
    dim3 grid(a.nby, a.n_frames, (a.nbx + (int)block.x - 1) / (int)block.x);
    if (tune_nt()) launch_packed_t<true, false>(b, sample_bytes, chroma, mode, grid, block, stream);
    else launch_packed_t<false, false>(b, sample_bytes, chroma, mode, grid, block, stream);
    return hipGetLastError();
}
