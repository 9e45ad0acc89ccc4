/*
 * deblock_h265.hip -- gfx950 kernels of the spec-exact mode (ITU-T H.265 clause 8.7.2):
 * the block filter, the bS derivation (8.7.2.4) and the 4:2:0 chroma bS gather.
 *
 * Same mapping as the reference-exact kernels: one lane owns one offset 8x8 block, a wave owns 64 consecutive blocks
 * of a block row, rows outside the picture are never loaded or stored.  Arithmetic is the 32-bit form of
 * deblock_h265.h for every operand kind (8/16-bit samples, luma/chroma, scalar QP or QP map).
 */
#include <hip/hip_runtime.h>

#include "deblock_h265.h"
#include "deblock_kernels.h"

namespace {

template <typename T>
struct Quad4; /* 4 consecutive samples as one memory word */
template <>
struct Quad4<uint8_t> {
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
struct Quad4<uint16_t> {
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

template <typename T, bool CHROMA>
__global__ __launch_bounds__(256) void dbk_h265_kernel(const DbkH265Args h)
{
    using Q = Quad4<T>;
    using W = typename Q::W;
    const DbkArgs &a = h.base;
    const int bx = blockIdx.x * 64 + threadIdx.x;
    const int by = blockIdx.y * 4 + threadIdx.y;
    const int f = blockIdx.z;
    if (bx >= a.nbx || by >= a.nby) return;

    int entry[4];
    dbk::load_block_bs_h265(a.vert_bs + (long long)f * a.vert_bs_stride, a.hor_bs + (long long)f * a.hor_bs_stride, bx, by,
                            a.nbx, a.nby, a.vstride, a.hstride, entry);
    /* chroma ignores bS 1 (8.7.2.5): blocks with nothing to filter move no samples at all when filtering in place */
    bool any = false;
#pragma unroll
    for (int s = 0; s < 4; s++) any |= CHROMA ? (entry[s] & dbk::kH265BsMask) == 2 : (entry[s] & dbk::kH265BsMask) != 0;
    if (!any && a.src == a.dst) return;

    const uint8_t *src = a.src + (long long)f * a.frame_stride;
    uint8_t *dst = a.dst + (long long)f * a.frame_stride;
    const int x0 = bx * 8 - 4, y0 = by * 8 - 4;
    const bool lv = bx > 0, rv = bx < a.nbx - 1;

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

    if (any) {
        int qpl[4];
        const int sc = CHROMA ? 2 : 1;
        dbk::h265_block_qpl(a.qp_map ? a.qp_map + (long long)f * a.map_frame_stride : nullptr, a.map_stride, a.ctu_log2, sc,
                            a.plane_w * sc, a.plane_h * sc, x0, y0, h.qp, qpl);
        const dbk::H265Prm prm = {h.tc_off, h.beta_off, h.c_qp_offset, a.shift, a.max_v};
        dbk::filter_block_h265<CHROMA>(v, entry, qpl, prm);
    }

#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int y = y0 + r;
        const bool rowv = (unsigned)y < (unsigned)a.plane_h;
        uint8_t *row = dst + (long long)y * a.pitch + (long long)x0 * (int)sizeof(T);
        if (rowv && lv) *reinterpret_cast<W *>(row) = Q::pack(v[r][0], v[r][1], v[r][2], v[r][3]);
        if (rowv && rv) *reinterpret_cast<W *>(row + 4 * sizeof(T)) = Q::pack(v[r][4], v[r][5], v[r][6], v[r][7]);
    }
}

/* 8.7.2.4: one thread per bS entry; blockIdx.y = 0 vertical edges, 1 horizontal edges */
__global__ __launch_bounds__(256) void dbk_h265_bs_kernel(const dbk::H265Units u, int w, int h, uint8_t *vert, uint8_t *hor)
{
    const long long uw = w / 4;
    const int vstride = w / 8 + 1, hstride = w / 4;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.y == 0) {
        if (i >= (long long)vstride * (h / 4)) return;
        const int y4 = (int)(i / vstride), bx = (int)(i % vstride);
        const bool edge = bx > 0 && bx < w / 8; /* x = 0 and x = w: picture boundary */
        vert[i] = edge ? (uint8_t)dbk::h265_bs_of_edge(u, y4 * uw + 2 * bx - 1, y4 * uw + 2 * bx, true) : (uint8_t)0;
    } else {
        if (i >= (long long)(h / 8 + 1) * hstride) return;
        const int by = (int)(i / hstride), x4 = (int)(i % hstride);
        const bool edge = by > 0 && by < h / 8;
        hor[i] = edge ? (uint8_t)dbk::h265_bs_of_edge(u, (2 * by - 1) * uw + x4, (2 * by) * uw + x4, false) : (uint8_t)0;
    }
}

/* 4:2:0 chroma arrays = the luma entry at twice the chroma position (8.7.2.5: bS[xDk * SubWidthC][yDm * SubHeightC]) */
__global__ __launch_bounds__(256) void dbk_h265_chroma_bs_kernel(const uint8_t *vert, const uint8_t *hor, int w, int h,
                                                                 uint8_t *cvert, uint8_t *chor)
{
    const int cw = w / 2, ch = h / 2;
    const int vstride = w / 8 + 1, hstride = w / 4, cvstride = cw / 8 + 1, chstride = cw / 4;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.y == 0) {
        if (i >= (long long)cvstride * (ch / 4)) return;
        const int m = (int)(i / cvstride), bx = (int)(i % cvstride);
        cvert[i] = vert[(long long)(2 * m) * vstride + 2 * bx];
    } else {
        if (i >= (long long)(ch / 8 + 1) * chstride) return;
        const int by = (int)(i / chstride), m = (int)(i % chstride);
        chor[i] = hor[(long long)(2 * by) * hstride + 2 * m];
    }
}

template <typename T, bool CHROMA>
hipError_t launch_t(const DbkH265Args &h, hipStream_t stream)
{
    dim3 block(64, 4, 1);
    dim3 grid((h.base.nbx + 63) / 64, (h.base.nby + 3) / 4, h.base.n_frames);
    hipLaunchKernelGGL((dbk_h265_kernel<T, CHROMA>), grid, block, 0, stream, h);
    return hipGetLastError();
}

} /* namespace */

hipError_t dbk_launch_h265(const DbkH265Args &h, int sample_bytes, bool chroma, hipStream_t stream)
{
    if (h.base.n_frames <= 0 || h.base.nbx <= 0 || h.base.nby <= 0) return hipSuccess;
    if (sample_bytes == 1) return chroma ? launch_t<uint8_t, true>(h, stream) : launch_t<uint8_t, false>(h, stream);
    return chroma ? launch_t<uint16_t, true>(h, stream) : launch_t<uint16_t, false>(h, stream);
}

hipError_t dbk_launch_h265_bs(const void *flags, const void *mv0, const void *mv1, const void *ref0, const void *ref1, int w,
                              int h, uint8_t *vert, uint8_t *hor, uint8_t *cvert, uint8_t *chor, hipStream_t stream)
{
    const dbk::H265Units u = {(const uint16_t *)flags, (const int16_t *)mv0, (const int16_t *)mv1, (const int32_t *)ref0,
                              (const int32_t *)ref1};
    const long long nv = (long long)(w / 8 + 1) * (h / 4), nh = (long long)(h / 8 + 1) * (w / 4);
    const long long n = nv > nh ? nv : nh;
    hipLaunchKernelGGL(dbk_h265_bs_kernel, dim3((unsigned)((n + 255) / 256), 2, 1), dim3(256), 0, stream, u, w, h, vert, hor);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !cvert || !chor) return e;
    return dbk_launch_h265_chroma_bs(vert, hor, w, h, cvert, chor, stream);
}

hipError_t dbk_launch_h265_chroma_bs(const uint8_t *vert, const uint8_t *hor, int w, int h, uint8_t *cvert, uint8_t *chor,
                                     hipStream_t stream)
{
    const int cw = w / 2, ch = h / 2;
    const long long nv = (long long)(cw / 8 + 1) * (ch / 4), nh = (long long)(ch / 8 + 1) * (cw / 4);
    const long long n = nv > nh ? nv : nh;
    hipLaunchKernelGGL(dbk_h265_chroma_bs_kernel, dim3((unsigned)((n + 255) / 256), 2, 1), dim3(256), 0, stream, vert, hor, w, h,
                       cvert, chor);
    return hipGetLastError();
}
