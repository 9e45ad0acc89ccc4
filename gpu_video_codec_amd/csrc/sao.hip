/*
 * sao.hip -- sample adaptive offset (ITU-T H.265 clause 8.7.3), the in-loop stage after deblocking.
 * Not present in the reference.  One lane = one 8x8 block of samples, a wave = one 64x64 region; src -> dst
 * because the edge classifier must see deblocked, not offset, neighbours.  The halo (one row above and below, one sample
 * left and right) comes from L1/L2: each sample is fetched from HBM once per pass.
 */
#include <hip/hip_runtime.h>

#include "deblock_kernels.h"

namespace {

template <typename T>
struct Q4;
template <>
struct Q4<uint8_t> {
    using W = uint32_t;
    static __device__ __forceinline__ void unpack(W w, int (&o)[4]) { o[0] = w & 0xff; o[1] = (w >> 8) & 0xff; o[2] = (w >> 16) & 0xff; o[3] = w >> 24; }
    static __device__ __forceinline__ W pack(const int (&o)[4]) { return (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3] << 24); }
};
template <>
struct Q4<uint16_t> {
    using W = uint2;
    static __device__ __forceinline__ void unpack(W w, int (&o)[4]) { o[0] = w.x & 0xffff; o[1] = w.x >> 16; o[2] = w.y & 0xffff; o[3] = w.y >> 16; }
    static __device__ __forceinline__ W pack(const int (&o)[4]) { return make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16)); }
};

/* samples x-1 .. x+8 of one row (x a multiple of 8) into o[0..9]; positions outside the row hold junk the caller never uses */
template <typename T>
__device__ __forceinline__ void load10(const uint8_t *row, int x, int w, int (&o)[10])
{
    using W = typename Q4<T>::W;
    int c0[4], c1[4], l[4], r[4];
    Q4<T>::unpack(*reinterpret_cast<const W *>(row + (size_t)x * sizeof(T)), c0);
    Q4<T>::unpack(*reinterpret_cast<const W *>(row + (size_t)(x + 4) * sizeof(T)), c1);
    Q4<T>::unpack(*reinterpret_cast<const W *>(row + (size_t)(x >= 4 ? x - 4 : 0) * sizeof(T)), l);
    Q4<T>::unpack(*reinterpret_cast<const W *>(row + (size_t)(x + 12 <= w ? x + 8 : w - 4) * sizeof(T)), r);
    o[0] = l[3];
#pragma unroll
    for (int i = 0; i < 4; i++) { o[1 + i] = c0[i]; o[5 + i] = c1[i]; }
    o[9] = r[0];
}
template <typename T>
__device__ __forceinline__ void load8(const uint8_t *row, int x, int (&o)[8])
{
    using W = typename Q4<T>::W;
    int c0[4], c1[4];
    Q4<T>::unpack(*reinterpret_cast<const W *>(row + (size_t)x * sizeof(T)), c0);
    Q4<T>::unpack(*reinterpret_cast<const W *>(row + (size_t)(x + 4) * sizeof(T)), c1);
#pragma unroll
    for (int i = 0; i < 4; i++) { o[i] = c0[i]; o[4 + i] = c1[i]; }
}
template <typename T>
__device__ __forceinline__ void store8(uint8_t *row, int x, const int (&o)[8])
{
    using W = typename Q4<T>::W;
    const int a[4] = {o[0], o[1], o[2], o[3]}, b[4] = {o[4], o[5], o[6], o[7]};
    *reinterpret_cast<W *>(row + (size_t)x * sizeof(T)) = Q4<T>::pack(a);
    *reinterpret_cast<W *>(row + (size_t)(x + 4) * sizeof(T)) = Q4<T>::pack(b);
}

__device__ __forceinline__ int sgn(int v) { return (v > 0) - (v < 0); }

/*
 * One lane = one 8x8 block of samples (the unit of the keep map; inside one CTB since CTBs are at least 8 samples): the
 * CTB parameters and the keep flag are fetched once, and the edge classifier slides a three-row window down the block, so a
 * row is loaded once per lane (10 rows for 8 rows of output) instead of three times per output row.
 */
template <typename T>
__global__ __launch_bounds__(256) void sao_kernel(const DbkSaoArgs a)
{
    /* a wave = the 8 x 8 blocks of one 64 x 64 region: with 64-sample CTBs every lane of a wave has the same SAO type and the
     * wave runs ONE of the three paths; a row-shaped wave (512 x 8) would span eight CTBs and run all of them */
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int x = (blockIdx.x * 4 + wv) * 64 + (l & 7) * 8;
    const int y0 = blockIdx.y * 64 + (l >> 3) * 8, f = blockIdx.z;
    if (x >= a.plane_w || y0 >= a.plane_h) return;
    const uint8_t *src = a.src + (long long)f * a.frame_stride;
    uint8_t *dst = a.dst + (long long)f * a.frame_stride;
    const DbkSaoCtb c = a.params[(long long)f * a.params_frame_stride + (long long)(y0 >> a.ctb_log2) * a.params_stride + (x >> a.ctb_log2)];
    const bool kept = a.keep && a.keep[(long long)f * a.keep_frame_stride + (long long)(y0 >> 3) * a.keep_stride + (x >> 3)];
    if (kept || c.type == 0 || c.type > 2) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            int o[8];
            load8<T>(src + (long long)(y0 + r) * a.pitch, x, o);
            store8<T>(dst + (long long)(y0 + r) * a.pitch, x, o);
        }
        return;
    }
    if (c.type == 1) { /* band offset: bandTable[(k + sao_band_position) & 31] = k + 1 */
#pragma unroll
        for (int r = 0; r < 8; r++) {
            int o[8];
            load8<T>(src + (long long)(y0 + r) * a.pitch, x, o);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int k = ((o[i] >> a.band_shift) - (int)c.cls) & 31;
                const int off = k == 0 ? c.offset[0] : (k == 1 ? c.offset[1] : (k == 2 ? c.offset[2] : (k == 3 ? c.offset[3] : 0)));
                const int v = o[i] + off;
                o[i] = v < 0 ? 0 : (v > a.max_v ? a.max_v : v);
            }
            store8<T>(dst + (long long)(y0 + r) * a.pitch, x, o);
        }
        return;
    }
    /* edge offset, Table 8-13: class 0 (-1,0)/(1,0); 1 (0,-1)/(0,1); 2 (-1,-1)/(1,1); 3 (1,-1)/(-1,1) */
    const int cls = c.cls & 3;
    const int dxa = cls == 1 ? 0 : (cls == 3 ? 1 : -1);
    const bool vertical = cls != 0; /* neighbours in the rows above and below */
    auto row_at = [&](int y) { return src + (long long)(y < 0 ? 0 : (y >= a.plane_h ? a.plane_h - 1 : y)) * a.pitch; };
    int up[10], mid[10], dn[10];
    load10<T>(row_at(y0 - 1), x, a.plane_w, up);
    load10<T>(row_at(y0), x, a.plane_w, mid);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int y = y0 + r;
        load10<T>(row_at(y + 1), x, a.plane_w, dn);
        const bool rows_ok = !vertical || (y > 0 && y < a.plane_h - 1);
        int o[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int rec = mid[1 + i];
            const int xa = x + i + dxa, xb = x + i - dxa;
            const bool ok = rows_ok && xa >= 0 && xa < a.plane_w && xb >= 0 && xb < a.plane_w;
            const int (&ra)[10] = vertical ? up : mid;
            const int (&rb)[10] = vertical ? dn : mid;
            const int na = dxa < 0 ? ra[i] : (dxa == 0 ? ra[1 + i] : ra[2 + i]);
            const int nb = dxa < 0 ? rb[2 + i] : (dxa == 0 ? rb[1 + i] : rb[i]);
            const int e = 2 + sgn(rec - na) + sgn(rec - nb);
            /* raw 0 -> SaoOffsetVal[1], 1 -> [2], 2 -> none, 3 -> [3], 4 -> [4] */
            const int off = e == 0 ? c.offset[0] : (e == 1 ? c.offset[1] : (e == 3 ? c.offset[2] : (e == 4 ? c.offset[3] : 0)));
            const int v = rec + (ok ? off : 0);
            o[i] = v < 0 ? 0 : (v > a.max_v ? a.max_v : v);
        }
        store8<T>(dst + (long long)y * a.pitch, x, o);
#pragma unroll
        for (int i = 0; i < 10; i++) { up[i] = mid[i]; mid[i] = dn[i]; }
    }
}

} /* namespace */

hipError_t dbk_launch_sao(const DbkSaoArgs &a, int sample_bytes, hipStream_t stream)
{
    if (a.n_frames <= 0 || a.plane_w <= 0 || a.plane_h <= 0) return hipSuccess;
    dim3 block(256, 1, 1), grid((a.plane_w + 255) / 256, (a.plane_h + 63) / 64, a.n_frames);
    if (sample_bytes == 1) hipLaunchKernelGGL(sao_kernel<uint8_t>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(sao_kernel<uint16_t>, grid, block, 0, stream, a);
    return hipGetLastError();
}
