/*
 * sao.hip -- sample adaptive offset (ITU-T H.265 clause 8.7.3), the in-loop stage after deblocking.
 * Not present in the reference.  One lane = four consecutive samples of one row (one aligned memory word), a wave =
 * 256 consecutive samples; src -> dst because the edge classifier must see deblocked, not offset, neighbours.  The two
 * neighbour rows and the two neighbour words come from L1/L2 (each sample is fetched from HBM once per pass).
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

/* samples x-1 .. x+4 of one row (x a multiple of 4); positions outside the row hold junk the caller never uses */
template <typename T>
__device__ __forceinline__ void load6(const uint8_t *row, int x, int w, int (&o)[6])
{
    using W = typename Q4<T>::W;
    int c[4], l[4], r[4];
    Q4<T>::unpack(*reinterpret_cast<const W *>(row + (size_t)x * sizeof(T)), c);
    Q4<T>::unpack(*reinterpret_cast<const W *>(row + (size_t)(x >= 4 ? x - 4 : 0) * sizeof(T)), l);
    Q4<T>::unpack(*reinterpret_cast<const W *>(row + (size_t)(x + 8 <= w ? x + 4 : w - 4) * sizeof(T)), r);
    o[0] = l[3]; o[1] = c[0]; o[2] = c[1]; o[3] = c[2]; o[4] = c[3]; o[5] = r[0];
}

__device__ __forceinline__ int sgn(int v) { return (v > 0) - (v < 0); }

template <typename T>
__global__ __launch_bounds__(256) void sao_kernel(const DbkSaoArgs a)
{
    using W = typename Q4<T>::W;
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int y = blockIdx.y, f = blockIdx.z;
    if (x >= a.plane_w) return;
    const uint8_t *src = a.src + (long long)f * a.frame_stride;
    uint8_t *dst = a.dst + (long long)f * a.frame_stride;
    const DbkSaoCtb c = a.params[(long long)f * a.params_frame_stride + (long long)(y >> a.ctb_log2) * a.params_stride + (x >> a.ctb_log2)];
    const bool kept = a.keep && a.keep[(long long)f * a.keep_frame_stride + (long long)(y >> 3) * a.keep_stride + (x >> 3)];
    const uint8_t *row = src + (long long)y * a.pitch;
    W *out = reinterpret_cast<W *>(dst + (long long)y * a.pitch + (size_t)x * sizeof(T));
    if (kept || c.type == 0 || c.type > 2) {
        *out = *reinterpret_cast<const W *>(row + (size_t)x * sizeof(T));
        return;
    }
    int o[4];
    if (c.type == 1) { /* band offset: bandTable[(k + sao_band_position) & 31] = k + 1 */
        Q4<T>::unpack(*reinterpret_cast<const W *>(row + (size_t)x * sizeof(T)), o);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int k = ((o[i] >> a.band_shift) - (int)c.cls) & 31;
            const int off = k == 0 ? c.offset[0] : (k == 1 ? c.offset[1] : (k == 2 ? c.offset[2] : (k == 3 ? c.offset[3] : 0)));
            const int v = o[i] + off;
            o[i] = v < 0 ? 0 : (v > a.max_v ? a.max_v : v);
        }
        *out = Q4<T>::pack(o);
        return;
    }
    /* edge offset, Table 8-13: class 0 (-1,0)/(1,0); 1 (0,-1)/(0,1); 2 (-1,-1)/(1,1); 3 (1,-1)/(-1,1) */
    const int cls = c.cls & 3;
    const int dxa = cls == 1 ? 0 : (cls == 3 ? 1 : -1), dya = cls == 0 ? 0 : -1;
    int m[6], ra[6], rb[6];
    load6<T>(row, x, a.plane_w, m);
    const int ya = y + dya, yb = y - dya;
    const bool rows_ok = ya >= 0 && yb < a.plane_h;
    load6<T>(src + (long long)(ya < 0 ? 0 : ya) * a.pitch, x, a.plane_w, ra);
    load6<T>(src + (long long)(yb >= a.plane_h ? a.plane_h - 1 : yb) * a.pitch, x, a.plane_w, rb);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int rec = m[1 + i];
        const int xa = x + i + dxa, xb = x + i - dxa;
        const bool ok = rows_ok && xa >= 0 && xa < a.plane_w && xb >= 0 && xb < a.plane_w;
        const int na = dxa < 0 ? ra[i] : (dxa == 0 ? ra[1 + i] : ra[2 + i]);
        const int nb = dxa < 0 ? rb[2 + i] : (dxa == 0 ? rb[1 + i] : rb[i]);
        int e = 2 + sgn(rec - na) + sgn(rec - nb);
        /* raw 0 -> SaoOffsetVal[1], 1 -> [2], 2 -> none, 3 -> [3], 4 -> [4] */
        const int off = e == 0 ? c.offset[0] : (e == 1 ? c.offset[1] : (e == 3 ? c.offset[2] : (e == 4 ? c.offset[3] : 0)));
        const int v = rec + (ok ? off : 0);
        o[i] = v < 0 ? 0 : (v > a.max_v ? a.max_v : v);
    }
    *out = Q4<T>::pack(o);
}

} /* namespace */

hipError_t dbk_launch_sao(const DbkSaoArgs &a, int sample_bytes, hipStream_t stream)
{
    if (a.n_frames <= 0 || a.plane_w <= 0 || a.plane_h <= 0) return hipSuccess;
    dim3 block(256, 1, 1), grid((a.plane_w / 4 + 255) / 256, a.plane_h, a.n_frames);
    if (sample_bytes == 1) hipLaunchKernelGGL(sao_kernel<uint8_t>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(sao_kernel<uint16_t>, grid, block, 0, stream, a);
    return hipGetLastError();
}
