/*
 * sao.hip -- sample adaptive offset (ITU-T H.265 clause 8.7.3), the in-loop stage after deblocking.
 * Not present in the reference.  One lane = one 8x8 block of samples, a wave = one 64x64 region; src -> dst
 * because the edge classifier must see deblocked, not offset, neighbours.  The halo (one row above and below, one sample
 * left and right) comes from L1/L2: each sample is fetched from HBM once per pass.
 */
#include <hip/hip_runtime.h>

#include "deblock_kernels.h"
#include "sao_packed.h"

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
 * Workgroup -> (256 x 64 strip, frame).  SWZ: a 1-D grid whose workgroups are renumbered so that each XCD (workgroups are
 * dealt to the 8 XCDs round-robin -- an observation used for speed only) works through a CONTIGUOUS range of strips,
 * row-major inside a frame.  The halo of an edge-offset CTB -- the 16-byte row loads reach 4 bytes into the strip to the
 * left and right, the three-row window one row into the strips above and below -- then lies in lines a neighbouring
 * workgroup of the SAME XCD fetches as well, i.e. in that XCD's L2, instead of being fetched from HBM once per XCD
 * (round 2: 1.48 x the plane's bytes read for one third edge CTBs; the same renumbering took the fused deblocking + SAO
 * kernel from 2.0 x to 1.004 x).  Grids too large for the exact 32-bit reciprocal divisions keep the 3-D numbering.
 */
template <bool SWZ>
__device__ __forceinline__ bool sao_strip(const DbkFusedGrid &g, int &wx, int &wy, int &f)
{
    if constexpr (!SWZ) {
        wx = blockIdx.x; wy = blockIdx.y; f = blockIdx.z;
        return true;
    } else {
        const uint32_t id = blockIdx.x;
        const uint32_t logical = (id & 7u) * g.per_xcd + (id >> 3);
        if (logical >= g.total) return false; /* padding workgroup */
        const uint32_t fr = g.tiles_per_frame == 1u ? logical : __umulhi(logical, g.magic_tpf);
        const uint32_t in_frame = logical - fr * g.tiles_per_frame;
        const uint32_t row = g.tiles_x == 1u ? in_frame : __umulhi(in_frame, g.magic_tx);
        wx = (int)(in_frame - row * g.tiles_x); wy = (int)row; f = (int)fr;
        return true;
    }
}

/*
 * One lane = one 8x8 block of samples (the unit of the keep map; inside one CTB since CTBs are at least 8 samples): the
 * CTB parameters and the keep flag are fetched once, and the edge classifier slides a three-row window down the block, so a
 * row is loaded once per lane (10 rows for 8 rows of output) instead of three times per output row.
 */
template <typename T, bool SWZ, bool PK16 = false> /* PK16: 16-bit containers up to 12 bit take the packed block procedure in waves off the border */
__global__ __launch_bounds__(256) void sao_kernel(const DbkSaoArgs a, const DbkFusedGrid g)
{
    /* a wave = the 8 x 8 blocks of one 64 x 64 region: with 64-sample CTBs every lane of a wave has the same SAO type and the
     * wave runs ONE of the three paths; a row-shaped wave (512 x 8) would span eight CTBs and run all of them */
    int wx, wy, f;
    if (!sao_strip<SWZ>(g, wx, wy, f)) return;
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int x = (wx * 4 + wv) * 64 + (l & 7) * 8;
    const int y0 = wy * 64 + (l >> 3) * 8;
    if (x >= a.plane_w || y0 >= a.plane_h) return;
    const uint8_t *src = a.src + (long long)f * a.frame_stride;
    uint8_t *dst = a.dst + (long long)f * a.frame_stride;
    const DbkSaoCtb c = a.params[(long long)f * a.params_frame_stride + (long long)(y0 >> a.ctb_log2) * a.params_stride + (x >> a.ctb_log2)];
    const bool kept = a.keep && a.keep[(long long)f * a.keep_frame_stride + (long long)(y0 >> 3) * a.keep_stride + (x >> 3)];
    if constexpr (PK16 && sizeof(T) == 2) {
        /* no lane of the wave on the picture border (nearly every wave): the packed 16-bit block procedure of the fused kernels
         * (sao_packed.h, sao16: the samples already are int16 pairs) on rows addressed through buffer resources, as the 8-bit
         * kernel below does; a region's row piece is a whole 128-byte line here, so the wave keeps its 64 x 64 shape */
        const bool border = x == 0 || x + 8 == a.plane_w || y0 == 0 || y0 + 8 >= a.plane_h;
        if (__builtin_amdgcn_ballot_w64(border) == 0ull) {
            typedef uint32_t u32x4b __attribute__((ext_vector_type(4)));
            const uint32_t plane_bytes = (uint32_t)a.pitch * (uint32_t)a.plane_h; /* < 2^31: checked by the launcher */
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, plane_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, plane_bytes, 0x00020000);
            const int sp = __builtin_amdgcn_readfirstlane((int)a.pitch);
            const uint32_t vrow = (uint32_t)y0 * (uint32_t)a.pitch + (uint32_t)x * 2u;
            const uint32_t vup = vrow - (uint32_t)a.pitch; /* raw row 0 = image row y0 - 1 */
            auto fetch = [&](int j, auto halo) {
                sao16::Raw q;
                const u32x4b m = __builtin_amdgcn_raw_buffer_load_b128(rs, vup, j * sp, 0); /* samples x .. x+7 */
                q.d[0] = q.d[7] = 0u;
                q.d[2] = m.x; q.d[3] = m.y; q.d[4] = m.z; q.d[5] = m.w;
                if constexpr (decltype(halo)::value) {
                    q.d[1] = __builtin_amdgcn_raw_buffer_load_b32(rs, vup - 4u, j * sp, 0);  /* s[-2], s[-1] */
                    q.d[6] = __builtin_amdgcn_raw_buffer_load_b32(rs, vup + 16u, j * sp, 0); /* s8, s9 */
                } else {
                    q.d[1] = q.d[6] = 0u;
                }
                return q;
            };
            auto store = [&](int r, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3) {
                u32x4b w;
                w.x = d0; w.y = d1; w.z = d2; w.w = d3;
                __builtin_amdgcn_raw_buffer_store_b128(w, rd, vrow, r * sp, 0);
                /* the wait states of the fused 16-bit kernel's stores (deblock_sao_fused.inc): a 16-byte buffer store with an SGPR
                 * offset followed at once by a VALU write of its data registers */
                asm volatile("s_nop 1" : : "v"(w.x), "v"(w.y), "v"(w.z), "v"(w.w) : "memory");
            };
            sao16::block<false, 8>(fetch, store, x, y0, a.plane_w, a.plane_h, c, kept, a.max_v, a.band_shift);
            return;
        }
    }
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

/* ------------------------------------------------------------------------------------------ */
/* 8-bit samples: packed-int16 arithmetic, table look-ups by v_perm_b32                         */
/*
 * Same mapping (one lane = one 8x8 block, a wave = one 64x64 region, so with 64-sample CTBs a wave runs ONE path) and the
 * same sliding three-row window, but two samples per VGPR and no per-sample selects:
 *   - a row's 8 samples are four int16 pairs E0 = (s0, s2), O0 = (s1, s3), E1 = (s4, s6), O1 = (s5, s7), one v_perm_b32
 *     each from the row's two dwords; the same instruction, fed the halo dword, makes the pairs shifted one sample left
 *     (lE0, lE1; the left neighbours of O0 / O1 are E0 / E1 themselves) and right (rO0, rO1): every neighbour pair of
 *     Table 8-13 is one of these eight registers of the row above, the row itself or the row below;
 *   - sign(rec - a) + sign(rec - b) + 2 = clamp(rec + 1 - a, 0, 2) + clamp(rec + 1 - b, 0, 2) in both halves at once;
 *   - that index (0..4), or min(bandIdx, 4) for the band offset, selects one of five offset bytes through v_perm_b32 (the
 *     bytes are biased by 128 so that they are plain unsigned bytes; the bias rides on rec);
 *   - picture-border samples (a neighbour outside the picture: no offset) force the index to "none" in the lanes / rows
 *     concerned, compiled only into the instantiation for waves that touch the border.
 * About 9 VALU instructions per sample instead of ~20.
 */
using sao8::SaoRow;
using sao8::SaoRaw;
using sao8::spk;
using sao8::supk;
using sao8::s_pk;
using sao8::s_splat;

typedef uint32_t sao_u32x4 __attribute__((ext_vector_type(4), aligned(4))); /* 16 bytes at a 4-byte-aligned address */
typedef uint32_t sao_u32x2b __attribute__((ext_vector_type(2)));             /* operands of the raw buffer builtins */
typedef uint32_t sao_u32x4b __attribute__((ext_vector_type(4)));

/* HALO: the neighbours to the left / right take part.  INNER (wave-uniform): every lane's 16 bytes x-4 .. x+11 lie inside
 * the row, so the row is ONE 16-byte load per lane; otherwise three loads with the halo positions moved inside the row */
template <bool HALO, bool INNER>
__device__ __forceinline__ SaoRow sao_load_row(const uint8_t *row, int x, int w)
{
    SaoRaw q;
    q.lh = q.rh = 0u;
    if constexpr (HALO && INNER) {
        const sao_u32x4 v = *reinterpret_cast<const sao_u32x4 *>(row + x - 4);
        q.lh = v.x; q.cx = v.y; q.cy = v.z; q.rh = v.w;
    } else {
        const uint2 c = *reinterpret_cast<const uint2 *>(row + x);
        q.cx = c.x; q.cy = c.y;
        if constexpr (HALO) {
            /* positions outside the row read a valid dword of the row instead; the border masks discard what comes of it */
            q.lh = *reinterpret_cast<const uint32_t *>(row + (x >= 4 ? x - 4 : 0));
            q.rh = *reinterpret_cast<const uint32_t *>(row + (x + 12 <= w ? x + 8 : w - 4));
        }
    }
    return sao8::unpack<HALO>(q);
}

__device__ __forceinline__ uint32_t sao_apply(uint32_t rec, uint32_t idx, uint32_t tab_lo, uint32_t tab_hi)
{
    return sao8::apply(rec, idx, tab_lo, tab_hi);
}
__device__ __forceinline__ uint32_t sao_edge_idx(uint32_t rec, uint32_t a, uint32_t b) { return sao8::edge_idx(rec, a, b); }

template <bool BORDER>
__device__ __forceinline__ void sao8_edge_block(const DbkSaoArgs &a, const uint8_t *src, uint8_t *dst, int x, int y0, int cls,
                                                uint32_t tab_lo, uint32_t tab_hi)
{
    auto row_at = [&](int y) { return src + (long long)(y < 0 ? 0 : (y >= a.plane_h ? a.plane_h - 1 : y)) * a.pitch; };
    const bool horizontal = cls != 1; /* neighbours to the left / right take part: the halo dwords are needed */
    const bool vertical = cls != 0;   /* neighbours in the rows above / below take part */
    SaoRow up, mid, dn;
    if (horizontal) {
        up = sao_load_row<true, !BORDER>(row_at(y0 - 1), x, a.plane_w);
        mid = sao_load_row<true, !BORDER>(row_at(y0), x, a.plane_w);
    } else {
        up = sao_load_row<false, false>(row_at(y0 - 1), x, a.plane_w);
        mid = sao_load_row<false, false>(row_at(y0), x, a.plane_w);
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int y = y0 + r;
        dn = horizontal ? sao_load_row<true, !BORDER>(row_at(y + 1), x, a.plane_w) : sao_load_row<false, false>(row_at(y + 1), x, a.plane_w);
        uint32_t i0, i1, i2, i3; /* indices of E0, O0, E1, O1 */
        if (cls == 0) {          /* (-1, 0) / (1, 0) */
            i0 = sao_edge_idx(mid.E0, mid.lE0, mid.O0);
            i1 = sao_edge_idx(mid.O0, mid.E0, mid.rO0);
            i2 = sao_edge_idx(mid.E1, mid.lE1, mid.O1);
            i3 = sao_edge_idx(mid.O1, mid.E1, mid.rO1);
        } else if (cls == 1) {   /* (0, -1) / (0, 1) */
            i0 = sao_edge_idx(mid.E0, up.E0, dn.E0);
            i1 = sao_edge_idx(mid.O0, up.O0, dn.O0);
            i2 = sao_edge_idx(mid.E1, up.E1, dn.E1);
            i3 = sao_edge_idx(mid.O1, up.O1, dn.O1);
        } else if (cls == 2) {   /* (-1, -1) / (1, 1) */
            i0 = sao_edge_idx(mid.E0, up.lE0, dn.O0);
            i1 = sao_edge_idx(mid.O0, up.E0, dn.rO0);
            i2 = sao_edge_idx(mid.E1, up.lE1, dn.O1);
            i3 = sao_edge_idx(mid.O1, up.E1, dn.rO1);
        } else {                 /* (1, -1) / (-1, 1) */
            i0 = sao_edge_idx(mid.E0, up.O0, dn.lE0);
            i1 = sao_edge_idx(mid.O0, up.rO0, dn.E0);
            i2 = sao_edge_idx(mid.E1, up.O1, dn.lE1);
            i3 = sao_edge_idx(mid.O1, up.rO1, dn.E1);
        }
        if constexpr (BORDER) { /* a neighbour outside the picture: edgeIdx 0 (8.7.3.2) */
            if (vertical && (y == 0 || y == a.plane_h - 1)) i0 = i1 = i2 = i3 = 0x00020002u | sao8::kSel;
            if (horizontal && x == 0) i0 = (i0 & 0xffff0000u) | 0x0c02u;                       /* sample 0: low half of E0 */
            if (horizontal && x + 8 == a.plane_w) i3 = (i3 & 0x0000ffffu) | 0x0c020000u;  /* sample 7: high half of O1 */
        }
        const uint32_t e0 = sao_apply(mid.E0, i0, tab_lo, tab_hi), o0 = sao_apply(mid.O0, i1, tab_lo, tab_hi);
        const uint32_t e1 = sao_apply(mid.E1, i2, tab_lo, tab_hi), o1 = sao_apply(mid.O1, i3, tab_lo, tab_hi);
        uint2 out;
        out.x = e0 | (o0 << 8);
        out.y = e1 | (o1 << 8);
        *reinterpret_cast<uint2 *>(dst + (long long)y * a.pitch + x) = out;
        up = mid;
        mid = dn;
    }
}

template <bool SWZ, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void sao8_kernel(const DbkSaoArgs a, const DbkFusedGrid g)
{
    int wx, wy, f;
    if (!sao_strip<SWZ>(g, wx, wy, f)) return;
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    /* Lane -> 8 x 8 block.  A wave takes one 64 x 64 region, i.e. with 64-sample CTBs one CTB and ONE path; its row pieces are
     * then 64 bytes, half a cache line.  Where the two CTBs of an aligned pair (waves 2k, 2k + 1 of the workgroup) take the
     * same path -- both edge offset of one class, or neither edge offset: SAO parameters are merged from the left / above
     * neighbour in most CTBs of a real stream -- the two waves split the pair the other way: each takes 32 rows of BOTH CTBs
     * (16 blocks across, 4 down), whole 128-byte lines, still one path per wave.  (Round 3 measured the wide shape for every
     * pair: +8 % where the paths agree, -7 % where they differ; per pair it only ever takes the gain.) */
    int x = (wx * WAVES + wv) * 64 + (l & 7) * 8;
    int y0 = wy * 64 + (l >> 3) * 8;
    bool zero_band = false;
    if constexpr (WAVES % 2 == 0) {
        const int px = (wx * WAVES + (wv & ~1)) * 64, py = wy * 64; /* the pair's origin */
        if (a.ctb_log2 == 6 && px + 128 <= a.plane_w && py + 64 <= a.plane_h) {
            const DbkSaoCtb *pc = a.params + (long long)f * a.params_frame_stride + (long long)(py >> 6) * a.params_stride + (px >> 6);
            /* "not applied" and band offset count as one path: in a wide wave the former runs as a band offset of zeros (below) */
            const bool e0 = pc[0].type == 2, e1 = pc[1].type == 2;
            const bool same = e0 == e1 && (!e0 || ((pc[0].cls ^ pc[1].cls) & 3) == 0);
            if (__builtin_amdgcn_readfirstlane(same ? 1 : 0)) { /* uniform by construction: every lane looked at the same two entries */
                x = px + (l & 15) * 8;
                y0 = py + (wv & 1) * 32 + (l >> 4) * 8;
                /* one CTB band offset, the other not applied: see below */
                zero_band = __builtin_amdgcn_readfirstlane((!e0 && (pc[0].type == 1) != (pc[1].type == 1)) ? 1 : 0) != 0;
            }
        }
    }
    if (x >= a.plane_w || y0 >= a.plane_h) return;
    const uint8_t *src = a.src + (long long)f * a.frame_stride;
    uint8_t *dst = a.dst + (long long)f * a.frame_stride;
    const DbkSaoCtb c = a.params[(long long)f * a.params_frame_stride + (long long)(y0 >> a.ctb_log2) * a.params_stride + (x >> a.ctb_log2)];
    const bool kept = a.keep && a.keep[(long long)f * a.keep_frame_stride + (long long)(y0 >> 3) * a.keep_stride + (x >> 3)];
    const bool border = x == 0 || x + 8 == a.plane_w || y0 == 0 || y0 + 8 >= a.plane_h;
    if (__builtin_amdgcn_ballot_w64(border) == 0ull) {
        /* no lane of the wave touches the picture border (nearly every wave): the shared block procedure (sao_packed.h, the
         * edge class resolved once per block) on rows addressed through buffer resources -- a lane's byte offset once, the
         * row in the scalar offset, no per-row 64-bit address arithmetic and no clamping of row numbers */
        const uint32_t plane_bytes = (uint32_t)a.pitch * (uint32_t)a.plane_h; /* < 2^31: checked by the launcher */
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, plane_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, plane_bytes, 0x00020000);
        const int sp = __builtin_amdgcn_readfirstlane((int)a.pitch);
        const uint32_t vrow = (uint32_t)y0 * (uint32_t)a.pitch + (uint32_t)x; /* (x, y0); y0 >= 8 and x >= 8 here */
        const uint32_t vup = vrow - (uint32_t)a.pitch;                         /* raw row 0 = image row y0 - 1 */
        auto fetch = [&](int j, auto halo) {
            SaoRaw q;
            if constexpr (decltype(halo)::value) {
                const sao_u32x4b v = __builtin_amdgcn_raw_buffer_load_b128(rs, vup - 4u, j * sp, 0);
                q.lh = v.x; q.cx = v.y; q.cy = v.z; q.rh = v.w;
            } else {
                const sao_u32x2b v = __builtin_amdgcn_raw_buffer_load_b64(rs, vup, j * sp, 0);
                q.lh = q.rh = 0u;
                q.cx = v.x; q.cy = v.y;
            }
            return q;
        };
        auto store = [&](int r, uint32_t lo, uint32_t hi) {
            sao_u32x2b w;
            w.x = lo;
            w.y = hi;
            __builtin_amdgcn_raw_buffer_store_b64(w, rd, vrow, r * sp, 0);
        };
        if (zero_band) {
            /* a wide wave over one band-offset CTB and one without SAO: the latter's blocks (and kept ones) run as a band offset
             * of zeros (rec + 0, clipped: the same bytes) so that the two CTBs' lanes issue the SAME loads and stores -- whole
             * lines -- instead of each half of the wave its own.  (A pair without SAO in either CTB keeps the plain copy:
             * the arithmetic costs 5 % there.) */
            DbkSaoCtb z = c;
            if (kept || c.type != 1) {
                z.type = 1; z.cls = 0;
                z.offset[0] = z.offset[1] = z.offset[2] = z.offset[3] = 0;
            }
            sao8::block<false, 8>(fetch, store, x, y0, a.plane_w, a.plane_h, z, false);
            return;
        }
        sao8::block<false, 8>(fetch, store, x, y0, a.plane_w, a.plane_h, c, kept);
        return;
    }
    if (kept || c.type == 0 || c.type > 2) {
#pragma unroll
        for (int r = 0; r < 8; r++)
            *reinterpret_cast<uint2 *>(dst + (long long)(y0 + r) * a.pitch + x) =
                *reinterpret_cast<const uint2 *>(src + (long long)(y0 + r) * a.pitch + x);
        return;
    }
    auto b = [](int v) { return (uint32_t)(v + 128) & 0xffu; };
    if (c.type == 1) { /* band offset: bandTable[(k + sao_band_position) & 31] = k + 1; index min(k, 4), entry 4 = no offset */
        const uint32_t tab_lo = b(c.offset[0]) | (b(c.offset[1]) << 8) | (b(c.offset[2]) << 16) | (b(c.offset[3]) << 24), tab_hi = b(0);
        const spk pos = s_splat((int)c.cls);
        auto band = [&](uint32_t rec) {
            return sao_apply(rec, sao8::band_sel(rec, 3, pos), tab_lo, tab_hi); /* 8 bit: bandShift = bitDepth - 5 = 3 */
        };
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const SaoRow m = sao_load_row<false, false>(src + (long long)(y0 + r) * a.pitch, x, a.plane_w);
            uint2 out;
            out.x = band(m.E0) | (band(m.O0) << 8);
            out.y = band(m.E1) | (band(m.O1) << 8);
            *reinterpret_cast<uint2 *>(dst + (long long)(y0 + r) * a.pitch + x) = out;
        }
        return;
    }
    /* edge offset: index 0 -> SaoOffsetVal[1], 1 -> [2], 2 -> none, 3 -> [3], 4 -> [4] */
    const uint32_t tab_lo = b(c.offset[0]) | (b(c.offset[1]) << 8) | (b(0) << 16) | (b(c.offset[2]) << 24), tab_hi = b(c.offset[3]);
    sao8_edge_block<true>(a, src, dst, x, y0, c.cls & 3, tab_lo, tab_hi); /* a wave with a lane on the picture border */
}

} /* namespace */

hipError_t dbk_launch_sao(const DbkSaoArgs &a, int sample_bytes, hipStream_t stream)
{
    if (a.n_frames <= 0 || a.plane_w <= 0 || a.plane_h <= 0) return hipSuccess;
    int waves = 4; /* 64 x 64 regions (waves) side by side in one workgroup */
#ifdef HEVCDBK_DIAG
    if (g_dbk_diag.wg_cap == 64 || g_dbk_diag.wg_cap == 128) waves = g_dbk_diag.wg_cap / 64; /* A/B knob: narrower workgroups of the packed 8-bit kernel */
#endif
    const bool aligned8 = a.pitch % 8 == 0 && a.frame_stride % 8 == 0 && ((uintptr_t)a.src % 8) == 0 && ((uintptr_t)a.dst % 8) == 0 &&
                          a.plane_w % 8 == 0 && a.plane_h % 8 == 0 && a.max_v == 255 && a.band_shift == 3 &&
                          (unsigned long long)a.pitch * (unsigned long long)a.plane_h < (1ull << 31); /* 32-bit buffer offsets */
    if (!(sample_bytes == 1 && aligned8)) waves = 4;
#ifdef HEVCDBK_DIAG
    /* the narrower strips of the wg_cap knob exist for the renumbered grid only (sao8_kernel<true, 1 | 2>): with the plain
     * 3-D numbering -- the noswz knob, or a grid too large to renumber -- the launch below is sao8_kernel<false, 4>, whose
     * strips are 256 samples wide, and block / grid must be sized for that (half or three quarters of each strip
     * row stayed unwritten when both knobs were set) */
    {
        const unsigned long long tx0 = (unsigned long long)(a.plane_w + 64 * waves - 1) / (64 * waves), tpf0 = tx0 * ((a.plane_h + 63) / 64);
        const unsigned long long tot0 = tpf0 * (unsigned long long)a.n_frames;
        const bool swz0 = !g_dbk_diag.noswz && tot0 + 8 < (1ull << 31) && (tot0 + 8) * tpf0 < (1ull << 32) && tpf0 * tx0 < (1ull << 32);
        if (!swz0) waves = 4;
    }
#endif
    const int strip_w = 64 * waves;
    const dim3 block(strip_w, 1, 1), grid3((a.plane_w + strip_w - 1) / strip_w, (a.plane_h + 63) / 64, a.n_frames);
    /* the renumbered 1-D grid (sao_strip): strips per frame and in total small enough for exact reciprocal division
     * (dividend < 2^32 / divisor) */
    DbkFusedGrid g = {};
    const unsigned long long tx = grid3.x, tpf = tx * grid3.y, total = tpf * (unsigned long long)a.n_frames;
    bool swz = total + 8 < (1ull << 31) && (total + 8) * tpf < (1ull << 32) && tpf * tx < (1ull << 32);
#ifdef HEVCDBK_DIAG
    if (g_dbk_diag.noswz) swz = false; /* A/B knob of the diagnostic library: the plain 3-D numbering */
#endif
    if (swz) {
        g.tiles_x = (uint32_t)tx;
        g.tiles_per_frame = (uint32_t)tpf;
        g.total = (uint32_t)total;
        g.magic_tpf = tpf <= 1 ? 0u : (uint32_t)((1ull << 32) / tpf + 1ull);
        g.magic_tx = tx <= 1 ? 0u : (uint32_t)((1ull << 32) / tx + 1ull);
        g.per_xcd = (uint32_t)((total + 7) / 8);
    }
    const dim3 grid = swz ? dim3(g.per_xcd * 8u, 1, 1) : grid3;
    /* 8-bit planes whose rows and frames are 8-byte aligned take the packed kernel (every lane moves 8 bytes at once) */
    if (sample_bytes == 1 && aligned8) {
#ifdef HEVCDBK_DIAG
        if (waves == 1 && swz) hipLaunchKernelGGL((sao8_kernel<true, 1>), grid, block, 0, stream, a, g);
        else if (waves == 2 && swz) hipLaunchKernelGGL((sao8_kernel<true, 2>), grid, block, 0, stream, a, g);
        else
#endif
        if (swz) hipLaunchKernelGGL((sao8_kernel<true, 4>), grid, block, 0, stream, a, g);
        else hipLaunchKernelGGL((sao8_kernel<false, 4>), grid, block, 0, stream, a, g);
    } else if (sample_bytes == 1) {
        if (swz) hipLaunchKernelGGL((sao_kernel<uint8_t, true>), grid, block, 0, stream, a, g);
        else hipLaunchKernelGGL((sao_kernel<uint8_t, false>), grid, block, 0, stream, a, g);
    } else {
        /* up to 12 bit (the packed procedure's int16 fields; SaoOffsetVal scaled as the standard does), planes the buffer
         * resources can address: the packed path in every wave off the picture border */
        const bool pk16 = a.max_v <= 4095 && a.band_shift >= 3 && (1 << (a.band_shift + 5)) - 1 == a.max_v && a.pitch % 4 == 0 &&
                          a.frame_stride % 4 == 0 && ((uintptr_t)a.src % 4) == 0 && ((uintptr_t)a.dst % 4) == 0 && a.plane_w % 8 == 0 &&
                          a.plane_h % 8 == 0 && (unsigned long long)a.pitch * (unsigned long long)a.plane_h < (1ull << 31);
        if (swz && pk16) hipLaunchKernelGGL((sao_kernel<uint16_t, true, true>), grid, block, 0, stream, a, g);
        else if (swz) hipLaunchKernelGGL((sao_kernel<uint16_t, true>), grid, block, 0, stream, a, g);
        else hipLaunchKernelGGL((sao_kernel<uint16_t, false>), grid, block, 0, stream, a, g);
    }
    return hipGetLastError();
}
