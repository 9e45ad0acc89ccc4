/*
 * sao_packed.h -- sample adaptive offset (H.265 8.7.3) of one 8x8 block of 8-bit samples held in registers, packed-int16
 * arithmetic (see sao.hip for the scheme).  Shared by the SAO pass (sao.hip) and the fused deblocking + SAO kernel
 * (deblock_sao_fused.inc): the block's ten rows y0-1 .. y0+8 arrive as SaoRaw (samples x-4 .. x+11 each) through a fetch functor
 * -- from the workgroup's LDS tile, row by row as they are needed -- and the eight output rows leave through a store functor.  Device code only.
 */
#pragma once
#include <stdint.h>

#include <type_traits>

#include "deblock_kernels.h"

namespace sao8 {

struct SaoRow {
    uint32_t E0, O0, E1, O1; /* the row's samples as int16 pairs: (s0, s2), (s1, s3), (s4, s6), (s5, s7) */
    uint32_t lE0, lE1;       /* (s[-1], s1), (s3, s5): left neighbours of E0, E1 */
    uint32_t rO0, rO1;       /* (s2, s4), (s6, s8): right neighbours of O0, O1 */
};

/* a row as it comes from memory: samples x-4 .. x+11 (lh | cx cy | rh); lh / rh are only looked at by the classes with
 * horizontal neighbours */
struct SaoRaw {
    uint32_t lh, cx, cy, rh;
};

template <bool HALO>
__device__ __forceinline__ SaoRow unpack(const SaoRaw &q)
{
    SaoRow r;
    r.E0 = __builtin_amdgcn_perm(q.cx, q.cx, 0x0c020c00u);
    r.O0 = __builtin_amdgcn_perm(q.cx, q.cx, 0x0c030c01u);
    r.E1 = __builtin_amdgcn_perm(q.cy, q.cy, 0x0c020c00u);
    r.O1 = __builtin_amdgcn_perm(q.cy, q.cy, 0x0c030c01u);
    if constexpr (HALO) {
        r.lE0 = __builtin_amdgcn_perm(q.cx, q.lh, 0x0c050c03u); /* (lh.b3, cx.b1) */
        r.lE1 = __builtin_amdgcn_perm(q.cy, q.cx, 0x0c050c03u); /* (cx.b3, cy.b1) */
        r.rO0 = __builtin_amdgcn_perm(q.cy, q.cx, 0x0c040c02u); /* (cx.b2, cy.b0) */
        r.rO1 = __builtin_amdgcn_perm(q.rh, q.cy, 0x0c040c02u); /* (cy.b2, rh.b0) */
    } else {
        r.lE0 = r.lE1 = r.rO0 = r.rO1 = 0u;
    }
    return r;
}

typedef short spk __attribute__((vector_size(4)));
typedef unsigned short supk __attribute__((vector_size(4)));
__device__ __forceinline__ spk s_pk(uint32_t v) { return __builtin_bit_cast(spk, v); }
__device__ __forceinline__ spk s_splat(int v) { return spk{(short)v, (short)v}; }
__device__ __forceinline__ supk s_upk(uint32_t v) { return __builtin_bit_cast(supk, v); }
/* max(a - b, 0) in both halves: ONE v_pk_sub_u16 with the clamp bit (unsigned saturation) */
__device__ __forceinline__ supk s_sub_sat(supk a, supk b) { return __builtin_elementwise_sub_sat(a, b); }

/* a table index (0..4) in each half as a v_perm_b32 selector: byte 0 / 2 = the index, byte 1 / 3 = 0x0c (constant zero) */
constexpr uint32_t kSel = 0x0c000c00u;

/* rec + offset[index], clipped to 8 bit: tab_lo / tab_hi hold the five offset bytes + 128, `sel` = the indices of both
 * halves + kSel.  rec + t is non-negative, so the lower clip is the saturation of the unsigned subtraction of the bias */
__device__ __forceinline__ uint32_t apply(uint32_t rec, uint32_t sel, uint32_t tab_lo, uint32_t tab_hi)
{
    const uint32_t t = __builtin_amdgcn_perm(tab_hi, tab_lo, sel);
    const supk v = s_sub_sat(s_upk(rec) + s_upk(t), supk{128, 128});
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(v, supk{255, 255}));
}
/* edge index of both samples of `rec` against the neighbour pairs a and b, as a selector (+ kSel): 0..4, 2 = neither minimum nor maximum;
 * clamp(rec + 1 - a, 0, 2) = min(saturating (rec + 1) - a, 2) */
__device__ __forceinline__ uint32_t edge_idx(uint32_t rec, uint32_t a, uint32_t b)
{
    const supk r1 = s_upk(rec) + supk{1, 1}, two = supk{2, 2};
    /* the halves hold 0..2 each: the sum and the selector constant are ONE 32-bit three-operand add (written out: hipcc sees
     * that no bits overlap, turns the second add into an OR with a literal and then cannot merge the two) */
    uint32_t r;
    asm("v_add3_u32 %0, %1, %2, %3"
        : "=v"(r)
        : "v"(__builtin_bit_cast(uint32_t, __builtin_elementwise_min(s_sub_sat(r1, s_upk(a)), two))),
          "v"(__builtin_bit_cast(uint32_t, __builtin_elementwise_min(s_sub_sat(r1, s_upk(b)), two))), "s"(kSel));
    return r;
}
/* band index of both samples of `rec` as a selector: min(((rec >> shift) - pos) & 31, 4) + kSel, the constant OR-ed in by
 * the mask instruction (v_and_or_b32, written out for the same reason) and carried through the minimum */
__device__ __forceinline__ uint32_t band_sel(uint32_t rec, int shift, spk pos)
{
    uint32_t k;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(k) : "v"(__builtin_bit_cast(uint32_t, (s_pk(rec) >> shift) - pos)), "s"(0x001f001fu), "v"(kSel));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(s_upk(k), s_upk(kSel | 0x00040004u)));
}

/* the NROWS (8) output rows of an edge-offset block from its NROWS + 2 (ten) raw rows, class CLS of Table 8-13: 0 (-1,0)/(1,0); 1 (0,-1)/(0,1);
 * 2 (-1,-1)/(1,1); 3 (1,-1)/(-1,1).  BORDER: the block may touch the picture border (x, y0 = its position, w x h the picture):
 * a sample with a neighbour outside the picture gets no offset (8.7.3.2), whatever the raw rows hold there.
 * store(r, lo, hi) takes output row r as its two dwords. */
template <int CLS, bool BORDER, int NROWS, typename Fetch, typename Store>
__device__ __forceinline__ void edge_rows(const Fetch &fetch, const Store &store, int x, int y0, int w, int h, uint32_t tab_lo,
                                          uint32_t tab_hi)
{
    constexpr bool horizontal = CLS != 1, vertical = CLS != 0;
    constexpr std::bool_constant<horizontal> halo{}; /* tells the fetch functor whether the samples left / right of the block are looked at */
    SaoRow up = unpack<horizontal>(fetch(0, halo)), mid = unpack<horizontal>(fetch(1, halo)), dn;
#pragma unroll
    for (int r = 0; r < NROWS; r++) {
        const int y = y0 + r;
        dn = unpack<horizontal>(fetch(r + 2, halo));
        uint32_t i0, i1, i2, i3; /* indices of E0, O0, E1, O1 */
        if constexpr (CLS == 0) {
            i0 = edge_idx(mid.E0, mid.lE0, mid.O0);
            i1 = edge_idx(mid.O0, mid.E0, mid.rO0);
            i2 = edge_idx(mid.E1, mid.lE1, mid.O1);
            i3 = edge_idx(mid.O1, mid.E1, mid.rO1);
        } else if constexpr (CLS == 1) {
            i0 = edge_idx(mid.E0, up.E0, dn.E0);
            i1 = edge_idx(mid.O0, up.O0, dn.O0);
            i2 = edge_idx(mid.E1, up.E1, dn.E1);
            i3 = edge_idx(mid.O1, up.O1, dn.O1);
        } else if constexpr (CLS == 2) {
            i0 = edge_idx(mid.E0, up.lE0, dn.O0);
            i1 = edge_idx(mid.O0, up.E0, dn.rO0);
            i2 = edge_idx(mid.E1, up.lE1, dn.O1);
            i3 = edge_idx(mid.O1, up.E1, dn.rO1);
        } else {
            i0 = edge_idx(mid.E0, up.O0, dn.lE0);
            i1 = edge_idx(mid.O0, up.rO0, dn.E0);
            i2 = edge_idx(mid.E1, up.O1, dn.lE1);
            i3 = edge_idx(mid.O1, up.rO1, dn.E1);
        }
        if constexpr (BORDER) {
            if (vertical && (y == 0 || y == h - 1)) i0 = i1 = i2 = i3 = 0x00020002u | sao8::kSel;
            if (horizontal && x == 0) i0 = (i0 & 0xffff0000u) | 0x0c02u;              /* sample 0: low half of E0 */
            if (horizontal && x + 8 == w) i3 = (i3 & 0x0000ffffu) | 0x0c020000u; /* sample 7: high half of O1 */
        }
        const uint32_t e0 = apply(mid.E0, i0, tab_lo, tab_hi), o0 = apply(mid.O0, i1, tab_lo, tab_hi);
        const uint32_t e1 = apply(mid.E1, i2, tab_lo, tab_hi), o1 = apply(mid.O1, i3, tab_lo, tab_hi);
        store(r, e0 | (o0 << 8), e1 | (o1 << 8));
        up = mid;
        mid = dn;
    }
}

/* one block of 8 x NROWS samples inside one CTB: not applied / kept (copy), band offset, or edge offset.  fetch(i) returns raw
 * row i = image row y0 - 1 + i (i = 0 .. NROWS + 1; the first and last are asked for by the edge classes only) -- from registers,
 * or from LDS as the rows are needed; its second argument (std::true_type / std::false_type) says whether the halo samples left
 * and right of the block will be looked at (a functor that reads memory can fetch less when they are not).  NROWS = 8: a lane's block is the keep map's unit and a wave (64 lanes) covers 64 x 64
 * samples -- one CTB of 64, i.e. ONE path per wave; NROWS = 2: a wave covers 32 x 32 samples, one CTB of 32 (every chroma CTB
 * of a 4:2:0 picture with 64-sample luma CTBs) -- the same, where 8-row lanes would spread a wave over four CTBs and run every
 * path that occurs among them with a quarter of its lanes. */
template <bool BORDER, int NROWS = 8, typename Fetch, typename Store>
__device__ __forceinline__ void block(const Fetch &fetch, const Store &store, int x, int y0, int w, int h, const DbkSaoCtb &c, bool kept)
{
    if (kept || c.type == 0 || c.type > 2) {
#pragma unroll
        for (int r = 0; r < NROWS; r++) {
            const SaoRaw q = fetch(r + 1, std::false_type{});
            store(r, q.cx, q.cy);
        }
        return;
    }
    auto b = [](int v) { return (uint32_t)(v + 128) & 0xffu; };
    if (c.type == 1) { /* band offset: bandTable[(k + sao_band_position) & 31] = k + 1; index min(k, 4), entry 4 = no offset */
        const uint32_t tab_lo = b(c.offset[0]) | (b(c.offset[1]) << 8) | (b(c.offset[2]) << 16) | (b(c.offset[3]) << 24), tab_hi = b(0);
        const spk pos = s_splat((int)c.cls);
        auto band = [&](uint32_t rec) {
            return apply(rec, band_sel(rec, 3, pos), tab_lo, tab_hi); /* 8 bit: bandShift = bitDepth - 5 = 3 */
        };
#pragma unroll
        for (int r = 0; r < NROWS; r++) {
            const SaoRow m = unpack<false>(fetch(r + 1, std::false_type{}));
            store(r, band(m.E0) | (band(m.O0) << 8), band(m.E1) | (band(m.O1) << 8));
        }
        return;
    }
    /* edge offset: index 0 -> SaoOffsetVal[1], 1 -> [2], 2 -> none, 3 -> [3], 4 -> [4] */
    const uint32_t tab_lo = b(c.offset[0]) | (b(c.offset[1]) << 8) | (b(0) << 16) | (b(c.offset[2]) << 24), tab_hi = b(c.offset[3]);
    const int cls = c.cls & 3;
    if (cls == 0) edge_rows<0, BORDER, NROWS>(fetch, store, x, y0, w, h, tab_lo, tab_hi);
    else if (cls == 1) edge_rows<1, BORDER, NROWS>(fetch, store, x, y0, w, h, tab_lo, tab_hi);
    else if (cls == 2) edge_rows<2, BORDER, NROWS>(fetch, store, x, y0, w, h, tab_lo, tab_hi);
    else edge_rows<3, BORDER, NROWS>(fetch, store, x, y0, w, h, tab_lo, tab_hi);
}

} /* namespace sao8 */

/*
 * 16-bit containers (bit depth 8..12): the same block procedure on samples that already are int16 pairs.  A row's eight
 * samples are four dwords P0..P3 = (s0,s1) (s2,s3) (s4,s5) (s6,s7); the pairs shifted one sample left are L0 = (s[-1],s0),
 * L1 = (s1,s2), L2 = (s3,s4), L3 = (s5,s6) -- one v_alignbit_b32 each -- and shifted right R0 = L1, R1 = L2, R2 = L3,
 * R3 = (s7,s8).  Offsets travel as bytes biased by 128 exactly as in the 8-bit form (the parameter entries are int8).  Raw row = samples x-4 .. x+11 as eight dwords (two 16-byte LDS reads); d[1] holds s[-1], d[6] holds s8.
 */
namespace sao16 {

using sao8::supk;
using sao8::spk;
using sao8::s_upk;
using sao8::s_pk;
using sao8::s_sub_sat;

struct Raw {
    uint32_t d[8];
};
struct Row {
    uint32_t P0, P1, P2, P3; /* the row's samples */
    uint32_t L0, L1, L2, L3; /* shifted one sample left; Rk = L(k+1) */
    uint32_t R3;
};
template <bool HALO>
__device__ __forceinline__ Row unpack(const Raw &q)
{
    Row r;
    r.P0 = q.d[2]; r.P1 = q.d[3]; r.P2 = q.d[4]; r.P3 = q.d[5];
    if constexpr (HALO) {
        r.L0 = __builtin_amdgcn_alignbit(q.d[2], q.d[1], 16); /* (d1.hi, d2.lo) = (s[-1], s0) */
        r.L1 = __builtin_amdgcn_alignbit(q.d[3], q.d[2], 16);
        r.L2 = __builtin_amdgcn_alignbit(q.d[4], q.d[3], 16);
        r.L3 = __builtin_amdgcn_alignbit(q.d[5], q.d[4], 16);
        r.R3 = __builtin_amdgcn_alignbit(q.d[6], q.d[5], 16); /* (s7, s8) */
    } else {
        r.L0 = r.L1 = r.L2 = r.L3 = r.R3 = 0u;
    }
    return r;
}

struct Tab {
    uint32_t lo, hi; /* the five offset bytes + 128 */
    uint32_t maxv;   /* max_v in both halves */
};
__device__ __forceinline__ uint32_t apply(uint32_t rec, uint32_t idx /* + sao8::kSel */, const Tab &t)
{
    const uint32_t o = __builtin_amdgcn_perm(t.hi, t.lo, idx);
    const supk v = s_sub_sat(s_upk(rec) + s_upk(o), supk{128, 128});
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(v, s_upk(t.maxv)));
}
using sao8::edge_idx;

template <int CLS, bool BORDER, int NROWS, typename Fetch, typename Store>
__device__ __forceinline__ void edge_rows(const Fetch &fetch, const Store &store, int x, int y0, int w, int h, const Tab &t)
{
    constexpr bool horizontal = CLS != 1, vertical = CLS != 0;
    constexpr std::bool_constant<horizontal> halo{};
    Row up = unpack<horizontal>(fetch(0, halo)), mid = unpack<horizontal>(fetch(1, halo)), dn;
#pragma unroll
    for (int r = 0; r < NROWS; r++) {
        const int y = y0 + r;
        dn = unpack<horizontal>(fetch(r + 2, halo));
        uint32_t i0, i1, i2, i3;
        if constexpr (CLS == 0) {        /* (-1, 0) / (1, 0) */
            i0 = edge_idx(mid.P0, mid.L0, mid.L1);
            i1 = edge_idx(mid.P1, mid.L1, mid.L2);
            i2 = edge_idx(mid.P2, mid.L2, mid.L3);
            i3 = edge_idx(mid.P3, mid.L3, mid.R3);
        } else if constexpr (CLS == 1) { /* (0, -1) / (0, 1) */
            i0 = edge_idx(mid.P0, up.P0, dn.P0);
            i1 = edge_idx(mid.P1, up.P1, dn.P1);
            i2 = edge_idx(mid.P2, up.P2, dn.P2);
            i3 = edge_idx(mid.P3, up.P3, dn.P3);
        } else if constexpr (CLS == 2) { /* (-1, -1) / (1, 1) */
            i0 = edge_idx(mid.P0, up.L0, dn.L1);
            i1 = edge_idx(mid.P1, up.L1, dn.L2);
            i2 = edge_idx(mid.P2, up.L2, dn.L3);
            i3 = edge_idx(mid.P3, up.L3, dn.R3);
        } else {                         /* (1, -1) / (-1, 1) */
            i0 = edge_idx(mid.P0, up.L1, dn.L0);
            i1 = edge_idx(mid.P1, up.L2, dn.L1);
            i2 = edge_idx(mid.P2, up.L3, dn.L2);
            i3 = edge_idx(mid.P3, up.R3, dn.L3);
        }
        if constexpr (BORDER) { /* a neighbour outside the picture: edgeIdx 0 (8.7.3.2) */
            if (vertical && (y == 0 || y == h - 1)) i0 = i1 = i2 = i3 = 0x00020002u | sao8::kSel;
            if (horizontal && x == 0) i0 = (i0 & 0xffff0000u) | 0x0c02u;              /* sample 0: low half of P0 */
            if (horizontal && x + 8 == w) i3 = (i3 & 0x0000ffffu) | 0x0c020000u; /* sample 7: high half of P3 */
        }
        store(r, apply(mid.P0, i0, t), apply(mid.P1, i1, t), apply(mid.P2, i2, t), apply(mid.P3, i3, t));
        up = mid;
        mid = dn;
    }
}

/* one block of 8 x NROWS 16-bit samples (NROWS = 8, or 2 for 32-sample CTBs: see sao8::block); fetch(i) = raw row i = image
 * row y0 - 1 + i; store(r, four dwords) */
template <bool BORDER, int NROWS = 8, typename Fetch, typename Store>
__device__ __forceinline__ void block(const Fetch &fetch, const Store &store, int x, int y0, int w, int h, const DbkSaoCtb &c, bool kept,
                                      int max_v, int band_shift)
{
    if (kept || c.type == 0 || c.type > 2) {
#pragma unroll
        for (int r = 0; r < NROWS; r++) {
            const Raw q = fetch(r + 1, std::false_type{});
            store(r, q.d[2], q.d[3], q.d[4], q.d[5]);
        }
        return;
    }
    /* the entries are SaoOffsetVal[1..4] themselves (already scaled to the bit depth by the caller, |v| <= 127): a byte biased by 128 */
    auto b = [](int v) { return (uint32_t)(v + 128) & 0xffu; };
    Tab t;
    t.maxv = (uint32_t)max_v * 0x00010001u;
    if (c.type == 1) { /* band offset: bandTable[(k + sao_band_position) & 31] = k + 1; index min(k, 4), entry 4 = no offset */
        t.lo = b(c.offset[0]) | (b(c.offset[1]) << 8) | (b(c.offset[2]) << 16) | (b(c.offset[3]) << 24);
        t.hi = b(0);
        const spk pos = sao8::s_splat((int)c.cls);
        auto band = [&](uint32_t rec) {
            return apply(rec, sao8::band_sel(rec, band_shift, pos), t);
        };
#pragma unroll
        for (int r = 0; r < NROWS; r++) {
            const Raw q = fetch(r + 1, std::false_type{});
            store(r, band(q.d[2]), band(q.d[3]), band(q.d[4]), band(q.d[5]));
        }
        return;
    }
    /* edge offset: index 0 -> SaoOffsetVal[1], 1 -> [2], 2 -> none, 3 -> [3], 4 -> [4] */
    t.lo = b(c.offset[0]) | (b(c.offset[1]) << 8) | (0x80u << 16) | (b(c.offset[2]) << 24);
    t.hi = b(c.offset[3]);
    const int cls = c.cls & 3;
    if (cls == 0) edge_rows<0, BORDER, NROWS>(fetch, store, x, y0, w, h, t);
    else if (cls == 1) edge_rows<1, BORDER, NROWS>(fetch, store, x, y0, w, h, t);
    else if (cls == 2) edge_rows<2, BORDER, NROWS>(fetch, store, x, y0, w, h, t);
    else edge_rows<3, BORDER, NROWS>(fetch, store, x, y0, w, h, t);
}

} /* namespace sao16 */

