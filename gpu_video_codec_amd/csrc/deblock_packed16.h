/*
 * deblock_packed16.h -- packed-int16 block arithmetic on 16-bit containers beyond reference-mode luma (which lives in
 * deblock_packed.h): reference-mode chroma, and the spec-exact mode's luma and chroma.  W[r][j] = columns (2j, 2j+1) of
 * row r as two uint16.  Samples are already 16 bit wide, so the horizontal chroma edges need no unpacking at all: a
 * dword IS a pair of two lines (adjacent columns).
 *
 * Value ranges: the luma normal filter's 9*(q0-p0) - 3*(q1-p1) + 8 needs 12*max_v + 8 <= 32767 (bit depth <= 11; 12 bit
 * takes the WIDE evaluation); chroma's 4*(p0-q0) + p1 - q1 + 4 needs 5*max_v + 4 <= 32767 (bit depth <= 12).
 */
#pragma once
#include "deblock_packed_h265.h"

namespace dbk {

DBK_HD uint32_t lo_hi(uint32_t lo_of, uint32_t hi_of) { return perm(hi_of, lo_of, 0x07060100u); } /* (lo_of.lo, hi_of.hi) */

/* ---- reference-exact chroma (cpu.h:450-993, 1431-1488) ---- */

template <int R0>
DBK_HD void chroma_ver16(uint32_t (&W)[8][4], int tc, int max_v)
{
    const pk c = splat(tc);
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int ra = R0 + (h ? 1 : 0), rb = R0 + (h ? 2 : 3);
        pk p0 = pick_hi(bits_pk(W[ra][1]), bits_pk(W[rb][1])), p1 = pick_lo(bits_pk(W[ra][1]), bits_pk(W[rb][1])); /* cols 3, 2 */
        pk q0 = pick_lo(bits_pk(W[ra][2]), bits_pk(W[rb][2])), q1 = pick_hi(bits_pk(W[ra][2]), bits_pk(W[rb][2])); /* cols 4, 5 */
        chroma_pair(p0, p1, q0, q1, c, max_v);
        W[ra][1] = pk_bits(pick_lo(bits_pk(W[ra][1]), p0));  /* (col 2, p0 of row ra) */
        W[rb][1] = lo_hi(W[rb][1], pk_bits(p0));              /* (col 2, p0 of row rb) */
        W[ra][2] = lo_hi(pk_bits(q0), W[ra][2]);              /* (q0 of row ra, col 5) */
        W[rb][2] = pk_bits(pick_hi(q0, bits_pk(W[rb][2])));   /* (q0 of row rb, col 5) */
    }
}

/* P = rows 3, 2 of column pair JP; Q = rows 4, 5 of column pair JQ */
template <int JP, int JQ>
DBK_HD void chroma_hor16(uint32_t (&W)[8][4], int tc, int max_v)
{
    const pk c = splat(tc);
    pk p0 = bits_pk(W[3][JP]), q0 = bits_pk(W[4][JQ]);
    chroma_pair(p0, bits_pk(W[2][JP]), q0, bits_pk(W[5][JQ]), c, max_v);
    W[3][JP] = pk_bits(p0);
    W[4][JQ] = pk_bits(q0);
}

DBK_HD void packed_filter_chroma_block16(uint32_t (&W)[8][4], const BlockBs &bs, const BlockQp &q, int max_v)
{
    if (bs.ver1 == 2) chroma_ver16<0>(W, q.tc[0], max_v);
    if (bs.ver2 == 2) chroma_ver16<4>(W, q.tc[1], max_v);
    if (bs.hor1 == 2) { chroma_hor16<0, 0>(W, q.tc[2], max_v); chroma_hor16<1, 1>(W, q.tc[2], max_v); }
    /* hor2 (cpu.h:644-718): P columns 4..7, Q columns 0..3 */
    if (bs.hor2 == 2) { chroma_hor16<2, 0>(W, q.tc[3], max_v); chroma_hor16<3, 1>(W, q.tc[3], max_v); }
}

/* ---- spec-exact mode (H.265 8.7.2) ---- */

template <int R0>
DBK_HD void chroma_ver16_h265(uint32_t (&W)[8][4], int tc, int entry, int max_v)
{
    if ((entry & kH265BsMask) != 2) return;
    const pk c = splat(tc), mp = splat((entry & kH265KeepP) ? 0 : -1), mq = splat((entry & kH265KeepQ) ? 0 : -1);
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int ra = R0 + (h ? 1 : 0), rb = R0 + (h ? 2 : 3);
        pk p0 = pick_hi(bits_pk(W[ra][1]), bits_pk(W[rb][1])), p1 = pick_lo(bits_pk(W[ra][1]), bits_pk(W[rb][1]));
        pk q0 = pick_lo(bits_pk(W[ra][2]), bits_pk(W[rb][2])), q1 = pick_hi(bits_pk(W[ra][2]), bits_pk(W[rb][2]));
        chroma_pair_h265(p0, p1, q0, q1, c, mp, mq, max_v);
        W[ra][1] = pk_bits(pick_lo(bits_pk(W[ra][1]), p0));
        W[rb][1] = lo_hi(W[rb][1], pk_bits(p0));
        W[ra][2] = lo_hi(pk_bits(q0), W[ra][2]);
        W[rb][2] = pk_bits(pick_hi(q0, bits_pk(W[rb][2])));
    }
}

/* both column pairs (J0, J0+1) of one horizontal segment */
template <int J0>
DBK_HD void chroma_hor16_h265(uint32_t (&W)[8][4], int tc, int entry, int max_v)
{
    if ((entry & kH265BsMask) != 2) return;
    const pk c = splat(tc), mp = splat((entry & kH265KeepP) ? 0 : -1), mq = splat((entry & kH265KeepQ) ? 0 : -1);
#pragma unroll
    for (int j = J0; j < J0 + 2; j++) {
        pk p0 = bits_pk(W[3][j]), q0 = bits_pk(W[4][j]);
        chroma_pair_h265(p0, bits_pk(W[2][j]), q0, bits_pk(W[5][j]), c, mp, mq, max_v);
        W[3][j] = pk_bits(p0);
        W[4][j] = pk_bits(q0);
    }
}

/* WIDE = 12 bit: the strong filter's sums (8*max_v + 4 = 32,764) still fit int16 here -- this mode keeps the plain clip
 * ranges -- so only the normal filter's delta needs its split evaluation */
template <bool WIDE = false, bool TAB = false>
DBK_HD void packed_filter_luma_block16_h265(uint32_t (&W)[8][4], const H265Seg &s, int max_v, const H265Uni *u = nullptr)
{
    Taps va1 = unpack_ver16(W[0], W[3]), vb1 = unpack_ver16(W[1], W[2]);
    Taps va2 = unpack_ver16(W[4], W[7]), vb2 = unpack_ver16(W[5], W[6]);
    const bool keep_any = h265_keep_any(s);
    luma_seg_h265<WIDE, TAB>(va1, vb1, s, 0, max_v, u, keep_any);
    luma_seg_h265<WIDE, TAB>(va2, vb2, s, 1, max_v, u, keep_any);
    Taps ha, hb, ga, gb;
    ha.p0 = pick_hi(va1.p3, va1.p0); hb.p0 = pick_hi(va1.p2, va1.p1);
    ha.p1 = pick_hi(vb1.p3, vb1.p0); hb.p1 = pick_hi(vb1.p2, vb1.p1);
    ha.p2 = pick_lo(vb1.p3, vb1.p0); hb.p2 = pick_lo(vb1.p2, vb1.p1);
    ha.p3 = pick_lo(va1.p3, va1.p0); hb.p3 = pick_lo(va1.p2, va1.p1);
    ha.q0 = pick_lo(va2.p3, va2.p0); hb.q0 = pick_lo(va2.p2, va2.p1);
    ha.q1 = pick_lo(vb2.p3, vb2.p0); hb.q1 = pick_lo(vb2.p2, vb2.p1);
    ha.q2 = pick_hi(vb2.p3, vb2.p0); hb.q2 = pick_hi(vb2.p2, vb2.p1);
    ha.q3 = pick_hi(va2.p3, va2.p0); hb.q3 = pick_hi(va2.p2, va2.p1);
    luma_seg_h265<WIDE, TAB>(ha, hb, s, 2, max_v, u, keep_any);
    ga.p0 = pick_hi(va1.q0, va1.q3); gb.p0 = pick_hi(va1.q1, va1.q2);
    ga.p1 = pick_hi(vb1.q0, vb1.q3); gb.p1 = pick_hi(vb1.q1, vb1.q2);
    ga.p2 = pick_lo(vb1.q0, vb1.q3); gb.p2 = pick_lo(vb1.q1, vb1.q2);
    ga.p3 = pick_lo(va1.q0, va1.q3); gb.p3 = pick_lo(va1.q1, va1.q2);
    ga.q0 = pick_lo(va2.q0, va2.q3); gb.q0 = pick_lo(va2.q1, va2.q2);
    ga.q1 = pick_lo(vb2.q0, vb2.q3); gb.q1 = pick_lo(vb2.q1, vb2.q2);
    ga.q2 = pick_hi(vb2.q0, vb2.q3); gb.q2 = pick_hi(vb2.q1, vb2.q2);
    ga.q3 = pick_hi(va2.q0, va2.q3); gb.q3 = pick_hi(va2.q1, va2.q2);
    luma_seg_h265<WIDE, TAB>(ga, gb, s, 3, max_v, u, keep_any);
    /* pair A = cols (0,3) / (4,7), pair B = cols (1,2) / (5,6):  (c0,c1) = (A.lo,B.lo), (c2,c3) = (B.hi,A.hi) */
#define DBK_ROW16H(r, A, B, j)            \
    W[r][j] = pk_bits(pick_lo(A, B));      \
    W[r][j + 1] = pk_bits(pick_hi(B, A));
    DBK_ROW16H(0, ha.p3, hb.p3, 0) DBK_ROW16H(1, ha.p2, hb.p2, 0) DBK_ROW16H(2, ha.p1, hb.p1, 0) DBK_ROW16H(3, ha.p0, hb.p0, 0)
    DBK_ROW16H(4, ha.q0, hb.q0, 0) DBK_ROW16H(5, ha.q1, hb.q1, 0) DBK_ROW16H(6, ha.q2, hb.q2, 0) DBK_ROW16H(7, ha.q3, hb.q3, 0)
    DBK_ROW16H(0, ga.p3, gb.p3, 2) DBK_ROW16H(1, ga.p2, gb.p2, 2) DBK_ROW16H(2, ga.p1, gb.p1, 2) DBK_ROW16H(3, ga.p0, gb.p0, 2)
    DBK_ROW16H(4, ga.q0, gb.q0, 2) DBK_ROW16H(5, ga.q1, gb.q1, 2) DBK_ROW16H(6, ga.q2, gb.q2, 2) DBK_ROW16H(7, ga.q3, gb.q3, 2)
#undef DBK_ROW16H
}

template <bool CHROMA, bool WIDE = false, bool TAB = false>
DBK_HD void packed_filter_block16_h265(uint32_t (&W)[8][4], const H265Seg &s, int max_v, const H265Uni *u = nullptr)
{
    if constexpr (CHROMA) {
        chroma_ver16_h265<0>(W, s.tc[0], s.entry[0], max_v);
        chroma_ver16_h265<4>(W, s.tc[1], s.entry[1], max_v);
        chroma_hor16_h265<0>(W, s.tc[2], s.entry[2], max_v);
        chroma_hor16_h265<2>(W, s.tc[3], s.entry[3], max_v);
    } else {
        packed_filter_luma_block16_h265<WIDE, TAB>(W, s, max_v, u);
    }
}

} /* namespace dbk */
