/*
 * deblock_h265.h -- per-offset-block arithmetic of the SPEC-EXACT mode (ITU-T H.265 clause 8.7.2), SURVEY 8f rank 3.
 *
 * The reference's filter is "HEVC-style" but not conformant (SURVEY 8a Q-list: thresholds of the strong decision,
 * clip of the normal delta, chroma sign, hor2 column pairing, filtered frame edges, constant bS and QP).  This mode is
 * what a decoder needs instead; it is selected explicitly and never changes the reference-exact mode's results.
 *
 * Why the offset-block formulation is still exact: clause 8.7.2.1 filters every vertical edge of the picture, then every
 * horizontal edge on the result.  A vertical edge at x = 8k reads columns 8k-4..8k+3 and writes 8k-3..8k+2; a horizontal
 * edge at y = 8m reads rows 8m-4..8m+3.  The offset block (image rows 8by-4..8by+3, cols 8bx-4..8bx+3) therefore holds
 * everything its one vertical and one horizontal edge crossing read, and the vertically filtered samples its horizontal
 * edge needs are produced by its own vertical edge: blocks stay independent.  tests/ check this against an oracle that is
 * written in the standard's picture order (oracle/h265_oracle.c).
 *
 * DBK_HD like deblock_core.h: tests/host_sim runs the same arithmetic on the CPU against that oracle.
 */
#pragma once
#include "deblock_core.h"

namespace dbk {

/* bS entry layout (include/hevc_deblock.h): bits 1:0 bS, bit 2 keep P samples, bit 3 keep Q samples */
constexpr int kH265BsMask = 3, kH265KeepP = 4, kH265KeepQ = 8;

/* Table 8-12 */
DBK_HD int h265_beta(int q)
{
    /* 0 for Q < 16; 6..18 for 16..28; then steps of 2 up to 64 at 51 */
    return q < 16 ? 0 : (q < 29 ? q - 10 : 2 * q - 38);
}
DBK_HD int h265_tc(int q)
{
    /* packed 4-bit would not hold 24: two 32-bit words per 8 entries of 5 bits is more trouble than a byte table */
    static constexpr uint8_t t[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1,  1,  1,  1,  1,  1,  1,
                               2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};
    return t[q];
}
/* Table 8-10, ChromaArrayType == 1: qPi < 30 -> qPi; 30..43 -> 29,30,31,32,33,33,34,34,35,35,36,36,37,37; > 43 -> qPi - 6 */
DBK_HD int h265_chroma_qp(int qpi)
{
    if (qpi < 30) return qpi;
    if (qpi > 43) return qpi - 6;
    return qpi < 35 ? qpi - 1 : (qpi + 32) >> 1; /* 30..34 -> 29..33; 35..43 -> 33,34,34,35,35,36,36,37,37 */
}

struct H265Prm {
    int tc_off;      /* slice_tc_offset_div2 << 1 */
    int beta_off;    /* slice_beta_offset_div2 << 1 */
    int c_qp_offset; /* cQpPicOffset of this plane */
    int shift;       /* bit depth - 8 */
    int max_v;
};

using SegHor2S = Seg<false, 3, 4, 4>; /* the conformant hor2: P and Q both in columns 4..7 (contrast SegHor2) */

/* normal filter of one line, 8.7.2.5.7 (delta clipped to +-tc; the reference clips to +-2tc, cpu.h:1256) */
template <class S, int I>
DBK_HD void h265_normal_line(int (&v)[8][8], int tc, bool dep, bool deq, int max_v)
{
    const int p0 = S::template P<I, 0>(v), p1 = S::template P<I, 1>(v), p2 = S::template P<I, 2>(v);
    const int q0 = S::template Q<I, 0>(v), q1 = S::template Q<I, 1>(v), q2 = S::template Q<I, 2>(v);
    const int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
    if (iabs(delta) < 10 * tc) {
        const int D = clip1(delta, tc);
        S::template P<I, 0>(v) = clip2(p0 + D, max_v);
        S::template Q<I, 0>(v) = clip2(q0 - D, max_v);
        if (dep) S::template P<I, 1>(v) = clip2(p1 + clip1((((p2 + p0 + 1) >> 1) - p1 + D) >> 1, tc >> 1), max_v);
        if (deq) S::template Q<I, 1>(v) = clip2(q1 + clip1((((q2 + q0 + 1) >> 1) - q1 - D) >> 1, tc >> 1), max_v);
    }
}

/* copy the three samples a luma filter may change on one side of line I */
template <class S, int I, bool PSIDE>
DBK_HD void h265_side(int (&v)[8][8], int (&keep)[4][3], bool restore)
{
    int *a, *b, *c;
    if constexpr (PSIDE) { a = &S::template P<I, 0>(v); b = &S::template P<I, 1>(v); c = &S::template P<I, 2>(v); }
    else { a = &S::template Q<I, 0>(v); b = &S::template Q<I, 1>(v); c = &S::template Q<I, 2>(v); }
    if (restore) { *a = keep[I][0]; *b = keep[I][1]; *c = keep[I][2]; }
    else { keep[I][0] = *a; keep[I][1] = *b; keep[I][2] = *c; }
}
template <class S, bool PSIDE>
DBK_HD void h265_sides(int (&v)[8][8], int (&keep)[4][3], bool restore)
{
    h265_side<S, 0, PSIDE>(v, keep, restore);
    h265_side<S, 1, PSIDE>(v, keep, restore);
    h265_side<S, 2, PSIDE>(v, keep, restore);
    h265_side<S, 3, PSIDE>(v, keep, restore);
}

/* one luma segment: 8.7.2.5.3 (decisions), 8.7.2.5.6 (dSam), 8.7.2.5.7 (filters); qpl = (QpQ + QpP + 1) >> 1 */
template <class S>
DBK_HD void h265_luma_segment(int (&v)[8][8], int entry, int qpl, const H265Prm &p)
{
    const int bs = entry & kH265BsMask;
    if (bs == 0) return;
    const int beta = h265_beta(clampi(qpl + p.beta_off, 0, 51)) << p.shift;
    const int tc = h265_tc(clampi(qpl + 2 * (bs - 1) + p.tc_off, 0, 53)) << p.shift;
    const int dp0 = iabs(S::template P<0, 2>(v) - 2 * S::template P<0, 1>(v) + S::template P<0, 0>(v));
    const int dp3 = iabs(S::template P<3, 2>(v) - 2 * S::template P<3, 1>(v) + S::template P<3, 0>(v));
    const int dq0 = iabs(S::template Q<0, 2>(v) - 2 * S::template Q<0, 1>(v) + S::template Q<0, 0>(v));
    const int dq3 = iabs(S::template Q<3, 2>(v) - 2 * S::template Q<3, 1>(v) + S::template Q<3, 0>(v));
    if (!(dp0 + dp3 + dq0 + dq3 < beta)) return;
    const int tpq = (5 * tc + 1) >> 1;
    const bool strong =
        (2 * (dp0 + dq0) < (beta >> 2)) && (2 * (dp3 + dq3) < (beta >> 2)) &&
        (iabs(S::template P<0, 3>(v) - S::template P<0, 0>(v)) + iabs(S::template Q<0, 0>(v) - S::template Q<0, 3>(v)) < (beta >> 3)) &&
        (iabs(S::template P<3, 3>(v) - S::template P<3, 0>(v)) + iabs(S::template Q<3, 0>(v) - S::template Q<3, 3>(v)) < (beta >> 3)) &&
        (iabs(S::template P<0, 0>(v) - S::template Q<0, 0>(v)) < tpq) && (iabs(S::template P<3, 0>(v) - S::template Q<3, 0>(v)) < tpq);
    int kp[4][3], kq[4][3];
    if (entry & kH265KeepP) h265_sides<S, true>(v, kp, false);
    if (entry & kH265KeepQ) h265_sides<S, false>(v, kq, false);
    if (strong) {
        /* Clip3(p - 2tc, p + 2tc, avg) == p + Clip1(avg - p, 2tc): the same line arithmetic as the reference mode */
        luma_strong_line<S, 0>(v, 2 * tc, p.max_v);
        luma_strong_line<S, 1>(v, 2 * tc, p.max_v);
        luma_strong_line<S, 2>(v, 2 * tc, p.max_v);
        luma_strong_line<S, 3>(v, 2 * tc, p.max_v);
    } else {
        const int tside = (beta + (beta >> 1)) >> 3;
        const bool dep = dp0 + dp3 < tside, deq = dq0 + dq3 < tside;
        h265_normal_line<S, 0>(v, tc, dep, deq, p.max_v);
        h265_normal_line<S, 1>(v, tc, dep, deq, p.max_v);
        h265_normal_line<S, 2>(v, tc, dep, deq, p.max_v);
        h265_normal_line<S, 3>(v, tc, dep, deq, p.max_v);
    }
    if (entry & kH265KeepP) h265_sides<S, true>(v, kp, true);
    if (entry & kH265KeepQ) h265_sides<S, false>(v, kq, true);
}

/* chroma, 8.7.2.5.5 (tc from QpC) and 8.7.2.5.8 */
template <class S, int I>
DBK_HD void h265_chroma_line(int (&v)[8][8], int tc, int entry, int max_v)
{
    const int p0 = S::template P<I, 0>(v), p1 = S::template P<I, 1>(v);
    const int q0 = S::template Q<I, 0>(v), q1 = S::template Q<I, 1>(v);
    const int delta = clip1(((q0 - p0) * 4 + p1 - q1 + 4) >> 3, tc);
    if (!(entry & kH265KeepP)) S::template P<I, 0>(v) = clip2(p0 + delta, max_v);
    if (!(entry & kH265KeepQ)) S::template Q<I, 0>(v) = clip2(q0 - delta, max_v);
}
template <class S>
DBK_HD void h265_chroma_segment(int (&v)[8][8], int entry, int qpl, const H265Prm &p)
{
    if ((entry & kH265BsMask) != 2) return;
    const int qpc = h265_chroma_qp(qpl + p.c_qp_offset);
    const int tc = h265_tc(clampi(qpc + 2 + p.tc_off, 0, 53)) << p.shift;
    h265_chroma_line<S, 0>(v, tc, entry, p.max_v);
    h265_chroma_line<S, 1>(v, tc, entry, p.max_v);
    h265_chroma_line<S, 2>(v, tc, entry, p.max_v);
    h265_chroma_line<S, 3>(v, tc, entry, p.max_v);
}

/* the four segments of one offset block; entry[] / qpl[] in the order ver1, ver2, hor1, hor2 */
template <bool CHROMA>
DBK_HD void filter_block_h265(int (&v)[8][8], const int (&entry)[4], const int (&qpl)[4], const H265Prm &p)
{
    if constexpr (CHROMA) {
        h265_chroma_segment<SegVer1>(v, entry[0], qpl[0], p);
        h265_chroma_segment<SegVer2>(v, entry[1], qpl[1], p);
        h265_chroma_segment<SegHor1>(v, entry[2], qpl[2], p);
        h265_chroma_segment<SegHor2S>(v, entry[3], qpl[3], p);
    } else {
        h265_luma_segment<SegVer1>(v, entry[0], qpl[0], p);
        h265_luma_segment<SegVer2>(v, entry[1], qpl[1], p);
        h265_luma_segment<SegHor1>(v, entry[2], qpl[2], p);
        h265_luma_segment<SegHor2S>(v, entry[3], qpl[3], p);
    }
}

/*
 * bS entries of offset block (bx, by) from the 4-sample-granular arrays (vert: (W/8+1) x (H/4), hor: (H/8+1) x (W/4)).
 * Edges on the picture boundary (x = 0, x = W, y = 0, y = H) are never filtered (8.7.2: filterEdgeFlag = 0), and the
 * half of an edge that lies outside the picture does not exist.
 */
DBK_HD void load_block_bs_h265(const uint8_t *vert, const uint8_t *hor, int bx, int by, int nbx, int nby, int vstride,
                               int hstride, int (&entry)[4])
{
    const bool vedge = bx > 0 && bx < nbx - 1, hedge = by > 0 && by < nby - 1;
    entry[0] = (vedge && by > 0) ? vert[(2 * by - 1) * vstride + bx] : 0;
    entry[1] = (vedge && by < nby - 1) ? vert[(2 * by) * vstride + bx] : 0;
    entry[2] = (hedge && bx > 0) ? hor[by * hstride + 2 * bx - 1] : 0;
    entry[3] = (hedge && bx < nbx - 1) ? hor[by * hstride + 2 * bx] : 0;
}

/* qPL = (QpQ + QpP + 1) >> 1 of the four segments of the block whose (0,0) is plane sample (x0, y0): QpY of the coding
 * units holding q0,0 / p0,0 of line 0 (8.7.2.5.3); chroma positions are doubled (sc = 2).  map == NULL: scalar qp. */
DBK_HD void h265_block_qpl(const uint8_t *map, int map_stride, int unit_log2, int sc, int lw, int lh, int x0, int y0, int qp,
                           int (&qpl)[4])
{
    if (!map) {
        qpl[0] = qpl[1] = qpl[2] = qpl[3] = qp;
        return;
    }
    qpl[0] = seg_qp_from_map(map, map_stride, unit_log2, sc, lw, lh, x0 + 3, y0 + 0, x0 + 4, y0 + 0);
    qpl[1] = seg_qp_from_map(map, map_stride, unit_log2, sc, lw, lh, x0 + 3, y0 + 4, x0 + 4, y0 + 4);
    qpl[2] = seg_qp_from_map(map, map_stride, unit_log2, sc, lw, lh, x0 + 0, y0 + 3, x0 + 0, y0 + 4);
    qpl[3] = seg_qp_from_map(map, map_stride, unit_log2, sc, lw, lh, x0 + 4, y0 + 3, x0 + 4, y0 + 4);
}
/* the packed kernels' form: four map look-ups instead of eight (block_unit_qps, deblock_core.h); same values */
DBK_HD void h265_block_qpl4(const uint8_t *map, int map_stride, int unit_log2, int sc, int lw, int lh, int x0, int y0, int (&qpl)[4])
{
    int q[4];
    block_unit_qps(map, map_stride, unit_log2, sc, lw, lh, x0, y0, q);
    qpl[0] = seg_qp_avg(q[0], q[1]); /* ver1: above-left | above-right */
    qpl[1] = seg_qp_avg(q[2], q[3]); /* ver2: below-left | below-right */
    qpl[2] = seg_qp_avg(q[0], q[2]); /* hor1: above-left / below-left */
    qpl[3] = seg_qp_avg(q[1], q[3]); /* hor2: above-right / below-right */
}

/* ---- bS derivation, 8.7.2.4, on per-4x4-unit prediction data -------------------------------------------- */

constexpr unsigned kUIntra = 0x0001, kUCbf = 0x0002, kUTuLeft = 0x0004, kUTuTop = 0x0008, kUPuLeft = 0x0010,
                   kUPuTop = 0x0020, kUKeep = 0x0040, kUDbkOff = 0x0080, kUPredL0 = 0x0100, kUPredL1 = 0x0200,
                   kUNoxLeft = 0x0400, kUNoxTop = 0x0800;

struct H265Units {
    const uint16_t *flags;
    const int16_t *mv0, *mv1; /* [unit][2] */
    const int32_t *ref0, *ref1;
};

DBK_HD bool mv_far(int ax, int ay, int bx, int by) { return iabs(ax - bx) >= 4 || iabs(ay - by) >= 4; }

DBK_HD unsigned h265_bs_of_edge(const H265Units &u, long long ip, long long iq, bool left)
{
    const unsigned fp = u.flags[ip], fq = u.flags[iq];
    const unsigned tu = left ? kUTuLeft : kUTuTop, pu = left ? kUPuLeft : kUPuTop, nox = left ? kUNoxLeft : kUNoxTop;
    if (!(fq & (tu | pu)) || (fq & (kUDbkOff | nox))) return 0;
    unsigned bs;
    if ((fp | fq) & kUIntra) {
        bs = 2;
    } else if ((fq & tu) && ((fp | fq) & kUCbf)) {
        bs = 1;
    } else {
        /* motion of each side as (count, ref A, mv A, ref B, mv B) with list-0 first */
        const bool p0 = fp & kUPredL0, p1 = fp & kUPredL1, q0 = fq & kUPredL0, q1 = fq & kUPredL1;
        const int np = (int)p0 + (int)p1, nq = (int)q0 + (int)q1;
        if (np != nq) {
            bs = 1;
        } else if (np == 0) {
            bs = 0;
        } else {
            const int pra = p0 ? u.ref0[ip] : u.ref1[ip], pax = p0 ? u.mv0[2 * ip] : u.mv1[2 * ip], pay = p0 ? u.mv0[2 * ip + 1] : u.mv1[2 * ip + 1];
            const int qra = q0 ? u.ref0[iq] : u.ref1[iq], qax = q0 ? u.mv0[2 * iq] : u.mv1[2 * iq], qay = q0 ? u.mv0[2 * iq + 1] : u.mv1[2 * iq + 1];
            if (np == 1) {
                bs = pra != qra || mv_far(pax, pay, qax, qay);
            } else {
                const int prb = u.ref1[ip], pbx = u.mv1[2 * ip], pby = u.mv1[2 * ip + 1];
                const int qrb = u.ref1[iq], qbx = u.mv1[2 * iq], qby = u.mv1[2 * iq + 1];
                const bool same_set = (pra == qra && prb == qrb) || (pra == qrb && prb == qra);
                if (!same_set) bs = 1;
                else if (pra != prb)
                    bs = pra == qra ? (mv_far(pax, pay, qax, qay) || mv_far(pbx, pby, qbx, qby))
                                    : (mv_far(pax, pay, qbx, qby) || mv_far(pbx, pby, qax, qay));
                else
                    bs = (mv_far(pax, pay, qax, qay) || mv_far(pbx, pby, qbx, qby)) &&
                         (mv_far(pax, pay, qbx, qby) || mv_far(pbx, pby, qax, qay));
            }
        }
    }
    if (bs == 0) return 0;
    return bs | ((fp & kUKeep) ? (unsigned)kH265KeepP : 0u) | ((fq & kUKeep) ? (unsigned)kH265KeepQ : 0u);
}

} /* namespace dbk */
