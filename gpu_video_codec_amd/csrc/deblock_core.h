/*
 * deblock_core.h -- per-offset-block filter arithmetic of the HIP kernels.
 *
 * One offset block (SURVEY 8 layout: padded rows 8by..8by+7, cols 8bx..8bx+7 = image rows
 * 8by-4..8by+3, cols 8bx-4..8bx+3) is held entirely in registers by one lane; all four
 * segments (ver1 -> ver2 -> hor1 -> hor2, cpu.h:159-446) read and write only that block, so
 * no cross-lane traffic is needed for the arithmetic.
 *
 * The functions are DBK_HD so that tests/host_sim can run exactly this arithmetic on the CPU
 * against the oracle without a GPU (the product library only ever instantiates them in device
 * code; there is no CPU execution path in the library).
 *
 * Citations: cpu.h = /root/reference/hevc_deblocking_filter/hevc_deblocking_filter_cpu.h
 */
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DBK_HD __host__ __device__ __forceinline__
#else
#define DBK_HD inline
#endif

namespace dbk {

DBK_HD int iabs(int x) { return x < 0 ? -x : x; }
DBK_HD int clip1(int v, int c) { return v < -c ? -c : (v > c ? c : v); } /* cpu.h:1117-1120 */
DBK_HD int clip2(int v, int c) { return v < 0 ? 0 : (v > c ? c : v); }   /* cpu.h:1123-1126 */

/* what the four guards and bS bytes of one offset block evaluate to */
struct BlockBs {
    int ver1, ver2, hor1, hor2;
};

/*
 * Segment addressing inside the 8x8 register block v[row][col].
 * VERT  (vertical edge, cpu.h:159-284):   line i = row R0+i,  P_k = col PC-k, Q_k = col QC+k
 * !VERT (horizontal edge, cpu.h:287-446): line i = col PC+i / QC+i, P_k = row R0-k, Q_k = row R0+1+k
 */
template <bool VERT, int R0, int PC, int QC>
struct Seg {
    template <int I, int K> static DBK_HD int &P(int (&v)[8][8]) {
        if constexpr (VERT) return v[R0 + I][PC - K]; else return v[R0 - K][PC + I];
    }
    template <int I, int K> static DBK_HD int &Q(int (&v)[8][8]) {
        if constexpr (VERT) return v[R0 + I][QC + K]; else return v[R0 + 1 + K][QC + I];
    }
};

/* strong filter of one line (cpu.h:1152-1211); c = 2*tc */
template <class S, int I>
DBK_HD void luma_strong_line(int (&v)[8][8], int c, int max_v)
{
    const int p0 = S::template P<I, 0>(v), p1 = S::template P<I, 1>(v), p2 = S::template P<I, 2>(v), p3 = S::template P<I, 3>(v);
    const int q0 = S::template Q<I, 0>(v), q1 = S::template Q<I, 1>(v), q2 = S::template Q<I, 2>(v), q3 = S::template Q<I, 3>(v);
    S::template P<I, 0>(v) = clip2(p0 + clip1((p2 + 2 * p1 - 6 * p0 + 2 * q0 + q1 + 4) >> 3, c), max_v);
    S::template P<I, 1>(v) = clip2(p1 + clip1((p2 - 3 * p1 + p0 + q0 + 2) >> 2, c), max_v);
    S::template P<I, 2>(v) = clip2(p2 + clip1((2 * p3 - 5 * p2 + p1 + p0 + q0 + 4) >> 3, c), max_v);
    S::template Q<I, 0>(v) = clip2(q0 + clip1((q2 + 2 * q1 - 6 * q0 + 2 * p0 + p1 + 4) >> 3, c), max_v);
    S::template Q<I, 1>(v) = clip2(q1 + clip1((q2 - 3 * q1 + q0 + p0 + 2) >> 2, c), max_v);
    S::template Q<I, 2>(v) = clip2(q2 + clip1((2 * q3 - 5 * q2 + q1 + q0 + p0 + 4) >> 3, c), max_v);
}

/* normal filter of one line (cpu.h:1251-1354) */
template <class S, int I>
DBK_HD void luma_normal_line(int (&v)[8][8], int tc, bool cond5, bool cond6, int max_v)
{
    const int p0 = S::template P<I, 0>(v), p1 = S::template P<I, 1>(v), p2 = S::template P<I, 2>(v);
    const int q0 = S::template Q<I, 0>(v), q1 = S::template Q<I, 1>(v), q2 = S::template Q<I, 2>(v);
    const int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
    if (iabs(delta) < 10 * tc) {
        const int D = clip1(delta, 2 * tc);
        const int dp1 = clip1((((p2 + p0 + 1) >> 1) - p1 + D) >> 1, tc / 2);
        const int dq1 = clip1((((q2 + q0 + 1) >> 1) - q1 - D) >> 1, tc / 2);
        S::template P<I, 0>(v) = clip2(p0 + D, max_v);
        S::template Q<I, 0>(v) = clip2(q0 - D, max_v);
        if (cond5) S::template P<I, 1>(v) = clip2(p1 + dp1, max_v);
        if (cond6) S::template Q<I, 1>(v) = clip2(q1 + dq1, max_v);
    }
}

/* cpu.h:1359-1429 DeblockingFilterLuma on one 4-line segment */
template <class S>
DBK_HD void luma_segment(int (&v)[8][8], int beta, int tc, int max_v)
{
    const int dp0 = iabs(S::template P<0, 2>(v) - 2 * S::template P<0, 1>(v) + S::template P<0, 0>(v));
    const int dp3 = iabs(S::template P<3, 2>(v) - 2 * S::template P<3, 1>(v) + S::template P<3, 0>(v));
    const int dq0 = iabs(S::template Q<0, 2>(v) - 2 * S::template Q<0, 1>(v) + S::template Q<0, 0>(v));
    const int dq3 = iabs(S::template Q<3, 2>(v) - 2 * S::template Q<3, 1>(v) + S::template Q<3, 0>(v));
    if (!(dp0 + dp3 + dq0 + dq3 < beta)) return; /* cpu.h:1086-1087 */
    const int b8 = beta / 8, tc52 = 5 * tc / 2;
    const bool strong =
        (dp0 + dq0 < b8) && (dp3 + dq3 < b8) && /* cpu.h:1099-1100 */
        (iabs(S::template P<0, 3>(v) - S::template P<0, 0>(v)) + iabs(S::template Q<0, 0>(v) - S::template Q<0, 3>(v)) < b8) &&
        (iabs(S::template P<3, 3>(v) - S::template P<3, 0>(v)) + iabs(S::template Q<3, 0>(v) - S::template Q<3, 3>(v)) < b8) && /* 1104-1105 */
        (iabs(S::template P<0, 0>(v) - S::template Q<0, 0>(v)) < tc52) &&
        (iabs(S::template P<3, 0>(v) - S::template Q<3, 0>(v)) < tc52); /* 1109-1110 */
    if (strong) {
        luma_strong_line<S, 0>(v, 2 * tc, max_v);
        luma_strong_line<S, 1>(v, 2 * tc, max_v);
        luma_strong_line<S, 2>(v, 2 * tc, max_v);
        luma_strong_line<S, 3>(v, 2 * tc, max_v);
    } else {
        const bool cond5 = dp0 + dp3 < 3 * beta / 16; /* cpu.h:1245 */
        const bool cond6 = dq0 + dq3 < 3 * beta / 16; /* cpu.h:1249 */
        luma_normal_line<S, 0>(v, tc, cond5, cond6, max_v);
        luma_normal_line<S, 1>(v, tc, cond5, cond6, max_v);
        luma_normal_line<S, 2>(v, tc, cond5, cond6, max_v);
        luma_normal_line<S, 3>(v, tc, cond5, cond6, max_v);
    }
}

/* cpu.h:1431-1488 DeblockingFilterChroma, one line */
template <class S, int I>
DBK_HD void chroma_line(int (&v)[8][8], int tc, int max_v)
{
    const int p0 = S::template P<I, 0>(v), p1 = S::template P<I, 1>(v);
    const int q0 = S::template Q<I, 0>(v), q1 = S::template Q<I, 1>(v);
    S::template P<I, 0>(v) = clip2(p0 + clip1(((p0 - q0) * 4 + p1 - q1 + 4) >> 3, tc), max_v); /* cpu.h:1453,1475 */
    S::template Q<I, 0>(v) = clip2(q0 - clip1(((q0 - p0) * 4 + q1 - p1 + 4) >> 3, tc), max_v); /* cpu.h:1458,1476 */
}

template <class S>
DBK_HD void chroma_segment(int (&v)[8][8], int tc, int max_v)
{
    chroma_line<S, 0>(v, tc, max_v);
    chroma_line<S, 1>(v, tc, max_v);
    chroma_line<S, 2>(v, tc, max_v);
    chroma_line<S, 3>(v, tc, max_v);
}

using SegVer1 = Seg<true, 0, 3, 4>;  /* rows 0..3, edge between cols 3|4       (cpu.h:159-220) */
using SegVer2 = Seg<true, 4, 3, 4>;  /* rows 4..7                              (cpu.h:223-284) */
using SegHor1 = Seg<false, 3, 0, 0>; /* cols 0..3, edge between rows 3|4       (cpu.h:287-365) */
using SegHor2 = Seg<false, 3, 4, 0>; /* P cols 4..7, Q cols 0..3   (cpu.h:368-446) */

/* per-segment tc/beta (scalar QP: all four equal) */
struct BlockQp {
    int tc[4], beta[4];
};

/* the whole block, generic 32-bit arithmetic: ver1 -> ver2 -> hor1 -> hor2 */
template <bool CHROMA>
DBK_HD void filter_block_generic(int (&v)[8][8], const BlockBs &bs, const BlockQp &q, int max_v)
{
    if constexpr (CHROMA) {
        if (bs.ver1 == 2) chroma_segment<SegVer1>(v, q.tc[0], max_v); /* cpu.h:463 */
        if (bs.ver2 == 2) chroma_segment<SegVer2>(v, q.tc[1], max_v); /* cpu.h:519 */
        if (bs.hor1 == 2) chroma_segment<SegHor1>(v, q.tc[2], max_v); /* cpu.h:572 */
        if (bs.hor2 == 2) chroma_segment<SegHor2>(v, q.tc[3], max_v); /* cpu.h:649 */
    } else {
        if (bs.ver1 > 0) luma_segment<SegVer1>(v, q.beta[0], q.tc[0], max_v); /* cpu.h:164 */
        if (bs.ver2 > 0) luma_segment<SegVer2>(v, q.beta[1], q.tc[1], max_v); /* cpu.h:228 */
        if (bs.hor1 > 0) luma_segment<SegHor1>(v, q.beta[2], q.tc[2], max_v); /* cpu.h:292 */
        if (bs.hor2 > 0) luma_segment<SegHor2>(v, q.beta[3], q.tc[3], max_v); /* cpu.h:373 */
    }
}

/*
 * Guards and bS bytes of offset block (bx,by) (cpu.h:159-163, 223-227, 287-291, 368-372 and the
 * chroma twins 457-462, 514-518, 567-571, 644-648).  limit_bx / limit_by are the values the
 * ver2 / hor2 guards compare against: the plane's own nbx-1 / nby-1 for luma, the LUMA plane's
 * for chroma, with reads past the arrays treated as bS 0 (output-neutral, Q9(i)).
 */
DBK_HD BlockBs load_block_bs(const uint8_t *vert_bs, const uint8_t *hor_bs, int bx, int by,
                             int vstride, int hstride, int limit_bx, int limit_by, int n_vert, int n_hor)
{
    BlockBs b{0, 0, 0, 0};
    if (by > 0) b.ver1 = vert_bs[(by - 1) * vstride + bx];
    if (by < limit_by) {
        const int idx = by * vstride + bx;
        if (idx < n_vert) b.ver2 = vert_bs[idx];
    }
    if (bx > 0) b.hor1 = hor_bs[by * hstride + bx - 1];
    if (bx < limit_bx) {
        const int idx = by * hstride + bx;
        if (idx < n_hor) b.hor2 = hor_bs[idx];
    }
    return b;
}

DBK_HD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* QP-map extension: QP of a segment = (QpP + QpQ + 1) >> 1 (see include/hevc_deblock.h) */
DBK_HD int seg_qp_from_map(const uint8_t *map, int map_stride, int ctu_log2, int sc, int lw, int lh,
                           int xp, int yp, int xq, int yq)
{
    const int lxp = clampi(xp * sc, 0, lw - 1), lyp = clampi(yp * sc, 0, lh - 1);
    const int lxq = clampi(xq * sc, 0, lw - 1), lyq = clampi(yq * sc, 0, lh - 1);
    const int qpp = map[(lyp >> ctu_log2) * map_stride + (lxp >> ctu_log2)];
    const int qpq = map[(lyq >> ctu_log2) * map_stride + (lxq >> ctu_log2)];
    const int q = (qpp + qpq + 1) >> 1;
    return q > 51 ? 51 : q;
}

/*
 * The same QPs with half the look-ups, for the packed kernels: an offset block lies across the corner where four 8x8-aligned
 * blocks meet, a map unit is at least 8x8 luma samples, so the eight positions the four segments ask for fall into FOUR map
 * units -- above-left, above-right, below-left, below-right of the block's centre -- whatever the unit size and the plane
 * (chroma positions are doubled first; out-of-picture positions clamp into the picture exactly as above).
 * q = {TL, TR, BL, BR}.  Requires unit_log2 >= 3 (checked by the entry points).
 */
DBK_HD void block_unit_qps(const uint8_t *map, int map_stride, int unit_log2, int sc, int lw, int lh, int x0, int y0, int (&q)[4])
{
    const int xl = clampi((x0 + 3) * sc, 0, lw - 1) >> unit_log2, xr = clampi((x0 + 4) * sc, 0, lw - 1) >> unit_log2;
    const int yt = clampi((y0 + 3) * sc, 0, lh - 1) >> unit_log2, yb = clampi((y0 + 4) * sc, 0, lh - 1) >> unit_log2;
    q[0] = map[yt * map_stride + xl];
    q[1] = map[yt * map_stride + xr];
    q[2] = map[yb * map_stride + xl];
    q[3] = map[yb * map_stride + xr];
}
DBK_HD int seg_qp_avg(int qp_p, int qp_q)
{
    const int q = (qp_p + qp_q + 1) >> 1;
    return q > 51 ? 51 : q;
}

} /* namespace dbk */
