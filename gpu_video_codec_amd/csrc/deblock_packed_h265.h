/*
 * deblock_packed_h265.h -- packed-int16 block arithmetic of the SPEC-EXACT mode (H.265 clause 8.7.2): the same
 * register layout, perms and carry-free tricks as deblock_packed.h, with the standard's thresholds, clips, hor2 geometry,
 * chroma formula and the keep-P / keep-Q flags.  See deblock_h265.h for the 32-bit form and the citations.
 *
 * Keep flags without keeping a copy of the block: a side that must stay unmodified gets a zero clip range
 * (strong filter: Clip3(p - 0, p + 0, x) = p) or a zero mask on its delta (normal / chroma), so no original has to stay
 * live across the filter.
 */
#pragma once
#include "deblock_h265.h"
#include "deblock_packed.h"

namespace dbk {

/* 8.7.2.5.3 + 8.7.2.5.6 on pair A (lines 0 and 3) */
DBK_HD Decision decide_h265(const Taps &a, int beta, int tc)
{
    const pk dp = absdiff(uadd(a.p2, a.p0), uadd(a.p1, a.p1));
    const pk dq = absdiff(uadd(a.q2, a.q0), uadd(a.q1, a.q1));
    const pk dpq = uadd(dp, dq);
    Decision d;
    d.filter = lohi_sum(dpq) < (unsigned)beta; /* d < beta */
    const pk e = uadd(absdiff(a.p3, a.p0), absdiff(a.q0, a.q3));
    const pk f = absdiff(a.p0, a.q0);
    /* 2*dpq < (beta >> 2)  <=>  dpq < ceil((beta >> 2) / 2) */
    const int t_dpq = ((beta >> 2) + 1) >> 1, t_e = beta >> 3, t_pq = (5 * tc + 1) >> 1;
    const uint32_t k1 = 0x80008000u | ((uint32_t)(t_dpq - 1) * 0x00010001u);
    const uint32_t k2 = 0x80008000u | ((uint32_t)(t_e - 1) * 0x00010001u);
    const uint32_t k3 = 0x80008000u | ((uint32_t)(t_pq - 1) * 0x00010001u);
    const uint32_t ok = (k1 - pk_bits(dpq)) & (k2 - pk_bits(e)) & (k3 - pk_bits(f)) & 0x80008000u;
    d.strong = t_dpq > 0 && t_e > 0 && t_pq > 0 && ok == 0x80008000u;
    const unsigned side = (unsigned)((beta + (beta >> 1)) >> 3);
    d.cond5 = lohi_sum(dp) < side; /* dEp */
    d.cond6 = lohi_sum(dq) < side; /* dEq */
    return d;
}

/* strong filter with separate clip ranges for the two sides (0 = side kept) */
DBK_HD void strong_pair_h265(Taps &t, pk cp, pk cq)
{
    const pk u2 = uaddc(uadd(t.p0, t.q0), 0x00020002u);
    const pk tp = uadd(u2, t.p1), tq = uadd(u2, t.q1);
    const pk bp = uadd(tp, t.p2), bq = uadd(tq, t.q2);
    const pk p32 = uadd(t.p3, t.p2), q32 = uadd(t.q3, t.q2);
    const pk s0p = uadd(uadd(tp, bp), t.q1) >> 3;
    const pk s1p = bp >> 2;
    const pk s2p = uaddc(uadd(uadd(p32, p32), bp), 0x00020002u) >> 3;
    const pk s0q = uadd(uadd(tq, bq), t.p1) >> 3;
    const pk s1q = bq >> 2;
    const pk s2q = uaddc(uadd(uadd(q32, q32), bq), 0x00020002u) >> 3;
    const pk np0 = pk_clamp(s0p, t.p0 - cp, uadd(t.p0, cp));
    const pk np1 = pk_clamp(s1p, t.p1 - cp, uadd(t.p1, cp));
    const pk np2 = pk_clamp(s2p, t.p2 - cp, uadd(t.p2, cp));
    const pk nq0 = pk_clamp(s0q, t.q0 - cq, uadd(t.q0, cq));
    const pk nq1 = pk_clamp(s1q, t.q1 - cq, uadd(t.q1, cq));
    const pk nq2 = pk_clamp(s2q, t.q2 - cq, uadd(t.q2, cq));
    t.p0 = np0; t.p1 = np1; t.p2 = np2;
    t.q0 = nq0; t.q1 = nq1; t.q2 = nq2;
}

/* normal filter, delta clipped to +-tc; mp0 / mq0 gate p0 / q0, mp1 / mq1 gate p1 / q1 (dEp, dEq and the keep flags) */
template <bool WIDE = false>
DBK_HD void normal_pair_h265(Taps &t, int tc, pk mp0, pk mq0, pk mp1, pk mq1)
{
    const pk c = splat(tc), c2 = splat(tc >> 1), lim = splat(10 * tc);
    const pk zero = splat(0);
    pk delta;
    if constexpr (WIDE) { /* 12 bit: see normal_pair_unclipped<true> in deblock_packed.h */
        const pk a = t.q0 - t.p0;
        delta = (a + (mad_k<-3>(t.q1 - t.p1, a + splat(8)) >> 3)) >> 1;
    } else {
        delta = mad_k<9>(t.q0 - t.p0, mad_kc<-3, 8>(t.q1 - t.p1)) >> 4;
    }
    const pk on = (pk_abs(delta) - lim) >> 15;
    const pk D = pk_clamp(delta, zero - c, c);
    const pk xp = uaddc(uadd(t.p2, t.p0), 0x00010001u);
    const pk xq = uaddc(uadd(t.q2, t.q0), 0x00010001u);
    const pk dp1 = pk_clamp(mad_k<2>(D, mad_k<-2>(t.p1, xp)) >> 2, zero - c2, c2);
    const pk dq1 = pk_clamp(mad_k<-2>(D, mad_k<-2>(t.q1, xq)) >> 2, zero - c2, c2);
    const pk Dm = D & on;
    t.p0 = t.p0 + (Dm & mp0);
    t.q0 = t.q0 - (Dm & mq0);
    t.p1 = t.p1 + (dp1 & on & mp1);
    t.q1 = t.q1 + (dq1 & on & mq1);
}

/* the per-block scalars of the one-QP spec-exact kernels: beta is one scalar, tc one of two (bS 1 / bS 2) */
struct H265Uni {
    int beta, tc1, tc2;
};
DBK_HD H265Uni h265_uni(int beta, int tc_bs1, int tc_bs2)
{
    H265Uni u;
    u.beta = beta; u.tc1 = tc_bs1; u.tc2 = tc_bs2;
    return u;
}

/*
 * One luma segment, one-QP kernels (round 3).  In nearly every wave the lanes that filter share ONE bS value and nobody
 * carries a keep flag (PCM / transquant bypass are rare, and a picture's edges are mostly intra = 2 or mostly inter = 1):
 * such a wave runs the SAME code as the reference-exact kernel -- decisions through saturating subtractions and biased
 * adds, the three-instruction strong clamp, one conservative |delta| test, operands in SGPRs -- with the standard's
 * thresholds and clip (LumaKLazy<true>).  A wave with both bS values runs that core with per-lane operands (LumaKEager).
 * Lanes with a keep flag take the general per-lane form further down.
 *
 * The forms are STAGES, one after the other, each behind a wave-uniform guard and each switched per lane (`enable`), all
 * updating the taps in place -- not the arms of one if / else: as alternatives their results met in a twelve-register PHI
 * that the compiler resolved with twelve copies at the end of the hot arm (and twelve more around a lane-level `if`): 96
 * v_mov_b32 per block, the whole difference between this kernel's 786 VALU instructions per wave and the reference-exact
 * kernel's 704.
 */
template <bool WIDE = false>
DBK_HD void luma_pairs_h265(Taps &a, Taps &b, int entry, int beta, int tc, int max_v);

#if !DBK_DEV
inline int &h265_sim_force_mixed_flag() { static int f = 0; return f; }
inline bool h265_sim_force_mixed() { return h265_sim_force_mixed_flag() != 0; }
#endif

/* a wave-uniform flag the optimiser cannot relate to the expression it came from (two guards of opposite sense stay two
 * `if`s instead of being fused into one if / else) */
DBK_HD bool opaque_uniform(bool v)
{
#if DBK_DEV
    int x = v ? 1 : 0;
    asm volatile("" : "+v"(x));
    return __builtin_amdgcn_readfirstlane(x) != 0;
#else
    return v;
#endif
}

/* keep_any (wave-uniform, once per block: h265_keep_any): some lane of the wave carries a keep flag in one of its four
 * segments.  PCM / transquant-bypass units are rare, so nearly every wave skips the keep-flag tests of all four segments on one
 * scalar branch each instead of evaluating them per lane. */
template <bool WIDE = false>
DBK_HD void luma_pairs_h265_uni(Taps &a, Taps &b, int entry, int beta, int tc, int max_v, const H265Uni &u, bool keep_any)
{
    const int bs = entry & kH265BsMask;
    bool fast = bs != 0;
    if (keep_any) {
        const bool keep = (entry & (kH265KeepP | kH265KeepQ)) != 0;
        if (any_lane(bs != 0 && keep)) luma_pairs_h265<WIDE>(a, b, keep ? entry : 0, beta, tc, max_v); /* entry 0: lane off */
        fast = fast && !keep;
    }
    const unsigned long long m1 = lane_ballot(fast && bs == 1), m2 = lane_ballot(fast && bs == 2);
#if DBK_DEV
    const bool one_bs = m1 == 0ull || m2 == 0ull;
#else
    /* CPU build (tests/host_sim): a "wave" is one block and never holds both bS values; the test switch sends every block down
     * the mixed-wave form instead, so that its arithmetic is checked against the oracle without a GPU */
    const bool one_bs = (m1 == 0ull || m2 == 0ull) && !h265_sim_force_mixed();
#endif
    if ((m1 | m2) != 0ull && one_bs) {
        /* the wave's one tc is picked on the scalar unit and the segment constants are built from it right here */
        luma_pairs<WIDE, true>(a, b, LumaKLazy<true>{u.beta, m2 != 0ull ? u.tc2 : u.tc1}, max_v, 0, fast);
    }
    if (opaque_uniform(!one_bs)) {
        /* bS 1 and bS 2 side by side (inter pictures): the same core with per-lane operands -- each tc-dependent operand one of
         * two scalars, picked per lane (LumaKSel) */
        luma_pairs<WIDE, false>(a, b, LumaKSel<true>{LumaKLazy<true>{u.beta, u.tc1}, LumaKLazy<true>{u.beta, u.tc2}, bs == 2}, max_v, 0, fast);
    }
}

/*
 * One luma segment, QP-map kernels: beta and tc differ from lane to lane anyway (tc already carries the lane's bS), so a
 * mix of bS values costs nothing: the shared core with per-lane operands (LumaKEager, the standard's thresholds) for every
 * lane without a keep flag, the general form for the others -- two stages, as above.
 */
template <bool WIDE = false>
DBK_HD void luma_pairs_h265_map(Taps &a, Taps &b, int entry, int beta, int tc, int max_v, bool keep_any)
{
    const int bs = entry & kH265BsMask;
    bool fast = bs != 0;
    if (keep_any) {
        const bool keep = (entry & (kH265KeepP | kH265KeepQ)) != 0;
        if (any_lane(bs != 0 && keep)) luma_pairs_h265<WIDE>(a, b, keep ? entry : 0, beta, tc, max_v);
        fast = fast && !keep;
    }
    if (any_lane(fast)) luma_pairs<WIDE, false>(a, b, LumaKEager::make<true>(beta, tc), max_v, 0, fast);
}

/* The same with the operands out of the workgroup's table (deblock_packed.h, ktab_build<true>: rows by the beta index
 * qPL + beta_offset and the tc index qPL + 2 (bS - 1) + tc_offset, round 4): no beta / tc values are looked up at all unless
 * a lane of the wave carries a keep flag. */
template <bool WIDE = false>
DBK_HD void luma_pairs_h265_tab(Taps &a, Taps &b, int entry, const DBK_LDS uint32_t *tab, int ib, int it, int shift, int max_v, bool keep_any)
{
    const int bs = entry & kH265BsMask;
    bool fast = bs != 0;
    if (keep_any) {
        const bool keep = (entry & (kH265KeepP | kH265KeepQ)) != 0;
        if (any_lane(bs != 0 && keep)) luma_pairs_h265<WIDE>(a, b, keep ? entry : 0, h265_beta(ib) << shift, h265_tc(it) << shift, max_v);
        fast = fast && !keep;
    }
    if (any_lane(fast)) luma_pairs<WIDE, false>(a, b, LumaKLds::rows(tab, ib, it), max_v, 0, fast);
}

/* one luma segment from its two pairs; entry = bS byte with the keep flags; beta / tc already looked up */
template <bool WIDE>
DBK_HD void luma_pairs_h265(Taps &a, Taps &b, int entry, int beta, int tc, int max_v)
{
    if ((entry & kH265BsMask) == 0) return;
    const Decision d = decide_h265(a, beta, tc);
    if (!d.filter) return;
    const bool kp = entry & kH265KeepP, kq = entry & kH265KeepQ;
    if (d.strong) {
        const pk cp = splat(kp ? 0 : 2 * tc), cq = splat(kq ? 0 : 2 * tc);
        strong_pair_h265(a, cp, cq);
        strong_pair_h265(b, cp, cq);
    } else {
        const pk mp0 = splat(kp ? 0 : -1), mq0 = splat(kq ? 0 : -1);
        const pk mp1 = mask_of(d.cond5 && !kp), mq1 = mask_of(d.cond6 && !kq);
        normal_pair_h265<WIDE>(a, tc, mp0, mq0, mp1, mq1);
        normal_pair_h265<WIDE>(b, tc, mp0, mq0, mp1, mq1);
        const uint32_t over = (pk_bits(a.p0) | pk_bits(a.q0) | pk_bits(a.p1) | pk_bits(a.q1) |
                               pk_bits(b.p0) | pk_bits(b.q0) | pk_bits(b.p1) | pk_bits(b.q1)) &
                              (0x00010001u * (0xffffu & ~(uint32_t)max_v));
        if (over) { /* Clip1Y */
            const pk zero = splat(0), maxv = splat(max_v);
            a.p0 = pk_clamp(a.p0, zero, maxv); a.q0 = pk_clamp(a.q0, zero, maxv);
            a.p1 = pk_clamp(a.p1, zero, maxv); a.q1 = pk_clamp(a.q1, zero, maxv);
            b.p0 = pk_clamp(b.p0, zero, maxv); b.q0 = pk_clamp(b.q0, zero, maxv);
            b.p1 = pk_clamp(b.p1, zero, maxv); b.q1 = pk_clamp(b.q1, zero, maxv);
        }
    }
}

/* per-segment operands of a block */
struct H265Seg {
    int entry[4]; /* bS bytes: ver1, ver2, hor1, hor2 */
    int tc[4], beta[4];
    /* TAB form (QP-map luma launches): rows of the workgroup's operand table instead of the values above */
    const DBK_LDS uint32_t *tab;
    int ib[4], it[4], shift;
};
/* a segment of a one-QP kernel (u != NULL) or of a QP-map kernel (TAB: operands from the workgroup's table) */
/* one test for the block's four segments (see luma_pairs_h265_uni) */
DBK_HD bool h265_keep_any(const H265Seg &s)
{
    return any_lane(((s.entry[0] | s.entry[1] | s.entry[2] | s.entry[3]) & (int)(kH265KeepP | kH265KeepQ)) != 0);
}
template <bool WIDE = false, bool TAB = false>
DBK_HD void luma_seg_h265(Taps &a, Taps &b, const H265Seg &s, int i, int max_v, const H265Uni *u, bool keep_any)
{
    if constexpr (TAB) {
        luma_pairs_h265_tab<WIDE>(a, b, s.entry[i], s.tab, s.ib[i], s.it[i], s.shift, max_v, keep_any);
    } else {
        if (u) luma_pairs_h265_uni<WIDE>(a, b, s.entry[i], s.beta[i], s.tc[i], max_v, *u, keep_any);
        else luma_pairs_h265_map<WIDE>(a, b, s.entry[i], s.beta[i], s.tc[i], max_v, keep_any);
    }
}

/* 8-bit luma block: ver1 -> ver2 -> hor1 -> hor2 with the conformant hor2 (P and Q both in columns 4..7) */
template <bool TAB = false>
DBK_HD void packed_filter_luma_block_h265(uint32_t (&L)[8], uint32_t (&R)[8], const H265Seg &s, const H265Uni *u = nullptr)
{
    Taps va1 = unpack_ver(L[0], L[3], R[0], R[3]), vb1 = unpack_ver(L[1], L[2], R[1], R[2]);
    Taps va2 = unpack_ver(L[4], L[7], R[4], R[7]), vb2 = unpack_ver(L[5], L[6], R[5], R[6]);
    const bool keep_any = h265_keep_any(s);
    luma_seg_h265<false, TAB>(va1, vb1, s, 0, 255, u, keep_any);
    luma_seg_h265<false, TAB>(va2, vb2, s, 1, 255, u, keep_any);

    Taps ha, hb, ga, gb;
    /* hor1: lines = cols 0..3 (ver taps p3..p0), P_k = row 3-k, Q_k = row 4+k */
    ha.p0 = pick_hi(va1.p3, va1.p0); hb.p0 = pick_hi(va1.p2, va1.p1);
    ha.p1 = pick_hi(vb1.p3, vb1.p0); hb.p1 = pick_hi(vb1.p2, vb1.p1);
    ha.p2 = pick_lo(vb1.p3, vb1.p0); hb.p2 = pick_lo(vb1.p2, vb1.p1);
    ha.p3 = pick_lo(va1.p3, va1.p0); hb.p3 = pick_lo(va1.p2, va1.p1);
    ha.q0 = pick_lo(va2.p3, va2.p0); hb.q0 = pick_lo(va2.p2, va2.p1);
    ha.q1 = pick_lo(vb2.p3, vb2.p0); hb.q1 = pick_lo(vb2.p2, vb2.p1);
    ha.q2 = pick_hi(vb2.p3, vb2.p0); hb.q2 = pick_hi(vb2.p2, vb2.p1);
    ha.q3 = pick_hi(va2.p3, va2.p0); hb.q3 = pick_hi(va2.p2, va2.p1);
    luma_seg_h265<false, TAB>(ha, hb, s, 2, 255, u, keep_any);
    /* hor2: lines = cols 4..7 (ver taps q0..q3), same rows */
    ga.p0 = pick_hi(va1.q0, va1.q3); gb.p0 = pick_hi(va1.q1, va1.q2);
    ga.p1 = pick_hi(vb1.q0, vb1.q3); gb.p1 = pick_hi(vb1.q1, vb1.q2);
    ga.p2 = pick_lo(vb1.q0, vb1.q3); gb.p2 = pick_lo(vb1.q1, vb1.q2);
    ga.p3 = pick_lo(va1.q0, va1.q3); gb.p3 = pick_lo(va1.q1, va1.q2);
    ga.q0 = pick_lo(va2.q0, va2.q3); gb.q0 = pick_lo(va2.q1, va2.q2);
    ga.q1 = pick_lo(vb2.q0, vb2.q3); gb.q1 = pick_lo(vb2.q1, vb2.q2);
    ga.q2 = pick_hi(vb2.q0, vb2.q3); gb.q2 = pick_hi(vb2.q1, vb2.q2);
    ga.q3 = pick_hi(va2.q0, va2.q3); gb.q3 = pick_hi(va2.q1, va2.q2);
    luma_seg_h265<false, TAB>(ga, gb, s, 3, 255, u, keep_any);

    L[0] = row_of(ha.p3, hb.p3); L[1] = row_of(ha.p2, hb.p2); L[2] = row_of(ha.p1, hb.p1); L[3] = row_of(ha.p0, hb.p0);
    L[4] = row_of(ha.q0, hb.q0); L[5] = row_of(ha.q1, hb.q1); L[6] = row_of(ha.q2, hb.q2); L[7] = row_of(ha.q3, hb.q3);
    R[0] = row_of(ga.p3, gb.p3); R[1] = row_of(ga.p2, gb.p2); R[2] = row_of(ga.p1, gb.p1); R[3] = row_of(ga.p0, gb.p0);
    R[4] = row_of(ga.q0, gb.q0); R[5] = row_of(ga.q1, gb.q1); R[6] = row_of(ga.q2, gb.q2); R[7] = row_of(ga.q3, gb.q3);
}

/* ---- chroma, 8.7.2.5.8: delta = Clip3(-tc, tc, (((q0 - p0) << 2) + p1 - q1 + 4) >> 3), p0 += delta, q0 -= delta ---- */
DBK_HD void chroma_pair_h265(pk &p0, pk p1, pk &q0, pk q1, pk tc, pk mp, pk mq, int max_v = 255)
{
    const pk zero = splat(0), maxv = splat(max_v);
    const pk d = pk_clamp((((q0 - p0) << 2) + p1 - q1 + splat(4)) >> 3, zero - tc, tc);
    const pk np0 = pk_clamp(p0 + (d & mp), zero, maxv);
    const pk nq0 = pk_clamp(q0 - (d & mq), zero, maxv);
    p0 = np0;
    q0 = nq0;
}

template <int R0>
DBK_HD void chroma_ver_h265(uint32_t (&L)[8], uint32_t (&R)[8], int tc, int entry)
{
    if ((entry & kH265BsMask) != 2) return;
    const pk c = splat(tc), mp = splat((entry & kH265KeepP) ? 0 : -1), mq = splat((entry & kH265KeepQ) ? 0 : -1);
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int ra = R0 + (h ? 1 : 0), rb = R0 + (h ? 2 : 3);
        pk p0 = bits_pk(perm(L[rb], L[ra], 0x0c070c03u)), p1 = bits_pk(perm(L[rb], L[ra], 0x0c060c02u));
        pk q0 = bits_pk(perm(R[rb], R[ra], 0x0c040c00u)), q1 = bits_pk(perm(R[rb], R[ra], 0x0c050c01u));
        chroma_pair_h265(p0, p1, q0, q1, c, mp, mq);
        L[ra] = perm(pk_bits(p0), L[ra], 0x04020100u);
        L[rb] = perm(pk_bits(p0), L[rb], 0x06020100u);
        R[ra] = perm(pk_bits(q0), R[ra], 0x03020104u);
        R[rb] = perm(pk_bits(q0), R[rb], 0x03020106u);
    }
}

/* X = the 8 row dwords of the column half (L: cols 0..3, R: cols 4..7); P rows 2,3 and Q rows 4,5 of the same half */
DBK_HD void chroma_hor_h265(uint32_t (&X)[8], int tc, int entry)
{
    if ((entry & kH265BsMask) != 2) return;
    const pk c = splat(tc), mp = splat((entry & kH265KeepP) ? 0 : -1), mq = splat((entry & kH265KeepQ) ? 0 : -1);
    pk ap0 = unpack_hor_a(X[3]), ap1 = unpack_hor_a(X[2]), aq0 = unpack_hor_a(X[4]), aq1 = unpack_hor_a(X[5]);
    pk bp0 = unpack_hor_b(X[3]), bp1 = unpack_hor_b(X[2]), bq0 = unpack_hor_b(X[4]), bq1 = unpack_hor_b(X[5]);
    chroma_pair_h265(ap0, ap1, aq0, aq1, c, mp, mq);
    chroma_pair_h265(bp0, bp1, bq0, bq1, c, mp, mq);
    X[3] = perm(pk_bits(bp0), pk_bits(ap0), 0x02060400u);
    X[4] = perm(pk_bits(bq0), pk_bits(aq0), 0x02060400u);
}

template <bool CHROMA, bool TAB = false>
DBK_HD void packed_filter_block_h265(uint32_t (&L)[8], uint32_t (&R)[8], const H265Seg &s, const H265Uni *u = nullptr)
{
    if constexpr (CHROMA) {
        chroma_ver_h265<0>(L, R, s.tc[0], s.entry[0]);
        chroma_ver_h265<4>(L, R, s.tc[1], s.entry[1]);
        chroma_hor_h265(L, s.tc[2], s.entry[2]);
        chroma_hor_h265(R, s.tc[3], s.entry[3]);
    } else {
        packed_filter_luma_block_h265<TAB>(L, R, s, u);
    }
}

/* per-segment tc / beta from qPL, the bS and the offsets (8.7.2.5.3 luma, 8.7.2.5.5 chroma); 8-bit samples */
template <bool CHROMA>
DBK_HD void h265_seg_params(const int (&entry)[4], const int (&qpl)[4], const H265Prm &p, H265Seg &s)
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int bs = entry[i] & kH265BsMask;
        s.entry[i] = entry[i];
        if constexpr (CHROMA) {
            s.beta[i] = 0;
            s.tc[i] = h265_tc(clampi(h265_chroma_qp(qpl[i] + p.c_qp_offset) + 2 + p.tc_off, 0, 53)) << p.shift;
        } else {
            s.beta[i] = h265_beta(clampi(qpl[i] + p.beta_off, 0, 51)) << p.shift;
            s.tc[i] = h265_tc(clampi(qpl[i] + 2 * (bs - 1) + p.tc_off, 0, 53)) << p.shift;
        }
    }
}

/* TAB form, luma: the table rows of the four segments (nothing is looked up here) */
DBK_HD void h265_seg_rows(const int (&entry)[4], const int (&qpl)[4], const H265Prm &p, const DBK_LDS uint32_t *tab, H265Seg &s)
{
    s.tab = tab;
    s.shift = p.shift;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int bs = entry[i] & kH265BsMask;
        s.entry[i] = entry[i];
        s.ib[i] = clampi(qpl[i] + p.beta_off, 0, 51);
        s.it[i] = clampi(qpl[i] + 2 * (bs - 1) + p.tc_off, 0, 53);
        s.tc[i] = 0;
        s.beta[i] = 0;
    }
}

} /* namespace dbk */
