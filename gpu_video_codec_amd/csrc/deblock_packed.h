/*
 * deblock_packed.h -- packed-int16 per-block arithmetic of the fast HIP kernel (8-bit samples).
 *
 * Register layout of one offset block in one lane: L[r] = cols 0..3 of row r, R[r] = cols 4..7
 * (little-endian bytes), 16 dwords.  A segment's 4 lines are processed as two PAIRS packed in the
 * two 16-bit halves of a VGPR so that every arithmetic instruction is a v_pk_*_i16:
 *     pair A = lines (0,3)  -- the two lines every decision uses, so all decision
 *                              arithmetic runs once, on pair A only
 *     pair B = lines (1,2)
 * Bytes <-> int16 pairs move through v_perm_b32 only (no shifts/masks); the pairs stay int16 from the
 * vertical to the horizontal edges (32 unpack + 24 re-pair + 20 pack perms per block).  Every
 * intermediate fits int16 (|9*255 + 3*255 + 8| = 3068; 16-bit containers up to 11 bit: 12*2047+8).
 *
 * Algebra used (bit-exact with the reference's delta form, cpu.h:1154-1211):
 *   (p2 + 2p1 - 6p0 + 2q0 + q1 + 4) >> 3  ==  ((p2 + 2p1 + 2p0 + 2q0 + q1 + 4) >> 3) - p0
 * because 8*p0 is a multiple of 8 (arithmetic shift = floor).  So p0' = clamp(S0>>3, p0-c, p0+c),
 * and since S0>>3 is a weighted mean of samples it already lies in [0,255]: the strong filter needs
 * no final Clip2.  Same for the 4-tap (>>2) and 5-tap (>>3) sums.
 *
 * DBK_HOST_SIM: tests/host_sim compiles this header for the CPU with emulated primitives so the
 * exact arithmetic below is checked against the oracle without a GPU.
 */
#pragma once
#include <stdint.h>

#include "deblock_core.h"

namespace dbk {

typedef short pk __attribute__((vector_size(4))); /* two int16 lanes in one VGPR */

#if defined(__HIP_DEVICE_COMPILE__)
#define DBK_DEV 1
#define DBK_LDS __attribute__((address_space(3))) /* a pointer into the workgroup's LDS (ds_read, not flat) */
#else
#define DBK_DEV 0
#define DBK_LDS
#endif

DBK_HD uint32_t pk_bits(pk a) { return __builtin_bit_cast(uint32_t, a); }
DBK_HD pk bits_pk(uint32_t a) { return __builtin_bit_cast(pk, a); }
DBK_HD pk splat(int v) { return pk{(short)v, (short)v}; }
/* the same value in both halves, built with 32-bit integer operations: for a wave-uniform v (scalar QP) the compiler keeps
 * all of it on the scalar unit and hands the packed constant to the VALU instruction as an SGPR operand */
DBK_HD pk splat_u(int v) { const uint32_t x = (uint32_t)v & 0xffffu; return bits_pk(x | (x << 16)); }
/* all-ones / all-zeros in both halves from a lane condition: one v_cndmask, no re-packing */
DBK_HD pk mask_of(bool c) { return bits_pk(c ? 0xffffffffu : 0u); }
/* 1 / 0 in both halves from a lane condition: the factor of a packed multiply-add that applies or drops a delta.  (Round 3
 * tried 1 / 0 in the low half only -- a v_cndmask between two inline constants, no v_mov of 0x00010001 -- with the low half
 * feeding both lanes through op_sel_hi; that needs the multiply-add as inline asm, and the register copies the compiler then
 * places at the branch joins cost 7 instructions per segment for the 1 saved.  The VOP3 select with the constant in an SGPR
 * is not encodable: mask + constant are two scalar operands, gfx9 allows one.) */
DBK_HD pk one_of(bool c) { return bits_pk(c ? 0x00010001u : 0u); }
/* a * b + c on three registers: v_pk_mad_u16 */
DBK_HD pk mad_vvv(pk a, pk b, pk c) { return a * b + c; }

DBK_HD pk pk_max(pk a, pk b)
{
#if DBK_DEV
    return __builtin_elementwise_max(a, b); /* v_pk_max_i16 */
#else
    return pk{a[0] > b[0] ? a[0] : b[0], a[1] > b[1] ? a[1] : b[1]};
#endif
}
DBK_HD pk pk_min(pk a, pk b)
{
#if DBK_DEV
    return __builtin_elementwise_min(a, b); /* v_pk_min_i16 */
#else
    return pk{a[0] < b[0] ? a[0] : b[0], a[1] < b[1] ? a[1] : b[1]};
#endif
}
DBK_HD pk pk_abs(pk a) { return pk_max(a, splat(0) - a); }
/* max of fields read as UNSIGNED 16-bit numbers (v_pk_max_u16): for fields that carry a 0x8000 bias */
DBK_HD pk pk_maxu(pk a, pk b)
{
#if DBK_DEV
    typedef unsigned short upk __attribute__((vector_size(4)));
    return __builtin_bit_cast(pk, __builtin_elementwise_max(__builtin_bit_cast(upk, a), __builtin_bit_cast(upk, b)));
#else
    const unsigned short a0 = (unsigned short)a[0], a1 = (unsigned short)a[1], b0 = (unsigned short)b[0], b1 = (unsigned short)b[1];
    return pk{(short)(a0 > b0 ? a0 : b0), (short)(a1 > b1 ? a1 : b1)};
#endif
}
/* max(a - b, 0) of non-negative fields: ONE v_pk_sub_u16 with the clamp bit (unsigned saturation) */
DBK_HD pk sub_sat(pk a, pk b)
{
#if DBK_DEV
    typedef unsigned short upk __attribute__((vector_size(4)));
    return __builtin_bit_cast(pk, __builtin_elementwise_sub_sat(__builtin_bit_cast(upk, a), __builtin_bit_cast(upk, b)));
#else
    const unsigned short a0 = (unsigned short)a[0], a1 = (unsigned short)a[1], b0 = (unsigned short)b[0], b1 = (unsigned short)b[1];
    return pk{(short)(a0 > b0 ? a0 - b0 : 0), (short)(a1 > b1 ? a1 - b1 : 0)};
#endif
}

/*
 * Carry-free SWAR forms.  On gfx950 the packed v_pk_add/sub_u16 issue at 4 cycles per wave64 while a
 * plain v_add_u32 / v_sub_u32 issues at 2 (measured, tools/ubench/valu_rate*.hip).  When both 16-bit
 * fields are known non-negative and the field results stay inside [0, 65535] no carry or borrow can
 * cross the field boundary, so a 32-bit add / sub of the two packed registers IS the packed result.
 */
DBK_HD pk uadd(pk a, pk b) { return bits_pk(pk_bits(a) + pk_bits(b)); }  /* fields >= 0, sums < 65536 */
DBK_HD pk usub(pk a, pk b) { return bits_pk(pk_bits(a) - pk_bits(b)); }  /* field-wise a >= b >= 0 */
DBK_HD pk uaddc(pk a, uint32_t c2) { return bits_pk(pk_bits(a) + c2); }  /* c2 = constant in both fields */
DBK_HD pk uadd3c(pk a, pk b, uint32_t c2) { return bits_pk(pk_bits(a) + pk_bits(b) + c2); } /* one v_add3_u32 */
DBK_HD pk uadd3(pk a, pk b, pk c) { return bits_pk(pk_bits(a) + pk_bits(b) + pk_bits(c)); }
/* |a - b| of non-negative fields without a signed negate: max - min */
DBK_HD pk absdiff(pk a, pk b) { return usub(pk_max(a, b), pk_min(a, b)); }
/* a*K + c and a*K + C in ONE v_pk_mad_i16 (K, C small compile-time constants = inline operands).
 * hipcc canonicalises x*2 into a shift and splits mul/add, so the instruction is written out. */
template <int K>
DBK_HD pk mad_k(pk a, pk c)
{
#if DBK_DEV
    pk d;
    asm("v_pk_mad_i16 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "n"(K), "v"(c));
    return d;
#else
    return a * splat(K) + c;
#endif
}
template <int K, int C>
DBK_HD pk mad_kc(pk a)
{
#if DBK_DEV
    pk d;
    asm("v_pk_mad_i16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "n"(K), "n"(C));
    return d;
#else
    return a * splat(K) + splat(C);
#endif
}

/* (a | b) & c in ONE instruction (v_bitop3_b32; hipcc otherwise emits v_or + v_and, and a 2-cycle op between 4-cycle ops costs 4) */
DBK_HD uint32_t or_and(uint32_t a, uint32_t b, uint32_t c)
{
#if DBK_DEV
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xA8);
#else
    return (a | b) & c;
#endif
}

/* low half + high half of non-negative fields, as a 16-bit unsigned value */
DBK_HD unsigned lohi_sum(pk a)
{
    const uint32_t x = pk_bits(a);
    return (uint16_t)(x + (x >> 16));
}
DBK_HD pk pk_clamp(pk v, pk lo, pk hi) { return pk_min(pk_max(v, lo), hi); }

/* v_perm_b32: result byte i = selector byte i picks from {hi:lo}: 0..3 = lo bytes, 4..7 = hi bytes, 0x0c = 0x00 */
DBK_HD uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
#if DBK_DEV
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    const uint64_t src = ((uint64_t)hi << 32) | lo;
    uint32_t out = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t s = (sel >> (8 * i)) & 0xff;
        const uint32_t b = s <= 7 ? (uint32_t)((src >> (8 * s)) & 0xff) : 0u; /* only 0..7 and 0x0c are used here */
        out |= b << (8 * i);
    }
    return out;
#endif
}

/* the 8 taps of one line pair */
struct Taps {
    pk p0, p1, p2, p3, q0, q1, q2, q3;
};

/* Operand range of the packed luma core: besides the samples (<= 11 bit plain, 12 bit WIDE) the largest tc a lane can
 * see -- already scaled by 1 << (bit_depth - 8) -- must keep every 16-bit field in range.  The reference's tables stay far
 * inside (tc <= 20 << 4); caller-supplied tables (hevcdbk_tables) can hold entries up to 255:
 *   strong filter: the 5-tap sum carries the clip offset, 8*max_v + 4 + 8*(2*tc); it is shifted arithmetically
 *                  (< 2^15) in the plain core and logically (< 2^16) in the WIDE one;
 *   normal filter: the threshold 10*tc and (p2 + p0 + 1 - 2*p1) + 2*clip(delta, 2*tc) are signed 16-bit values.
 * Outside this range the launcher takes the 32-bit kernel. */
DBK_HD bool packed_luma_tc_fits(int max_v, int tc_max)
{
    const long long strong = 8ll * max_v + 4 + 16ll * tc_max;
    return strong <= (max_v > 2047 ? 65535 : 32767) && 10ll * tc_max <= 32767 && 2ll * max_v + 1 + 4ll * tc_max <= 32767;
}

/* ---- unpack ------------------------------------------------------------------------------------ */

/* vertical-edge segment (cpu.h:159-284): lines are rows ra (low half) and rb (high half);
 * P_k = col 3-k of the L dword, Q_k = col 4+k = byte k of the R dword */
DBK_HD Taps unpack_ver(uint32_t La, uint32_t Lb, uint32_t Ra, uint32_t Rb)
{
    Taps t;
    t.p0 = bits_pk(perm(Lb, La, 0x0c070c03u));
    t.p1 = bits_pk(perm(Lb, La, 0x0c060c02u));
    t.p2 = bits_pk(perm(Lb, La, 0x0c050c01u));
    t.p3 = bits_pk(perm(Lb, La, 0x0c040c00u));
    t.q0 = bits_pk(perm(Rb, Ra, 0x0c040c00u));
    t.q1 = bits_pk(perm(Rb, Ra, 0x0c050c01u));
    t.q2 = bits_pk(perm(Rb, Ra, 0x0c060c02u));
    t.q3 = bits_pk(perm(Rb, Ra, 0x0c070c03u));
    return t;
}

/* horizontal-edge segment (cpu.h:287-446): lines are columns; X holds the 4 line samples of one tap.
 * pair A = columns (0,3), pair B = columns (1,2) */
DBK_HD pk unpack_hor_a(uint32_t x) { return bits_pk(perm(x, x, 0x0c030c00u)); }
DBK_HD pk unpack_hor_b(uint32_t x) { return bits_pk(perm(x, x, 0x0c020c01u)); }

/* ---- decisions on pair A (cpu.h:1074-1114, 1243-1249) ------------------------------------------ */

struct Decision {
    bool filter; /* cond1 */
    bool strong; /* cond2 && cond3 && cond4 */
    bool cond5, cond6;
    pk tp1, tq1; /* p2 - 2p1 + p0 + 1 and q2 - 2q1 + q0 + 1 of pair A (signed): the normal filter's p1 / q1 terms start here */
};

/*
 * Everything a luma segment needs that depends on (beta, tc) and the filter mode only, behind one interface with two
 * shapes:
 *   LumaKLazy  -- wave-uniform beta / tc (one QP per launch; the spec-exact mode's uniform waves): holds the two scalars and
 *                 derives each operand where it is used, with a handful of scalar instructions.  Nothing but beta and tc
 *                 stays live across a segment: measured in round 3, the same instruction stream with all operands built at
 *                 the top of the segment needs 93 instead of 78 SGPRs and runs 1-2 % slower -- 800 SGPRs per SIMD are
 *                 handed out in blocks of 16, eight waves of 96 fill them to the brim and a new wave does not always find
 *                 its block free (17 % fewer waves resident on average, SQ_WAVE_CYCLES / SQ_BUSY_CYCLES);
 *   LumaKEager -- per-lane beta / tc (per-CTU QP map): the operands are vector registers, built once per segment.
 * H265 == false: the reference's filter (cpu.h);  H265 == true: the standard's (8.7.2.5.3 / .6 / .7) -- they differ in
 * the three thresholds of the strong decision, in the cond5 / cond6 threshold and in the clip of the normal delta.
 */
template <bool H265 = false>
struct LumaKLazy {
    int beta, tc;
    /* decisions.  reference: d(P,i)+d(Q,i) < beta/8, |p3-p0|+|q0-q3| < beta/8, |p0-q0| < 5*tc/2 (cpu.h:1099-1110);
     * standard: 2*(d(P,i)+d(Q,i)) < beta>>2, i.e. d < ceil((beta>>2)/2); ... < beta>>3; |p0-q0| < (5*tc+1)>>1 */
    DBK_HD int t_dpq() const { return H265 ? ((beta >> 2) + 1) >> 1 : beta >> 3; }
    DBK_HD int t_e() const { return beta >> 3; }
    DBK_HD int t_f() const { return H265 ? (5 * tc + 1) >> 1 : (5 * tc) >> 1; }
    DBK_HD uint32_t filter_thr() const { return (uint32_t)beta + 4u; }                            /* cpu.h:1086-1087; the field is d + 4 */
    /* biases in both halves: field + bias has bit 15 set <=> field >= its threshold */
    DBK_HD uint32_t km_dpq() const { return (uint32_t)(0x8000 - t_dpq() - 2) * 0x00010001u; }     /* the field is d + 2 */
    DBK_HD uint32_t km_e() const { return (uint32_t)(0x8000 - t_e()) * 0x00010001u; }
    DBK_HD uint32_t kf() const { return (uint32_t)(0x8000 - t_f()) * 0x00010001u; }
    DBK_HD bool strong_possible() const { return t_dpq() > 0 && t_e() > 0 && t_f() > 0; }
    DBK_HD uint32_t side_thr() const { return (uint32_t)(H265 ? (beta + (beta >> 1)) >> 3 : (3 * beta) >> 4) + 2u; } /* cpu.h:1243-1249, + 2 */
    /* strong filter: clip range c = 2*tc */
    DBK_HD uint32_t c16() const { return (uint32_t)(2 * tc) & 0xffffu; }
    DBK_HD pk sc2() const { return bits_pk((2u * c16()) * 0x00010001u); }
    DBK_HD uint32_t snegc() const { return 0u - c16() * 0x00010001u; }
    DBK_HD pk sk() const { return bits_pk((4u * c16() + 2u) * 0x00010001u); }                     /* 2 + 4c: carries the + c */
    /* normal filter: clip of delta 2*tc (reference, cpu.h:1256) / tc (standard); tc/2; 10*tc */
    DBK_HD int ncv() const { return H265 ? tc : 2 * tc; }
    DBK_HD pk nc() const { return splat_u(ncv()); }
    DBK_HD pk nnegc() const { return splat_u(-ncv()); }
    DBK_HD pk nc2() const { return splat_u(tc >> 1); }
    DBK_HD pk nnegc2() const { return splat_u(-(tc >> 1)); }
    DBK_HD pk nlim() const { return splat_u(10 * tc); }
    /* largest power of two <= 10*tc in both halves, and the bits at or above twice that */
    DBK_HD uint32_t p2() const { return tc > 0 ? 1u << (31 - __builtin_clz((unsigned)(10 * tc))) : 1u; }
    DBK_HD uint32_t nP() const { return p2() * 0x00010001u; }
    DBK_HD uint32_t nmask() const { return 0x00010001u * (0xffffu & ~(2u * p2() - 1u)); }
    DBK_HD bool tc_zero() const { return tc <= 0; }
};
struct LumaKEager {
    uint32_t filter_thr_, km_dpq_, km_e_, kf_, side_thr_, snegc_, nP_, nmask_;
    bool strong_possible_, tc_zero_;
    pk sc2_, sk_, nc_, nnegc_, nc2_, nnegc2_, nlim_;
    template <bool H265>
    DBK_HD static LumaKEager make(int beta, int tc)
    {
        const LumaKLazy<H265> z{beta, tc};
        LumaKEager k;
        k.filter_thr_ = z.filter_thr(); k.km_dpq_ = z.km_dpq(); k.km_e_ = z.km_e(); k.kf_ = z.kf(); k.side_thr_ = z.side_thr();
        k.snegc_ = z.snegc(); k.nP_ = z.nP(); k.nmask_ = z.nmask(); k.tc_zero_ = z.tc_zero();
        /* per lane the three comparisons would cost five instructions per segment and decide nothing: a threshold of zero
         * makes its bias 0x8000 - 0, which sets bit 15 of the biased field whatever the samples are (decide(): `bad`) */
        k.strong_possible_ = true;
        k.sc2_ = z.sc2(); k.sk_ = z.sk(); k.nc_ = z.nc(); k.nnegc_ = z.nnegc(); k.nc2_ = z.nc2(); k.nnegc2_ = z.nnegc2(); k.nlim_ = z.nlim();
        return k;
    }
    DBK_HD uint32_t filter_thr() const { return filter_thr_; }
    DBK_HD uint32_t km_dpq() const { return km_dpq_; }
    DBK_HD uint32_t km_e() const { return km_e_; }
    DBK_HD uint32_t kf() const { return kf_; }
    DBK_HD bool strong_possible() const { return strong_possible_; }
    DBK_HD uint32_t side_thr() const { return side_thr_; }
    DBK_HD pk sc2() const { return sc2_; }
    DBK_HD uint32_t snegc() const { return snegc_; }
    DBK_HD pk sk() const { return sk_; }
    DBK_HD pk nc() const { return nc_; }
    DBK_HD pk nnegc() const { return nnegc_; }
    DBK_HD pk nc2() const { return nc2_; }
    DBK_HD pk nnegc2() const { return nnegc2_; }
    DBK_HD pk nlim() const { return nlim_; }
    DBK_HD uint32_t nP() const { return nP_; }
    DBK_HD uint32_t nmask() const { return nmask_; }
    DBK_HD bool tc_zero() const { return tc_zero_; }
    /* the same operand set from the workgroup's table (below): ib = row of the beta-dependent operands, it = row of the
     * tc-dependent ones -- four 16-byte LDS reads and no arithmetic where make() spends about 45 VALU instructions */
    DBK_HD static LumaKEager load(const DBK_LDS uint32_t *tab, int ib, int it);
};

/* One beta, one of TWO tc values per lane (the spec-exact one-QP kernels in a wave that holds bS 1 next to bS 2: tc is the
 * table value of QP + 2 (bS - 1)): both operand sets are wave-uniform, so each tc-dependent operand is computed twice on the
 * scalar unit and picked per lane with one v_cndmask -- eleven selects per segment where LumaKEager::make spends about 45
 * VALU instructions; the beta-dependent operands stay scalars (round 4). */
template <bool H265>
struct LumaKSel {
    LumaKLazy<H265> z1, z2;
    bool second; /* this lane takes z2 */
    DBK_HD uint32_t filter_thr() const { return z1.filter_thr(); }
    DBK_HD uint32_t km_dpq() const { return z1.km_dpq(); }
    DBK_HD uint32_t km_e() const { return z1.km_e(); }
    DBK_HD uint32_t side_thr() const { return z1.side_thr(); }
    DBK_HD bool strong_possible() const { return true; } /* a zero threshold's bias says so by itself (LumaKEager::make) */
    DBK_HD uint32_t kf() const { return second ? z2.kf() : z1.kf(); }
    DBK_HD uint32_t snegc() const { return second ? z2.snegc() : z1.snegc(); }
    DBK_HD pk sc2() const { return bits_pk(second ? pk_bits(z2.sc2()) : pk_bits(z1.sc2())); }
    DBK_HD pk sk() const { return bits_pk(second ? pk_bits(z2.sk()) : pk_bits(z1.sk())); }
    DBK_HD uint32_t nP() const { return second ? z2.nP() : z1.nP(); }
    DBK_HD uint32_t nmask() const { return second ? z2.nmask() : z1.nmask(); }
    DBK_HD pk nc() const { return bits_pk(second ? pk_bits(z2.nc()) : pk_bits(z1.nc())); }
    DBK_HD pk nnegc() const { return bits_pk(second ? pk_bits(z2.nnegc()) : pk_bits(z1.nnegc())); }
    DBK_HD pk nc2() const { return bits_pk(second ? pk_bits(z2.nc2()) : pk_bits(z1.nc2())); }
    DBK_HD pk nnegc2() const { return bits_pk(second ? pk_bits(z2.nnegc2()) : pk_bits(z1.nnegc2())); }
    DBK_HD pk nlim() const { return bits_pk(second ? pk_bits(z2.nlim()) : pk_bits(z1.nlim())); }
    DBK_HD bool tc_zero() const { return second ? z2.tc_zero() : z1.tc_zero(); }
};

/*
 * Round 4: the per-lane operand set of a QP-map launch comes out of a TABLE.  Every operand is a function of beta alone or of
 * tc alone, beta and tc are table values of a QP index, so a launch has at most 52 (reference tables) / 54 (H.265 Table 8-12)
 * different operand rows of each kind.  A workgroup computes all of them once, in LDS, with its first 54 lanes (ktab_build:
 * about 60 instructions of one wave, then one barrier), and a segment's operands are then FOUR ds_read_b128 by the lane's
 * two indices -- the LDS pipe is otherwise idle in these kernels -- instead of LumaKEager::make per segment (4 x ~45 of the
 * 1043 VALU instructions per wave that made the QP-map kernels arithmetic-bound).  The eight per-lane byte loads
 * of tc_tab / beta_tab from the kernel arguments go away with it.
 *   rows 0 .. kKTabRows-1 of kKTabBetaDw dwords: filter_thr, km_dpq, km_e, side_thr
 *   then rows of kKTabTcDw dwords:               kf, snegc, sc2, sk | nP, nmask, nc, nnegc | nc2, nnegc2, nlim, (unused)
 *                                                (the decision's / strong filter's / normal filter's operands side by side)
 */
constexpr int kKTabRows = 54, kKTabBetaDw = 4, kKTabTcDw = 12;
constexpr int kKTabTcBase = kKTabRows * kKTabBetaDw;
constexpr int kKTabDwords = kKTabRows * (kKTabBetaDw + kKTabTcDw); /* 864 dwords = 3456 bytes */

template <bool H265>
DBK_HD void ktab_fill_beta(uint32_t *o, int beta)
{
    const LumaKLazy<H265> z{beta, 0};
    o[0] = z.filter_thr(); o[1] = z.km_dpq(); o[2] = z.km_e(); o[3] = z.side_thr();
}
template <bool H265>
DBK_HD void ktab_fill_tc(uint32_t *o, int tc)
{
    const LumaKLazy<H265> z{0, tc};
    o[0] = z.kf(); o[1] = z.snegc(); o[2] = pk_bits(z.sc2()); o[3] = pk_bits(z.sk());
    o[4] = z.nP(); o[5] = z.nmask(); o[6] = pk_bits(z.nc()); o[7] = pk_bits(z.nnegc());
    o[8] = pk_bits(z.nc2()); o[9] = pk_bits(z.nnegc2()); o[10] = pk_bits(z.nlim()); o[11] = 0u;
}
/* thread `tid` of `nthreads` fills its share of the rows: beta_of(i) / tc_of(i) = the (bit-depth scaled) table values of index i */
template <bool H265, class FB, class FT>
DBK_HD void ktab_build(uint32_t *tab, int tid, int nthreads, FB beta_of, FT tc_of)
{
    for (int i = tid; i < kKTabRows; i += nthreads) {
        ktab_fill_beta<H265>(tab + i * kKTabBetaDw, beta_of(i));
        ktab_fill_tc<H265>(tab + kKTabTcBase + i * kKTabTcDw, tc_of(i));
    }
}

/* The operand set of a segment as two row pointers into the table: every operand is read where it is used, so the decision's
 * five operands, the strong filter's three and the normal filter's seven are never all live at once (15 VGPRs when loaded up
 * front: 71 registers and 7 waves per SIMD for the QP-map kernel, against 8 waves with the reads placed by use); the compiler
 * merges neighbouring reads of one basic block into ds_read_b64 / b128. */
struct LumaKLds {
    const DBK_LDS uint32_t *b, *t; /* beta row, tc row */
    DBK_HD uint32_t filter_thr() const { return b[0]; }
    DBK_HD uint32_t km_dpq() const { return b[1]; }
    DBK_HD uint32_t km_e() const { return b[2]; }
    DBK_HD uint32_t side_thr() const { return b[3]; }
    DBK_HD uint32_t kf() const { return t[0]; }
    DBK_HD bool strong_possible() const { return true; } /* a zero threshold's bias says so by itself (LumaKEager::make) */
    DBK_HD uint32_t snegc() const { return t[1]; }
    DBK_HD pk sc2() const { return bits_pk(t[2]); }
    DBK_HD pk sk() const { return bits_pk(t[3]); }
    DBK_HD uint32_t nP() const { return t[4]; }
    DBK_HD uint32_t nmask() const { return t[5]; }
    DBK_HD pk nc() const { return bits_pk(t[6]); }
    DBK_HD pk nnegc() const { return bits_pk(t[7]); }
    DBK_HD pk nc2() const { return bits_pk(t[8]); }
    DBK_HD pk nnegc2() const { return bits_pk(t[9]); }
    DBK_HD pk nlim() const { return bits_pk(t[10]); }
    DBK_HD bool tc_zero() const { return t[10] == 0u; } /* nlim = 10 * tc in both halves */
    DBK_HD static LumaKLds rows(const DBK_LDS uint32_t *tab, int ib, int it)
    {
        return LumaKLds{tab + ib * kKTabBetaDw, tab + kKTabTcBase + it * kKTabTcDw};
    }
};

DBK_HD LumaKEager LumaKEager::load(const DBK_LDS uint32_t *tab, int ib, int it)
{
    LumaKEager k;
    const DBK_LDS uint32_t *br = tab + ib * kKTabBetaDw, *tr = tab + kKTabTcBase + it * kKTabTcDw;
#if DBK_DEV
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u3 __attribute__((ext_vector_type(3)));
    const u4 b = *(const DBK_LDS u4 *)br, t0 = *(const DBK_LDS u4 *)tr, t1 = *(const DBK_LDS u4 *)(tr + 4);
    const u3 t2 = *(const DBK_LDS u3 *)(tr + 8); /* 12 bytes: the row's last dword is padding, and a register not spent on it */
#else
    struct { uint32_t x, y, z, w; } b{br[0], br[1], br[2], br[3]}, t0{tr[0], tr[1], tr[2], tr[3]}, t1{tr[4], tr[5], tr[6], tr[7]},
        t2{tr[8], tr[9], tr[10], 0u};
#endif
    k.filter_thr_ = b.x; k.km_dpq_ = b.y; k.km_e_ = b.z; k.side_thr_ = b.w;
    k.kf_ = t0.x; k.snegc_ = t0.y; k.sc2_ = bits_pk(t0.z); k.sk_ = bits_pk(t0.w);
    k.nP_ = t1.x; k.nmask_ = t1.y; k.nc_ = bits_pk(t1.z); k.nnegc_ = bits_pk(t1.w);
    k.nc2_ = bits_pk(t2.x); k.nnegc2_ = bits_pk(t2.y); k.nlim_ = bits_pk(t2.z);
    k.tc_zero_ = t2.z == 0u; /* nlim = 10 * tc in both halves */
    k.strong_possible_ = true; /* as in make(): a zero threshold's bias says so by itself */
    return k;
}

/*
 * Round 3 form.  Everything carries a small constant so that no instruction exists only to add one:
 *   tp1 = p2 + p0 + 1 - 2*p1 (the "+ 1" is the rounding term of the normal filter's p1 update, cpu.h:1281, and rides in the
 *         three-operand add that forms p2 + p0);  |tp| + 1 = max(tp1, 2 - tp1)  -- still two instructions;
 *   every threshold the +1 / +2 reaches is a wave-uniform scalar and moves with it.
 * |a - b| of unsigned fields = (a -sat b) + (b -sat a) (one of the two is 0): the two saturating subtractions replace
 * max / min / sub, and the sum rides in the three-operand add that follows anyway.
 * "field < T" tests: field + (0x8000 - T) has bit 15 set iff field >= T; the bias rides in the same adds, and the biased
 * fields of the two tests that share... nothing but the mask are merged by an UNSIGNED packed max.
 */
template <class K>
DBK_HD Decision decide(const Taps &a, const K &k)
{
    Decision d;
    const pk tp1 = mad_k<-2>(a.p1, uadd3c(a.p2, a.p0, 0x00010001u)), tq1 = mad_k<-2>(a.q1, uadd3c(a.q2, a.q0, 0x00010001u));
    const pk dp1 = pk_max(tp1, splat(2) - tp1), dq1 = pk_max(tq1, splat(2) - tq1); /* |.| + 1 on lines 0 and 3 */
    d.tp1 = tp1;
    d.tq1 = tq1;
    /* sum of the two halves in the low 16 bits: x + (x >> 16), compared as a 16-bit value */
    d.filter = lohi_sum(uadd(dp1, dq1)) < k.filter_thr();
    const pk dpqk = uadd3c(dp1, dq1, k.km_dpq());                        /* d(P,i) + d(Q,i), biased */
    const pk ek = uadd3c(uadd3(sub_sat(a.p3, a.p0), sub_sat(a.p0, a.p3), sub_sat(a.q0, a.q3)), sub_sat(a.q3, a.q0), k.km_e());
    const pk fk = uadd3c(sub_sat(a.p0, a.q0), sub_sat(a.q0, a.p0), k.kf());
    /* bit 15 of a half set <=> that line violates a strong-filter condition */
    const uint32_t bad = or_and(pk_bits(pk_maxu(dpqk, ek)), pk_bits(fk), 0x80008000u);
    d.strong = k.strong_possible() && bad == 0u;
    d.cond5 = lohi_sum(dp1) < k.side_thr();
    d.cond6 = lohi_sum(dq1) < k.side_thr();
    return d;
}

/* ---- filters on one line pair ------------------------------------------------------------------- */

/* x >> N of fields that are non-negative but may exceed 2^15 (12-bit sums): 32-bit logical shift + field mask (two
 * 2-cycle instructions; the packed arithmetic shift would read such a field as negative) */
template <int N>
DBK_HD pk lsr(pk x) { return bits_pk((pk_bits(x) >> N) & (0x00010001u * (0xffffu >> N))); }
/* WIDE = bit depth 12 (max_v <= 4095): the strong filter's sums reach 38,908 and 9(q0-p0) - 3(q1-p1) + 8 reaches 49,148,
 * both beyond int16; everything else still fits.  WIDE == false is the <= 11-bit code, unchanged. */
template <bool WIDE, int N>
DBK_HD pk shr_sum(pk x)
{
    if constexpr (WIDE) return lsr<N>(x);
    else return x >> N;
}

/* strong filter (cpu.h:1152-1211), c = 2*tc.
 *
 * p' = clip(s, p - c, p + c) is evaluated as p + min(max((s + c) - p, 0), 2c) - c: the "+ c" rides for free on the rounding
 * constants of the sums (s0 and s1 carry the shared p0+q0+2 term twice / once under >>3 / >>2, so that term takes 2 + 4c; s2
 * gets the same constant once more); max(. - p, 0) is ONE saturating unsigned subtraction, the cap 2c and the final - c are
 * wave-uniform operands, and p + x - c is one three-operand add on the packed register: both fields of p + x are >= c (x = 0
 * only where s < p - c, i.e. p > c, because s >= 0), so no borrow crosses the field boundary.  Three instructions per output
 * (round 1-2: max, add, min, sub).  All fields stay non-negative and below 2^15 (8*max_v + 4 + 8c; WIDE: below 2^16). */
template <bool WIDE = false, bool UNI = true, class K>
DBK_HD void strong_pair(Taps &t, const K &lk)
{
    const pk k = lk.sk(), c2 = lk.sc2();
    const uint32_t negc = lk.snegc();
    const pk u2 = uadd(uadd(t.p0, t.q0), k);
    const pk tp = uadd(u2, t.p1);       /* p1+p0+q0+2 (+4c) */
    const pk tq = uadd(u2, t.q1);
    const pk bp = uadd(tp, t.p2);       /* p2+p1+p0+q0+2 (+4c) */
    const pk bq = uadd(tq, t.q2);
    const pk p32 = uadd(t.p3, t.p2), q32 = uadd(t.q3, t.q2);
    const pk s0p = shr_sum<WIDE, 3>(uadd(uadd(tp, bp), t.q1));              /* (p2+2p1+2p0+2q0+q1+4)>>3  + c */
    const pk s1p = shr_sum<WIDE, 2>(bp);                                    /* (p2+p1+p0+q0+2)>>2        + c */
    const pk s2p = shr_sum<WIDE, 3>(uadd(uadd(uadd(p32, p32), bp), k));     /* (2p3+3p2+p1+p0+q0+4)>>3   + c */
    const pk s0q = shr_sum<WIDE, 3>(uadd(uadd(tq, bq), t.p1));
    const pk s1q = shr_sum<WIDE, 2>(bq);
    const pk s2q = shr_sum<WIDE, 3>(uadd(uadd(uadd(q32, q32), bq), k));
    /* hipcc splits p + x + negc into (p - c) + x, two instructions: the three-operand add is written out.  UNI: -c is
     * wave-uniform and sits in an SGPR; otherwise (per-CTU QP map, per-lane bS in the spec-exact mode) it is a vector
     * register -- an "s" constraint on a divergent value would let a toolchain legalise it with v_readfirstlane and use
     * lane 0's tc for every lane */
    auto fin = [&](pk s, pk p) {
        const pk x = pk_min(sub_sat(s, p), c2);
#if DBK_DEV
        uint32_t d;
        if constexpr (UNI) asm("v_add3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(pk_bits(p)), "v"(pk_bits(x)), "s"(negc));
        else asm("v_add3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(pk_bits(p)), "v"(pk_bits(x)), "v"(negc));
        return bits_pk(d);
#else
        return bits_pk(pk_bits(p) + pk_bits(x) + negc);
#endif
    };
    const pk np0 = fin(s0p, t.p0), np1 = fin(s1p, t.p1), np2 = fin(s2p, t.p2);
    const pk nq0 = fin(s0q, t.q0), nq1 = fin(s1q, t.q1), nq2 = fin(s2q, t.q2);
    t.p0 = np0; t.p1 = np1; t.p2 = np2;
    t.q0 = nq0; t.q1 = nq1; t.q2 = nq2;
}

/* (9(q0-p0) - 3(q1-p1) + 8) >> 4 (cpu.h:1253) as two multiply-adds (v_pk_mad_i16) */
template <bool WIDE>
DBK_HD pk normal_delta(const Taps &t)
{
    if constexpr (WIDE) {
        /* 9a - 3b + 8 = 8a + r with r = a - 3b + 8 (|r| <= 16,388): (8a + r) >> 4 == (a + (r >> 3)) >> 1, floors included:
         * r = 8t + rho, 0 <= rho < 8, adds rho/16 < 1/2 to (a + t)/2, which cannot reach the next integer */
        const pk a = t.q0 - t.p0;
        const pk r = mad_k<-3>(t.q1 - t.p1, a + splat(8));
        return (a + (r >> 3)) >> 1;
    } else {
        return mad_k<9>(t.q0 - t.p0, mad_kc<-3, 8>(t.q1 - t.p1)) >> 4;
    }
}

/* normal filter (cpu.h:1251-1354) of one line pair given its delta: the three deltas it applies -- D to p0 / q0 (cpu.h:1256),
 * dp1 / dq1 to p1 / q1 where cond5 / cond6 hold (cpu.h:1281-1300).  ALL_ON: the caller has established that
 * |delta| < 10*tc (cpu.h:1254) holds in every line of every lane of the wave, so no per-line mask is needed */
struct NormalD {
    pk D, dp1, dq1;
};
template <bool ALL_ON, bool HAVE_T = false, class K>
DBK_HD NormalD normal_deltas(const Taps &t, pk delta, const K &k, pk tp = pk{0, 0}, pk tq = pk{0, 0})
{
    NormalD n;
    pk on = pk{0, 0};
    if constexpr (ALL_ON) {
        n.D = pk_min(pk_max(delta, k.nnegc()), k.nc());
    } else {
        /* all ones where |delta| < 10*tc (cpu.h:1254).  The mask goes onto delta BEFORE the clip (clip(0) = 0): written as
         * clip(delta) & on, the clip is an expression both forms share, the compiler hoists it ahead of the branch, and the
         * unmasked form then pays a register copy at the join (one v_mov per segment in the path every wave takes) */
        on = (pk_abs(delta) - k.nlim()) >> 15;
        n.D = pk_min(pk_max(delta & on, k.nnegc()), k.nc());
    }
    /* (((p2+p0+1)>>1) - p1 + D) >> 1  ==  (p2 + p0 + 1 - 2*p1 + 2*D) >> 2   (floor of a floor: the dropped
     * bit of the inner shift is worth 1/4 and cannot carry across an integer).  HAVE_T: p2 + p0 + 1 - 2*p1 (and the Q twin)
     * of this pair is already there from the decisions (pair A) */
    pk ip, iq;
    if constexpr (HAVE_T) { /* tp / tq already carry the + 1 (decide) */
        ip = tp;
        iq = tq;
    } else {
        ip = mad_k<-2>(t.p1, uadd3c(t.p2, t.p0, 0x00010001u));
        iq = mad_k<-2>(t.q1, uadd3c(t.q2, t.q0, 0x00010001u));
    }
    n.dp1 = pk_min(pk_max(mad_k<2>(n.D, ip) >> 2, k.nnegc2()), k.nc2());
    n.dq1 = pk_min(pk_max(mad_k<-2>(n.D, iq) >> 2, k.nnegc2()), k.nc2());
    if constexpr (!ALL_ON) {
        n.dp1 = n.dp1 & on;
        n.dq1 = n.dq1 & on;
    }
    return n;
}
/* m5 / m6 = 1 in both halves where cond5 / cond6 hold, else 0 (one_of): p1 + dp1 * m5 is ONE multiply-add instead of a mask
 * and an add.  Up to, but not including, the final Clip2 to [0, max_v] */
DBK_HD void normal_update(Taps &t, const NormalD &n, pk m5, pk m6)
{
    t.p0 = t.p0 + n.D;
    t.q0 = t.q0 - n.D;
    t.p1 = mad_vvv(n.dp1, m5, t.p1);
    t.q1 = mad_vvv(n.dq1, m6, t.q1);
}

/* wave-level "does any active lane say yes": on the GPU one ballot; in the CPU build of this header (one block at a
 * time) the lane's own answer */
DBK_HD bool any_lane(bool v)
{
#if DBK_DEV
    return __builtin_amdgcn_ballot_w64(v) != 0ull;
#else
    return v;
#endif
}

/* the lanes of the wave that say yes, as a mask (CPU build: this block's own answer in bit 0) */
DBK_HD unsigned long long lane_ballot(bool v)
{
#if DBK_DEV
    return __builtin_amdgcn_ballot_w64(v);
#else
    return v ? 1ull : 0ull;
#endif
}

/* both pairs of a normal-filtered segment.  The final Clip2 (cpu.h:1268-1275) only ever acts on samples
 * within 2*tc of 0 or max_v; one OR over the eight results shows whether any field left [0, max_v]
 * (a negative field has its top bits set, a too-large one has a bit above max_v), and the 16
 * min/max instructions run only in waves where some lane needs them.  Likewise the |delta| < 10*tc switch of a line
 * (cpu.h:1254): a line fails it only across a real picture edge, so the per-line masks are built only in waves where
 * some lane has such a line. */
template <bool WIDE = false, bool HAVE_T = false, class K>
DBK_HD void normal_pairs(Taps &a, Taps &b, const K &k, pk m5, pk m6, int max_v, pk tp = pk{0, 0}, pk tq = pk{0, 0})
{
    const pk da = normal_delta<WIDE>(a), db = normal_delta<WIDE>(b);
    /* One test for the four lines of the segment, and a conservative one: with P the largest power of two <= 10*tc,
     * |delta| < P in every line <=> (delta + P) has no bit at or above 2P in any half -- two adds, one OR-AND, one compare
     * (round 2: the exact test per pair, 3 + 3).  A wave in which some line has P <= |delta| takes the masked form, which is
     * exact for every line; on picture content |delta| >= 10*tc/2 happens across real edges only.  The wave-uniform tc == 0
     * case joins the ballot as a scalar OR. */
    const uint32_t wide = or_and(pk_bits(da + bits_pk(k.nP())), pk_bits(db + bits_pk(k.nP())), k.nmask());
    NormalD na, nb;
    if (__builtin_expect(any_lane(k.tc_zero() || wide != 0u), 0)) {
        na = normal_deltas<false, HAVE_T>(a, da, k, tp, tq);
        nb = normal_deltas<false, false>(b, db, k);
    } else {
        na = normal_deltas<true, HAVE_T>(a, da, k, tp, tq);
        nb = normal_deltas<true, false>(b, db, k);
    }
    normal_update(a, na, m5, m6); /* the updates themselves: once, behind the join */
    normal_update(b, nb, m5, m6);
    const uint32_t over = (pk_bits(a.p0) | pk_bits(a.q0) | pk_bits(a.p1) | pk_bits(a.q1) |
                           pk_bits(b.p0) | pk_bits(b.q0) | pk_bits(b.p1) | pk_bits(b.q1)) &
                          (0x00010001u * (0xffffu & ~(uint32_t)max_v));
    if (over) {
        const pk zero = splat(0), maxv = splat(max_v);
        a.p0 = pk_clamp(a.p0, zero, maxv); a.q0 = pk_clamp(a.q0, zero, maxv);
        a.p1 = pk_clamp(a.p1, zero, maxv); a.q1 = pk_clamp(a.q1, zero, maxv);
        b.p0 = pk_clamp(b.p0, zero, maxv); b.q0 = pk_clamp(b.q0, zero, maxv);
        b.p1 = pk_clamp(b.p1, zero, maxv); b.q1 = pk_clamp(b.q1, zero, maxv);
    }
}

/* one luma segment given its two unpacked pairs; returns false when nothing changed */
/* ablate (diagnostic builds of the benchmark only, 0 in the product): 1 = treat strong segments as
 * normal, 2 = skip the normal filter -- wrong pixels, used to price each path on the GPU */
/* enable: a per-lane switch that joins the filter decision (the spec-exact mode's bS == 0: a lane-level `if` AROUND this
 * function nests its divergent region inside another one, and the compiler then initialises the twelve result registers
 * with copies of the taps twice per segment -- 96 v_mov_b32 per block in the spec-exact kernels of round 3) */
template <bool WIDE = false, bool UNI = true, class K>
DBK_HD bool luma_pairs(Taps &a, Taps &b, const K &k, int max_v = 255, int ablate = 0, bool enable = true)
{
    const Decision d = decide(a, k);
    if (!(d.filter && enable)) return false;
    if (d.strong && ablate != 1) {
        strong_pair<WIDE, UNI>(a, k);
        strong_pair<WIDE, UNI>(b, k);
    } else if (ablate != 2) {
        normal_pairs<WIDE, true>(a, b, k, one_of(d.cond5), one_of(d.cond6), max_v, d.tp1, d.tq1);
    }
    return true;
}

/* ---- chroma (cpu.h:1431-1488): p0/q0 only, no decisions ------------------------------------------- */

DBK_HD void chroma_pair(pk &p0, pk p1, pk &q0, pk q1, pk tc, int max_v = 255)
{
    const pk zero = splat(0), maxv = splat(max_v);
    const pk dp = (((p0 - q0) << 2) + p1 - q1 + splat(4)) >> 3; /* cpu.h:1453 */
    const pk dq = (((q0 - p0) << 2) + q1 - p1 + splat(4)) >> 3; /* cpu.h:1458 */
    const pk np0 = pk_clamp(p0 + pk_clamp(dp, zero - tc, tc), zero, maxv);
    const pk nq0 = pk_clamp(q0 - pk_clamp(dq, zero - tc, tc), zero, maxv);
    p0 = np0;
    q0 = nq0;
}

template <int R0>
DBK_HD void chroma_ver(uint32_t (&L)[8], uint32_t (&R)[8], int tc)
{
    const pk c = splat(tc);
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int ra = R0 + (h ? 1 : 0), rb = R0 + (h ? 2 : 3);
        pk p0 = bits_pk(perm(L[rb], L[ra], 0x0c070c03u)), p1 = bits_pk(perm(L[rb], L[ra], 0x0c060c02u));
        pk q0 = bits_pk(perm(R[rb], R[ra], 0x0c040c00u)), q1 = bits_pk(perm(R[rb], R[ra], 0x0c050c01u));
        chroma_pair(p0, p1, q0, q1, c);
        /* put p0 into byte 3 of the L rows, q0 into byte 0 of the R rows */
        L[ra] = perm(pk_bits(p0), L[ra], 0x04020100u);
        L[rb] = perm(pk_bits(p0), L[rb], 0x06020100u);
        R[ra] = perm(pk_bits(q0), R[ra], 0x03020104u);
        R[rb] = perm(pk_bits(q0), R[rb], 0x03020106u);
    }
}

DBK_HD void chroma_hor(uint32_t (&PX)[8], uint32_t (&L)[8], int tc)
{
    const pk c = splat(tc);
    pk ap0 = unpack_hor_a(PX[3]), ap1 = unpack_hor_a(PX[2]), aq0 = unpack_hor_a(L[4]), aq1 = unpack_hor_a(L[5]);
    pk bp0 = unpack_hor_b(PX[3]), bp1 = unpack_hor_b(PX[2]), bq0 = unpack_hor_b(L[4]), bq1 = unpack_hor_b(L[5]);
    chroma_pair(ap0, ap1, aq0, aq1, c);
    chroma_pair(bp0, bp1, bq0, bq1, c);
    PX[3] = perm(pk_bits(bp0), pk_bits(ap0), 0x02060400u);
    L[4] = perm(pk_bits(bq0), pk_bits(aq0), 0x02060400u);
}

/* ---- luma block with the int16 pairs kept across ver -> hor ------------------------------------------
 *
 * The ver segments leave their 32 tap registers (pair = two rows, tap = column) in place; a hor
 * segment wants pair = two columns, tap = row.  One v_perm_b32 per hor register picks the two halves
 * it needs from two ver registers, so the block is never re-packed to bytes between the vertical and
 * the horizontal edges: 32 unpack + 24 pick + 20 final-pack perms instead of 64 + 44.
 *
 * Row r of the block lives in: rows 0,3 = lo,hi of va1 ; rows 1,2 = lo,hi of vb1 ;
 *                              rows 4,7 = lo,hi of va2 ; rows 5,6 = lo,hi of vb2.
 * Column c of the block is ver tap: cols 0..3 = p3,p2,p1,p0 ; cols 4..7 = q0,q1,q2,q3.
 */
DBK_HD pk pick_lo(pk x, pk y) { return bits_pk(perm(pk_bits(y), pk_bits(x), 0x05040100u)); } /* (x.lo, y.lo) */
DBK_HD pk pick_hi(pk x, pk y) { return bits_pk(perm(pk_bits(y), pk_bits(x), 0x07060302u)); } /* (x.hi, y.hi) */
DBK_HD uint32_t row_of(pk a, pk b) { return perm(pk_bits(b), pk_bits(a), 0x02060400u); }     /* [a.lo, b.lo, b.hi, a.hi] */

/* diagnostic only (ablate == 4): two workgroup barriers per stage, to price the synchronisation a
 * workgroup-level exchange between the stages would need */
DBK_HD void diag_barriers(int ablate)
{
#if DBK_DEV
    if (ablate == 4) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
    }
#else
    (void)ablate;
#endif
}

/* the four segments on already-unpacked ver registers; leaves the final values in
 * ha/hb (cols 0..3 of rows 0..3 as P, taps p3..p0), ga/gb (P = cols 4..7 of rows 0..3, Q = cols 0..3 of
 * rows 4..7) and va2/vb2 q taps (cols 4..7 of rows 4..7) */
/* one reference-exact luma segment from its tc / beta: wave-uniform values take the lazy operand set, per-lane ones the eager */
template <bool WIDE, bool UNI>
DBK_HD void luma_seg(Taps &a, Taps &b, int beta, int tc, int max_v, int ablate)
{
    if constexpr (UNI) luma_pairs<WIDE, true>(a, b, LumaKLazy<false>{beta, tc}, max_v, ablate);
    else luma_pairs<WIDE, false>(a, b, LumaKEager::make<false>(beta, tc), max_v, ablate);
}

/* Where the operands of segment s of a block come from:
 *   QsValues<UNI>  beta / tc values.  UNI: one QP for the whole launch (q.beta[] / q.tc[] hold four copies of two scalars), the
 *                  segment constants are wave-uniform and live in SGPRs; otherwise per-lane values, built per segment;
 *   QsTable        per-lane INDICES into the workgroup's operand table (ktab_build): QP-map launches, round 4. */
template <bool UNI>
struct QsValues {
    const BlockQp &q;
    template <bool WIDE>
    DBK_HD void seg(int s, Taps &a, Taps &b, int max_v, int ablate) const { luma_seg<WIDE, UNI>(a, b, q.beta[s], q.tc[s], max_v, ablate); }
};
struct QsTable {
    const DBK_LDS uint32_t *tab;
    int ib[4], it[4];
    template <bool WIDE>
    DBK_HD void seg(int s, Taps &a, Taps &b, int max_v, int ablate) const
    {
        luma_pairs<WIDE, false>(a, b, LumaKLds::rows(tab, ib[s], it[s]), max_v, ablate);
    }
};

template <bool WIDE, class QS>
DBK_HD void luma_block_core_src(Taps &va1, Taps &vb1, Taps &va2, Taps &vb2, const BlockBs &bs, const QS &q,
                                int max_v, Taps &ha, Taps &hb, Taps &ga, Taps &gb, int ablate = 0)
{
    if (bs.ver1 > 0) q.template seg<WIDE>(0, va1, vb1, max_v, ablate); /* cpu.h:164 */
    if (bs.ver2 > 0) q.template seg<WIDE>(1, va2, vb2, max_v, ablate); /* cpu.h:228 */
    diag_barriers(ablate);

    /* hor1: lines = cols 0..3, pair A = cols (0,3) = ver taps (p3,p0), pair B = cols (1,2) = (p2,p1);
     * P_k = row 3-k, Q_k = row 4+k (cpu.h:287-365) */
    ha.p0 = pick_hi(va1.p3, va1.p0); hb.p0 = pick_hi(va1.p2, va1.p1); /* row 3 */
    ha.p1 = pick_hi(vb1.p3, vb1.p0); hb.p1 = pick_hi(vb1.p2, vb1.p1); /* row 2 */
    ha.p2 = pick_lo(vb1.p3, vb1.p0); hb.p2 = pick_lo(vb1.p2, vb1.p1); /* row 1 */
    ha.p3 = pick_lo(va1.p3, va1.p0); hb.p3 = pick_lo(va1.p2, va1.p1); /* row 0 */
    ha.q0 = pick_lo(va2.p3, va2.p0); hb.q0 = pick_lo(va2.p2, va2.p1); /* row 4 */
    ha.q1 = pick_lo(vb2.p3, vb2.p0); hb.q1 = pick_lo(vb2.p2, vb2.p1); /* row 5 */
    ha.q2 = pick_hi(vb2.p3, vb2.p0); hb.q2 = pick_hi(vb2.p2, vb2.p1); /* row 6 */
    ha.q3 = pick_hi(va2.p3, va2.p0); hb.q3 = pick_hi(va2.p2, va2.p1); /* row 7 */
    if (bs.hor1 > 0) q.template seg<WIDE>(2, ha, hb, max_v, ablate); /* cpu.h:292 */
    diag_barriers(ablate);

    /* hor2: P lines = cols 4..7 (ver taps q0..q3) of rows 3..0, pair A = cols (4,7), B = cols (5,6);
     * Q = the same registers hor1 just used for its Q side: cols 0..3 of rows 4..7 (cpu.h:368-446) */
    ga.p0 = pick_hi(va1.q0, va1.q3); gb.p0 = pick_hi(va1.q1, va1.q2); /* row 3 */
    ga.p1 = pick_hi(vb1.q0, vb1.q3); gb.p1 = pick_hi(vb1.q1, vb1.q2); /* row 2 */
    ga.p2 = pick_lo(vb1.q0, vb1.q3); gb.p2 = pick_lo(vb1.q1, vb1.q2); /* row 1 */
    ga.p3 = pick_lo(va1.q0, va1.q3); gb.p3 = pick_lo(va1.q1, va1.q2); /* row 0 */
    ga.q0 = ha.q0; ga.q1 = ha.q1; ga.q2 = ha.q2; ga.q3 = ha.q3;
    gb.q0 = hb.q0; gb.q1 = hb.q1; gb.q2 = hb.q2; gb.q3 = hb.q3;
    if (bs.hor2 > 0) q.template seg<WIDE>(3, ga, gb, max_v, ablate); /* cpu.h:373 */
    diag_barriers(ablate);
}
template <bool WIDE = false, bool UNI = true>
DBK_HD void luma_block_core(Taps &va1, Taps &vb1, Taps &va2, Taps &vb2, const BlockBs &bs, const BlockQp &q,
                            int max_v, Taps &ha, Taps &hb, Taps &ga, Taps &gb, int ablate = 0)
{
    luma_block_core_src<WIDE>(va1, vb1, va2, vb2, bs, QsValues<UNI>{q}, max_v, ha, hb, ga, gb, ablate);
}

/* 8-bit samples: L[r] = cols 0..3, R[r] = cols 4..7 of row r as bytes */
template <class QS>
DBK_HD void packed_filter_luma_block_src(uint32_t (&L)[8], uint32_t (&R)[8], const BlockBs &bs, const QS &q, int ablate = 0)
{
    Taps va1 = unpack_ver(L[0], L[3], R[0], R[3]), vb1 = unpack_ver(L[1], L[2], R[1], R[2]);
    Taps va2 = unpack_ver(L[4], L[7], R[4], R[7]), vb2 = unpack_ver(L[5], L[6], R[5], R[6]);
    Taps ha, hb, ga, gb;
    luma_block_core_src<false>(va1, vb1, va2, vb2, bs, q, 255, ha, hb, ga, gb, ablate);

    /* final pack, once per row dword */
    L[0] = row_of(ha.p3, hb.p3); L[1] = row_of(ha.p2, hb.p2); L[2] = row_of(ha.p1, hb.p1); L[3] = row_of(ha.p0, hb.p0);
    L[4] = row_of(ga.q0, gb.q0); L[5] = row_of(ga.q1, gb.q1); L[6] = row_of(ga.q2, gb.q2); L[7] = row_of(ga.q3, gb.q3);
    R[0] = row_of(ga.p3, gb.p3); R[1] = row_of(ga.p2, gb.p2); R[2] = row_of(ga.p1, gb.p1); R[3] = row_of(ga.p0, gb.p0);
    {   /* cols 4..7 of rows 4..7: only ver2 touched them; [q0,q1,q2,q3] per row */
        const uint32_t u1 = perm(pk_bits(va2.q1), pk_bits(va2.q0), 0x06020400u);
        const uint32_t u2 = perm(pk_bits(va2.q3), pk_bits(va2.q2), 0x06020400u);
        R[4] = perm(u2, u1, 0x05040100u);
        R[7] = perm(u2, u1, 0x07060302u);
        const uint32_t w1 = perm(pk_bits(vb2.q1), pk_bits(vb2.q0), 0x06020400u);
        const uint32_t w2 = perm(pk_bits(vb2.q3), pk_bits(vb2.q2), 0x06020400u);
        R[5] = perm(w2, w1, 0x05040100u);
        R[6] = perm(w2, w1, 0x07060302u);
    }
}

template <bool UNI = true>
DBK_HD void packed_filter_luma_block(uint32_t (&L)[8], uint32_t (&R)[8], const BlockBs &bs, const BlockQp &q, int ablate = 0)
{
    packed_filter_luma_block_src(L, R, bs, QsValues<UNI>{q}, ablate);
}

/* 16-bit containers (bit depth 8..16): W[r][j] = columns (2j, 2j+1) of row r as two uint16.
 * Same arithmetic; only the moves in and out of the pair registers differ (samples are already
 * 16 bit wide, so a move is "pick two halves" instead of "pick two bytes and zero-extend"). */
DBK_HD Taps unpack_ver16(const uint32_t (&a)[4], const uint32_t (&b)[4])
{
    Taps t;
    t.p3 = pick_lo(bits_pk(a[0]), bits_pk(b[0])); t.p2 = pick_hi(bits_pk(a[0]), bits_pk(b[0])); /* cols 0,1 */
    t.p1 = pick_lo(bits_pk(a[1]), bits_pk(b[1])); t.p0 = pick_hi(bits_pk(a[1]), bits_pk(b[1])); /* cols 2,3 */
    t.q0 = pick_lo(bits_pk(a[2]), bits_pk(b[2])); t.q1 = pick_hi(bits_pk(a[2]), bits_pk(b[2])); /* cols 4,5 */
    t.q2 = pick_lo(bits_pk(a[3]), bits_pk(b[3])); t.q3 = pick_hi(bits_pk(a[3]), bits_pk(b[3])); /* cols 6,7 */
    return t;
}

template <bool WIDE, class QS>
DBK_HD void packed_filter_luma_block16_src(uint32_t (&W)[8][4], const BlockBs &bs, const QS &q, int max_v)
{
    Taps va1 = unpack_ver16(W[0], W[3]), vb1 = unpack_ver16(W[1], W[2]);
    Taps va2 = unpack_ver16(W[4], W[7]), vb2 = unpack_ver16(W[5], W[6]);
    Taps ha, hb, ga, gb;
    luma_block_core_src<WIDE>(va1, vb1, va2, vb2, bs, q, max_v, ha, hb, ga, gb);

    /* pair A = cols (0,3) / (4,7), pair B = cols (1,2) / (5,6):  (c0,c1) = (A.lo,B.lo), (c2,c3) = (B.hi,A.hi) */
#define DBK_ROW16(r, A, B, j)                                 \
    W[r][j] = pk_bits(pick_lo(A, B));                          \
    W[r][j + 1] = pk_bits(pick_hi(B, A));
    DBK_ROW16(0, ha.p3, hb.p3, 0) DBK_ROW16(1, ha.p2, hb.p2, 0) DBK_ROW16(2, ha.p1, hb.p1, 0) DBK_ROW16(3, ha.p0, hb.p0, 0)
    DBK_ROW16(4, ga.q0, gb.q0, 0) DBK_ROW16(5, ga.q1, gb.q1, 0) DBK_ROW16(6, ga.q2, gb.q2, 0) DBK_ROW16(7, ga.q3, gb.q3, 0)
    DBK_ROW16(0, ga.p3, gb.p3, 2) DBK_ROW16(1, ga.p2, gb.p2, 2) DBK_ROW16(2, ga.p1, gb.p1, 2) DBK_ROW16(3, ga.p0, gb.p0, 2)
#undef DBK_ROW16
    /* cols 4..7 of rows 4..7 from the ver2 registers: rows (4,7) = lo,hi of va2 ; rows (5,6) of vb2 */
    W[4][2] = pk_bits(pick_lo(va2.q0, va2.q1)); W[4][3] = pk_bits(pick_lo(va2.q2, va2.q3));
    W[7][2] = pk_bits(pick_hi(va2.q0, va2.q1)); W[7][3] = pk_bits(pick_hi(va2.q2, va2.q3));
    W[5][2] = pk_bits(pick_lo(vb2.q0, vb2.q1)); W[5][3] = pk_bits(pick_lo(vb2.q2, vb2.q3));
    W[6][2] = pk_bits(pick_hi(vb2.q0, vb2.q1)); W[6][3] = pk_bits(pick_hi(vb2.q2, vb2.q3));
}

template <bool WIDE = false, bool UNI = true>
DBK_HD void packed_filter_luma_block16(uint32_t (&W)[8][4], const BlockBs &bs, const BlockQp &q, int max_v)
{
    packed_filter_luma_block16_src<WIDE>(W, bs, QsValues<UNI>{q}, max_v);
}

/* ---- the whole block: ver1 -> ver2 -> hor1 -> hor2 --------------------------------------- */
template <bool CHROMA, bool UNI = true>
DBK_HD void packed_filter_block(uint32_t (&L)[8], uint32_t (&R)[8], const BlockBs &bs, const BlockQp &q, int ablate = 0)
{
    if constexpr (CHROMA) {
        if (bs.ver1 == 2) chroma_ver<0>(L, R, q.tc[0]);
        if (bs.ver2 == 2) chroma_ver<4>(L, R, q.tc[1]);
        if (bs.hor1 == 2) chroma_hor(L, L, q.tc[2]);
        if (bs.hor2 == 2) chroma_hor(R, L, q.tc[3]);
    } else {
        packed_filter_luma_block<UNI>(L, R, bs, q, ablate);
    }
}

} /* namespace dbk */
