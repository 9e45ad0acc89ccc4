/*
 * h265_oracle.h -- CPU restatement of the ITU-T H.265 (HEVC) in-loop deblocking filter, clause 8.7.2,
 * for the spec-exact mode of the HIP path.
 *
 * TEST INFRASTRUCTURE ONLY: only tests/ may call it; the product library never links or loads it.
 *
 * PARITY UNPINNED.  The reference (/root/reference/hevc_deblocking_filter) does not implement this mode: its
 * filter deviates from the standard in the thresholds of the strong-filter decision, the clip of the normal
 * filter's delta, the chroma formula, the hor2 column pairing, the frame-edge handling and the constant bS/QP
 * (SURVEY 8a Q-list).  No HEVC decoder, conformance stream or third-party implementation is present in this
 * image either.  This file restates the published text of clause 8.7.2 (edge order 8.7.2.1, bS 8.7.2.4, luma
 * decisions 8.7.2.5.3 / 8.7.2.5.6, luma filter 8.7.2.5.7, chroma 8.7.2.5.5 / 8.7.2.5.8, tables 8-12 and 8-10) and
 * is deliberately written in the standard's own order -- every vertical edge of the picture first, then every
 * horizontal edge on the result -- so that it shares no structure with the kernels' offset-block formulation.
 * What IS pinned: the line arithmetic it shares with the pinned reference-mode oracle is cross-checked in
 * tests/test_h265_oracle.py by running both on inputs where the two modes must agree.
 */
#ifndef H265_ORACLE_H
#define H265_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Table 8-12: beta' for Q = 0..51, tc' for Q = 0..53 */
extern const uint8_t dbko_h265_beta_table[52];
extern const uint8_t dbko_h265_tc_table[54];

/* bS entry: bits 1:0 = bS (0..2); bit 2 = P samples unmodified (nDp = 0: pcm_loop_filter_disabled_flag with a
 * PCM block, or cu_transquant_bypass_flag); bit 3 = Q samples unmodified (nDq = 0) */
#define DBKO_H265_BS_MASK   3u
#define DBKO_H265_KEEP_P    4u
#define DBKO_H265_KEEP_Q    8u

typedef struct {
    int tc_offset_div2;    /* slice_tc_offset_div2 (or pps_), -6..6 */
    int beta_offset_div2;  /* slice_beta_offset_div2, -6..6 */
    int c_qp_offset;       /* cQpPicOffset of THIS plane: pps_cb_qp_offset or pps_cr_qp_offset (-12..12) */
} dbko_h265_params;

/* sizes of the 4-sample-granular bS arrays of a plane_w x plane_h plane:
 * vert: (plane_w/8+1) columns x (plane_h/4) rows, entry (y4, bx) = edge x = 8*bx, rows 4*y4..4*y4+3
 * hor:  (plane_h/8+1) rows x (plane_w/4) columns, entry (by, x4) = edge y = 8*by, cols 4*x4..4*x4+3 */
size_t dbko_h265_num_vert_bs(unsigned plane_w, unsigned plane_h);
size_t dbko_h265_num_hor_bs(unsigned plane_w, unsigned plane_h);

/*
 * Deblock one plane in place.  Luma (c_idx 0): bS > 0 edges, 8.7.2.5.3/.6/.7.  Chroma (c_idx 1, 2; 4:2:0): bS == 2
 * edges on the plane's own 8-sample grid, 8.7.2.5.5/.8; its bS arrays have the CHROMA plane's geometry (see
 * dbko_h265_chroma_bs).  QP: the scalar `qp` when qp_map == NULL, else QpY of the unit covering a luma sample:
 * qp_map[(y >> unit_log2) * map_stride + (x >> unit_log2)] (chroma positions are doubled first).
 * Edges on the picture boundary are never filtered, whatever the arrays hold.
 */
int dbko_h265_filter_plane(void *plane, unsigned plane_w, unsigned plane_h, size_t pitch_bytes, unsigned bit_depth,
                           unsigned sample_bytes, int c_idx, const uint8_t *vert_bs4, const uint8_t *hor_bs4,
                           unsigned qp, const uint8_t *qp_map, unsigned map_stride, unsigned unit_log2,
                           const dbko_h265_params *prm);

/* ---- bS derivation, 8.7.2.4 ---- */

/* per 4x4 luma unit flags */
#define DBKO_U_INTRA        0x0001u /* CuPredMode == MODE_INTRA */
#define DBKO_U_CBF          0x0002u /* the luma transform block covering the unit has non-zero coefficient levels */
#define DBKO_U_TU_LEFT      0x0004u /* the unit's left border is a transform block edge */
#define DBKO_U_TU_TOP       0x0008u
#define DBKO_U_PU_LEFT      0x0010u /* the unit's left border is a prediction block edge */
#define DBKO_U_PU_TOP       0x0020u
#define DBKO_U_KEEP         0x0040u /* samples stay unmodified: PCM with pcm_loop_filter_disabled_flag, or transquant bypass */
#define DBKO_U_DBK_OFF      0x0080u /* slice_deblocking_filter_disabled_flag of the slice holding the unit */
#define DBKO_U_PRED_L0      0x0100u /* uses reference list 0 (mv0 / ref0 valid) */
#define DBKO_U_PRED_L1      0x0200u
#define DBKO_U_NOX_LEFT     0x0400u /* left border is a slice / tile boundary that in-loop filters must not cross */
#define DBKO_U_NOX_TOP      0x0800u

typedef struct {
    const uint16_t *flags; /* (W/4) * (H/4), row-major */
    const int16_t *mv0;    /* [unit][2]: x, y in quarter luma samples */
    const int16_t *mv1;
    const int32_t *ref0;   /* identity of the reference PICTURE (e.g. its POC), not the index in the list */
    const int32_t *ref1;
} dbko_h265_units;

/* luma arrays of dbko_h265_num_vert_bs(W,H) / _hor_bs(W,H) entries */
int dbko_h265_derive_bs(const dbko_h265_units *u, unsigned w, unsigned h, uint8_t *vert_bs4, uint8_t *hor_bs4);

/* 4:2:0 chroma arrays (geometry of the (w/2) x (h/2) plane) from the luma arrays: the bS of the luma segment at
 * twice the chroma position (8.7.2.5: bS[xDk*SubWidthC][yDm*SubHeightC]) */
void dbko_h265_chroma_bs(const uint8_t *vert_bs4, const uint8_t *hor_bs4, unsigned w, unsigned h,
                         uint8_t *c_vert_bs4, uint8_t *c_hor_bs4);


/* ---- sample adaptive offset, clause 8.7.3 (PARITY UNPINNED like the rest of this file) ---- */

typedef struct {
    uint8_t type;      /* SaoTypeIdx: 0 = off, 1 = band offset, 2 = edge offset */
    uint8_t cls;       /* band offset: sao_band_position (0..31); edge offset: SaoEoClass (0 hor, 1 ver, 2 135 deg, 3 45 deg) */
    int8_t offset[4];  /* SaoOffsetVal[1..4], already scaled (<< log2OffsetScale) and signed */
} dbko_sao_ctb;

/*
 * SAO of one plane, src -> dst (the edge classifier reads DEBLOCKED neighbours, never offset ones).  `ctb_log2` is the CTB
 * size of THIS plane in samples (luma 4..6, 4:2:0 chroma one less); params[(y >> ctb_log2) * params_stride + (x >> ctb_log2)].
 * keep (may be NULL): one byte per 8x8 block of this plane's samples, row stride keep_stride; non-zero = the sample stays
 * unmodified (pcm_loop_filter_disabled_flag with PCM, cu_transquant_bypass_flag).  Edge offset leaves a sample alone when
 * one of its two neighbours lies outside the picture.
 */
int dbko_h265_sao_plane(const void *src, void *dst, unsigned plane_w, unsigned plane_h, size_t pitch_bytes, unsigned bit_depth,
                        unsigned sample_bytes, const dbko_sao_ctb *params, unsigned params_stride, unsigned ctb_log2,
                        const uint8_t *keep, unsigned keep_stride);

#ifdef __cplusplus
}
#endif
#endif
