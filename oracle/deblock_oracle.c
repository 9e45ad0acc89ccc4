/*
 * deblock_oracle.c -- CPU restatement of the reference's deblocking filter (plain C).
 * TEST INFRASTRUCTURE ONLY: see the header of deblock_oracle.h for who may call this and for
 * the parity status (8-bit scalar-QP pinned against oracle/_ref and tests/golden; 10-bit and
 * QP-map "parity unpinned").
 *
 * Citations "cpu.h:N" are to
 * /root/reference/hevc_deblocking_filter/hevc_deblocking_filter_cpu.h.
 */
#include "deblock_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* cpu.h:1021-1026 */
const unsigned dbko_beta_table[52] = {
    0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,
    6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24,
    26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56,
    58, 60, 62, 64};
/* cpu.h:1028-1033 */
const unsigned dbko_tc_table[52] = {
    0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,
    0,  0,  1,  1,  1,  1,  1,  1,  1,  1,  1,  2,  2,  2,  2,  3,
    3,  3,  3,  4,  4,  4,  5,  5,  6,  6,  7,  8,  9,  10, 11, 13,
    14, 16, 18, 20};

/* cpu.h:86 / 104: (W/8 + 1) * H / 8 */
size_t dbko_num_vert_bs(unsigned w, unsigned h) { return (size_t)(w / 8 + 1) * h / 8; }
/* cpu.h:87 / 105: (H/8 + 1) * W / 8 */
size_t dbko_num_hor_bs(unsigned w, unsigned h) { return (size_t)(h / 8 + 1) * w / 8; }

/* cpu.h:92-99 and 110-117 */
void dbko_default_bs(unsigned w, unsigned h, uint8_t *vert_bs, uint8_t *hor_bs)
{
    size_t nv = dbko_num_vert_bs(w, h), nh = dbko_num_hor_bs(w, h);
    for (size_t i = 0; i < nv; i++) vert_bs[i] = (i % (w / 8 + 1) == 0) ? 0 : 2;
    /* the zeroing stride is H/8+1 although the array is consumed with row stride W/8 */
    for (size_t i = 0; i < nh; i++) hor_bs[i] = (i % (h / 8 + 1) == 0) ? 0 : 2;
}

/* cpu.h:1117-1120 */
static inline int clip1(int v, int c) { return v < -c ? -c : (v > c ? c : v); }
/* cpu.h:1123-1126 */
static inline int clip2(int v, int c) { return v < 0 ? 0 : (v > c ? c : v); }

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/*
 * tc / beta of one segment.  Scalar QP: cpu.h:136-137 + 1064-1072 (index clamped to 51, no
 * bS offset, no chroma QP mapping -- SURVEY Q6/Q8).  bit_depth > 8 scales both by
 * 1 << (bit_depth - 8) as H.265 8.7.2.5.3 does (extension, parity unpinned).
 * QP map (extension, parity unpinned): QP = (QpP + QpQ + 1) >> 1 where QpP / QpQ are the map
 * entries of the CTUs holding P_0 / Q_0 of line 0, coordinates clamped into the image;
 * (xp,yp)/(xq,yq) are plane coordinates (chroma planes are scaled x2 to luma coordinates).
 */
static void seg_tc_beta(const dbko_qp *qp, const unsigned *tc_tab, const unsigned *beta_tab,
                        int shift, int is_chroma, unsigned plane_w, unsigned plane_h,
                        int xp, int yp, int xq, int yq, int *tc, int *beta)
{
    unsigned q = qp->qp;
    if (qp->map) {
        int sc = is_chroma ? 2 : 1;
        int lw = (int)plane_w * sc, lh = (int)plane_h * sc;
        int lxp = clampi(xp * sc, 0, lw - 1), lyp = clampi(yp * sc, 0, lh - 1);
        int lxq = clampi(xq * sc, 0, lw - 1), lyq = clampi(yq * sc, 0, lh - 1);
        unsigned qpp = qp->map[(size_t)(lyp >> qp->ctu_log2) * qp->map_stride + (lxp >> qp->ctu_log2)];
        unsigned qpq = qp->map[(size_t)(lyq >> qp->ctu_log2) * qp->map_stride + (lxq >> qp->ctu_log2)];
        q = (qpp + qpq + 1) >> 1;
    }
    if (q > 51) q = 51;
    *tc = (int)(tc_tab[q] << shift);
    *beta = (int)(beta_tab[q] << shift);
}

#define SAMPLE_T uint8_t
#define SFX 8
#include "deblock_oracle_impl.inc"
#undef SAMPLE_T
#undef SFX

#define SAMPLE_T uint16_t
#define SFX 16
#include "deblock_oracle_impl.inc"
#undef SAMPLE_T
#undef SFX

/* ---------------------------------------------------------------------------------------- */
/* frame container (class ReadYuvFrame, cpu.h:33-132, 995-1018, 1035-1062)                   */

struct dbko_frame {
    unsigned width, height, bit_depth, bps; /* bps = bytes per sample */
    int has_chroma;
    void *ext[3];               /* padded planes, (w+8)*(h+8) samples, zero padding */
    uint8_t *vert_bs, *hor_bs;  /* luma */
    uint8_t *cvert_bs, *chor_bs;
    size_t n_vert, n_hor, n_cvert, n_chor;
};

static unsigned plane_w(const dbko_frame *f, int pl) { return pl ? f->width / 2 : f->width; }
static unsigned plane_h(const dbko_frame *f, int pl) { return pl ? f->height / 2 : f->height; }

static void copy_in(dbko_frame *f, int pl, const void *src, size_t pitch)
{
    unsigned w = plane_w(f, pl), h = plane_h(f, pl);
    size_t pw = (size_t)w + 8;
    for (unsigned r = 0; r < h; r++) /* cpu.h:68-71: image at (4,4) */
        memcpy((uint8_t *)f->ext[pl] + ((size_t)(r + 4) * pw + 4) * f->bps,
               (const uint8_t *)src + (size_t)r * pitch, (size_t)w * f->bps);
}

static void copy_out(const dbko_frame *f, int pl, void *dst, size_t pitch)
{
    unsigned w = plane_w(f, pl), h = plane_h(f, pl);
    size_t pw = (size_t)w + 8;
    for (unsigned r = 0; r < h; r++) /* cpu.h:1000-1004 */
        memcpy((uint8_t *)dst + (size_t)r * pitch,
               (const uint8_t *)f->ext[pl] + ((size_t)(r + 4) * pw + 4) * f->bps, (size_t)w * f->bps);
}

static int bad_depth(unsigned bit_depth, unsigned sample_bytes)
{
    return bit_depth < 8 || bit_depth > 16 || (sample_bytes != 1 && sample_bytes != 2) ||
           (sample_bytes == 1 && bit_depth != 8);
}

int dbko_frame_create(dbko_frame **out, unsigned width, unsigned height, unsigned bit_depth,
                      unsigned sample_bytes,
                      const void *y, size_t y_pitch, const void *u, size_t u_pitch,
                      const void *v, size_t v_pitch)
{
    if (!out || !y || bad_depth(bit_depth, sample_bytes)) return DBKO_ERR_ARG;
    if (width == 0 || height == 0 || width % 8 != 0 || height % 8 != 0) return DBKO_ERR_DIMENSIONS; /* cpu.h:46-48 */
    int has_chroma = (u != NULL && v != NULL);
    if (has_chroma && ((width / 2) % 8 != 0 || (height / 2) % 8 != 0)) return DBKO_ERR_DIMENSIONS;
    dbko_frame *f = (dbko_frame *)calloc(1, sizeof(*f));
    if (!f) return DBKO_ERR_NOMEM;
    f->width = width; f->height = height; f->bit_depth = bit_depth;
    f->bps = sample_bytes;
    f->has_chroma = has_chroma;
    for (int pl = 0; pl < (has_chroma ? 3 : 1); pl++) {
        /* calloc: padding == 0 (the reference's new[] returns zero pages in a fresh process) */
        f->ext[pl] = calloc((size_t)(plane_w(f, pl) + 8) * (plane_h(f, pl) + 8), f->bps);
        if (!f->ext[pl]) { dbko_frame_destroy(f); return DBKO_ERR_NOMEM; }
    }
    copy_in(f, 0, y, y_pitch);
    if (has_chroma) { copy_in(f, 1, u, u_pitch); copy_in(f, 2, v, v_pitch); }

    f->n_vert = dbko_num_vert_bs(width, height);
    f->n_hor = dbko_num_hor_bs(width, height);
    f->vert_bs = (uint8_t *)malloc(f->n_vert ? f->n_vert : 1);
    f->hor_bs = (uint8_t *)malloc(f->n_hor ? f->n_hor : 1);
    if (!f->vert_bs || !f->hor_bs) { dbko_frame_destroy(f); return DBKO_ERR_NOMEM; }
    dbko_default_bs(width, height, f->vert_bs, f->hor_bs);
    if (has_chroma) {
        f->n_cvert = dbko_num_vert_bs(width / 2, height / 2);
        f->n_chor = dbko_num_hor_bs(width / 2, height / 2);
        f->cvert_bs = (uint8_t *)malloc(f->n_cvert ? f->n_cvert : 1);
        f->chor_bs = (uint8_t *)malloc(f->n_chor ? f->n_chor : 1);
        if (!f->cvert_bs || !f->chor_bs) { dbko_frame_destroy(f); return DBKO_ERR_NOMEM; }
        dbko_default_bs(width / 2, height / 2, f->cvert_bs, f->chor_bs);
    }
    *out = f;
    return DBKO_OK;
}

int dbko_frame_create_from_file(dbko_frame **out, const char *path, unsigned width, unsigned height)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return DBKO_ERR_FILE_SIZE;
    fseek(fp, 0, SEEK_END);
    long length = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    /* same order as the reference: size check first (cpu.h:43-45), then divisibility (46-48) */
    if ((unsigned long)length != 3ul * width * height / 2) { fclose(fp); return DBKO_ERR_FILE_SIZE; }
    if (width % 8 != 0 || height % 8 != 0) { fclose(fp); return DBKO_ERR_DIMENSIONS; }
    uint8_t *buf = (uint8_t *)malloc((size_t)length ? (size_t)length : 1);
    if (!buf) { fclose(fp); return DBKO_ERR_NOMEM; }
    size_t got = fread(buf, 1, (size_t)length, fp);
    fclose(fp);
    if (got != (size_t)length) { free(buf); return DBKO_ERR_FILE_SIZE; }
    size_t ysz = (size_t)width * height, csz = ysz / 4;
    int rc = dbko_frame_create(out, width, height, 8, 1, buf, width, buf + ysz, width / 2,
                               buf + ysz + csz, width / 2);
    free(buf);
    return rc;
}

void dbko_frame_destroy(dbko_frame *f)
{
    if (!f) return;
    for (int pl = 0; pl < 3; pl++) free(f->ext[pl]);
    free(f->vert_bs); free(f->hor_bs); free(f->cvert_bs); free(f->chor_bs);
    free(f);
}

int dbko_frame_set_boundary_strength(dbko_frame *f, const uint8_t *vert_bs, size_t n_vert,
                                     const uint8_t *hor_bs, size_t n_hor)
{
    if (!f || !vert_bs || !hor_bs) return DBKO_ERR_ARG;
    if (f->n_hor != n_hor || f->n_vert != n_vert) return DBKO_ERR_BS_SIZE; /* cpu.h:122-123 */
    memcpy(f->vert_bs, vert_bs, n_vert);
    memcpy(f->hor_bs, hor_bs, n_hor);
    return DBKO_OK;
}

int dbko_frame_set_chroma_boundary_strength(dbko_frame *f, const uint8_t *vert_bs, size_t n_vert,
                                            const uint8_t *hor_bs, size_t n_hor)
{
    if (!f || !vert_bs || !hor_bs || !f->has_chroma) return DBKO_ERR_ARG;
    if (f->n_chor != n_hor || f->n_cvert != n_vert) return DBKO_ERR_BS_SIZE;
    memcpy(f->cvert_bs, vert_bs, n_vert);
    memcpy(f->chor_bs, hor_bs, n_hor);
    return DBKO_OK;
}

static void filter_one(dbko_frame *f, int pl, const uint8_t *vbs, const uint8_t *hbs,
                       const dbko_qp *qp, const unsigned *tc_tab, const unsigned *beta_tab,
                       unsigned num_threads)
{
    if (f->bps == 1)
        filter_plane_8((uint8_t *)f->ext[pl], plane_w(f, pl), plane_h(f, pl), pl != 0, vbs, hbs,
                       qp, tc_tab, beta_tab, f->bit_depth, num_threads);
    else
        filter_plane_16((uint16_t *)f->ext[pl], plane_w(f, pl), plane_h(f, pl), pl != 0, vbs, hbs,
                        qp, tc_tab, beta_tab, f->bit_depth, num_threads);
}

int dbko_frame_filter(dbko_frame *f, const dbko_qp *qp, const dbko_tables *tables,
                      unsigned planes, unsigned num_threads)
{
    if (!f || !qp) return DBKO_ERR_ARG;
    const unsigned *tc_tab = (tables && tables->tc) ? tables->tc : dbko_tc_table;
    const unsigned *beta_tab = (tables && tables->beta) ? tables->beta : dbko_beta_table;
    /* cpu.h:144 luma, :452 U, :723 V -- in this order */
    if (planes & DBKO_PLANE_Y) filter_one(f, 0, f->vert_bs, f->hor_bs, qp, tc_tab, beta_tab, num_threads);
    if (f->has_chroma) {
        if (planes & DBKO_PLANE_U) filter_one(f, 1, f->cvert_bs, f->chor_bs, qp, tc_tab, beta_tab, num_threads);
        if (planes & DBKO_PLANE_V) filter_one(f, 2, f->cvert_bs, f->chor_bs, qp, tc_tab, beta_tab, num_threads);
    }
    return DBKO_OK;
}

int dbko_frame_save(const dbko_frame *f, void *y, size_t y_pitch, void *u, size_t u_pitch,
                    void *v, size_t v_pitch)
{
    if (!f) return DBKO_ERR_ARG;
    if (y) copy_out(f, 0, y, y_pitch);
    if (f->has_chroma) {
        if (u) copy_out(f, 1, u, u_pitch);
        if (v) copy_out(f, 2, v, v_pitch);
    }
    return DBKO_OK;
}

int dbko_frame_save_to_file(const dbko_frame *f, const char *path)
{
    if (!f || f->bps != 1 || !f->has_chroma) return DBKO_ERR_ARG;
    size_t ysz = (size_t)f->width * f->height, csz = ysz / 4;
    uint8_t *buf = (uint8_t *)malloc(ysz + 2 * csz);
    if (!buf) return DBKO_ERR_NOMEM;
    dbko_frame_save(f, buf, f->width, buf + ysz, f->width / 2, buf + ysz + csz, f->width / 2);
    FILE *fp = fopen(path, "wb");
    if (!fp) { free(buf); return DBKO_ERR_ARG; }
    fwrite(buf, 1, ysz + 2 * csz, fp); /* Y, U, V order: cpu.h:1000-1016 */
    fclose(fp);
    free(buf);
    return DBKO_OK;
}

int dbko_filter_plane(void *plane, unsigned pw_, unsigned ph_, size_t pitch, unsigned bit_depth,
                      unsigned sample_bytes, int is_chroma, const uint8_t *vert_bs, const uint8_t *hor_bs,
                      const dbko_qp *qp, const dbko_tables *tables, unsigned num_threads)
{
    if (!plane || !qp || bad_depth(bit_depth, sample_bytes)) return DBKO_ERR_ARG;
    if (pw_ == 0 || ph_ == 0 || pw_ % 8 != 0 || ph_ % 8 != 0) return DBKO_ERR_DIMENSIONS;
    const unsigned *tc_tab = (tables && tables->tc) ? tables->tc : dbko_tc_table;
    const unsigned *beta_tab = (tables && tables->beta) ? tables->beta : dbko_beta_table;
    unsigned bps = sample_bytes;
    size_t ew = (size_t)pw_ + 8, eh = (size_t)ph_ + 8;
    void *ext = calloc(ew * eh, bps);
    size_t nv = dbko_num_vert_bs(pw_, ph_), nh = dbko_num_hor_bs(pw_, ph_);
    uint8_t *dv = (uint8_t *)malloc(nv ? nv : 1), *dh = (uint8_t *)malloc(nh ? nh : 1);
    if (!ext || !dv || !dh) { free(ext); free(dv); free(dh); return DBKO_ERR_NOMEM; }
    dbko_default_bs(pw_, ph_, dv, dh);
    if (vert_bs) memcpy(dv, vert_bs, nv);
    if (hor_bs) memcpy(dh, hor_bs, nh);
    for (unsigned r = 0; r < ph_; r++)
        memcpy((uint8_t *)ext + ((size_t)(r + 4) * ew + 4) * bps, (uint8_t *)plane + (size_t)r * pitch,
               (size_t)pw_ * bps);
    if (bps == 1)
        filter_plane_8((uint8_t *)ext, pw_, ph_, is_chroma, dv, dh, qp, tc_tab, beta_tab, bit_depth, num_threads);
    else
        filter_plane_16((uint16_t *)ext, pw_, ph_, is_chroma, dv, dh, qp, tc_tab, beta_tab, bit_depth, num_threads);
    for (unsigned r = 0; r < ph_; r++)
        memcpy((uint8_t *)plane + (size_t)r * pitch, (uint8_t *)ext + ((size_t)(r + 4) * ew + 4) * bps,
               (size_t)pw_ * bps);
    free(ext); free(dv); free(dh);
    return DBKO_OK;
}
