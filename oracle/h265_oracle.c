/*
 * h265_oracle.c -- see h265_oracle.h.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (restates H.265 clause 8.7.2).
 */
#include "h265_oracle.h"

#include <stdlib.h>
#include <string.h>

/* Table 8-12 */
const uint8_t dbko_h265_beta_table[52] = {
    0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15,
    16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};
const uint8_t dbko_h265_tc_table[54] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1,  1,  1,  1,  1,  1,  1,
    2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};

static int iabs(int x) { return x < 0 ? -x : x; }
static int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

size_t dbko_h265_num_vert_bs(unsigned w, unsigned h) { return (size_t)(w / 8 + 1) * (h / 4); }
size_t dbko_h265_num_hor_bs(unsigned w, unsigned h) { return (size_t)(h / 8 + 1) * (w / 4); }

/* Table 8-10, ChromaArrayType == 1 */
static int chroma_qp(int qpi)
{
    static const int8_t t[14] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37};
    if (qpi < 30) return qpi;
    if (qpi > 43) return qpi - 6;
    return t[qpi - 30];
}

typedef struct {
    int *s;          /* working samples, plane_w x plane_h */
    int w, h;
    int bit_depth;
    int c_idx;
    unsigned qp;
    const uint8_t *qp_map;
    unsigned map_stride, unit_log2;
    dbko_h265_params prm;
} plane_t;

/* QpY of the coding unit covering plane sample (x, y) */
static int qp_at(const plane_t *p, int x, int y)
{
    if (!p->qp_map) return (int)p->qp;
    const int sc = p->c_idx ? 2 : 1;
    return p->qp_map[((unsigned)(y * sc) >> p->unit_log2) * p->map_stride + ((unsigned)(x * sc) >> p->unit_log2)];
}

/*
 * One 4-line segment of an edge.  P0 / Q0 of line 0 sit at (xp, yp) / (xq, yq); the next line is (lx, ly) further;
 * tap k of P is k steps (-tx, -ty) away from P0, tap k of Q is k steps (+tx, +ty) away from Q0.
 */
static void filter_segment(plane_t *p, unsigned b, int xq, int yq, int tx, int ty, int lx, int ly)
{
    const int bs = (int)(b & DBKO_H265_BS_MASK);
    const int xp = xq - tx, yp = yq - ty;
    const int shift = p->bit_depth - 8, max_v = (1 << p->bit_depth) - 1;
    const int qpl = (qp_at(p, xq, yq) + qp_at(p, xp, yp) + 1) >> 1; /* 8.7.2.5.3: QpQ, QpP of the CUs of q0,0 / p0,0 */
    int *s = p->s;
    const int w = p->w;
#define P_(i, k) s[(yp + (i) * ly - (k) * ty) * w + (xp + (i) * lx - (k) * tx)]
#define Q_(i, k) s[(yq + (i) * ly + (k) * ty) * w + (xq + (i) * lx + (k) * tx)]
    if (p->c_idx == 0) {
        if (bs == 0) return;
        const int beta = dbko_h265_beta_table[clip3(0, 51, qpl + (p->prm.beta_offset_div2 << 1))] << shift;
        const int tc = dbko_h265_tc_table[clip3(0, 53, qpl + 2 * (bs - 1) + (p->prm.tc_offset_div2 << 1))] << shift;
        const int dp0 = iabs(P_(0, 2) - 2 * P_(0, 1) + P_(0, 0)), dp3 = iabs(P_(3, 2) - 2 * P_(3, 1) + P_(3, 0));
        const int dq0 = iabs(Q_(0, 2) - 2 * Q_(0, 1) + Q_(0, 0)), dq3 = iabs(Q_(3, 2) - 2 * Q_(3, 1) + Q_(3, 0));
        const int dpq0 = dp0 + dq0, dpq3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3, d = dpq0 + dpq3;
        if (!(d < beta)) return; /* dE = 0 */
        /* 8.7.2.5.6 on lines 0 and 3, dpq = 2 * dpqN */
        const int dsam0 = (2 * dpq0 < (beta >> 2)) && (iabs(P_(0, 3) - P_(0, 0)) + iabs(Q_(0, 0) - Q_(0, 3)) < (beta >> 3)) &&
                          (iabs(P_(0, 0) - Q_(0, 0)) < ((5 * tc + 1) >> 1));
        const int dsam3 = (2 * dpq3 < (beta >> 2)) && (iabs(P_(3, 3) - P_(3, 0)) + iabs(Q_(3, 0) - Q_(3, 3)) < (beta >> 3)) &&
                          (iabs(P_(3, 0) - Q_(3, 0)) < ((5 * tc + 1) >> 1));
        const int de = (dsam0 && dsam3) ? 2 : 1;
        const int dep = dp < ((beta + (beta >> 1)) >> 3), deq = dq < ((beta + (beta >> 1)) >> 3);
        const int keep_p = (b & DBKO_H265_KEEP_P) != 0, keep_q = (b & DBKO_H265_KEEP_Q) != 0;
        for (int i = 0; i < 4; i++) {
            const int p0 = P_(i, 0), p1 = P_(i, 1), p2 = P_(i, 2), p3 = P_(i, 3);
            const int q0 = Q_(i, 0), q1 = Q_(i, 1), q2 = Q_(i, 2), q3 = Q_(i, 3);
            int np[3] = {p0, p1, p2}, nq[3] = {q0, q1, q2};
            if (de == 2) { /* 8.7.2.5.7, strong */
                np[0] = clip3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
                np[1] = clip3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0 + 2) >> 2);
                np[2] = clip3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
                nq[0] = clip3(q0 - 2 * tc, q0 + 2 * tc, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
                nq[1] = clip3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0 + q1 + q2 + 2) >> 2);
                nq[2] = clip3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3);
            } else {
                int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
                if (iabs(delta) < tc * 10) {
                    delta = clip3(-tc, tc, delta);
                    np[0] = clip3(0, max_v, p0 + delta);
                    nq[0] = clip3(0, max_v, q0 - delta);
                    if (dep) np[1] = clip3(0, max_v, p1 + clip3(-(tc >> 1), tc >> 1, (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1));
                    if (deq) nq[1] = clip3(0, max_v, q1 + clip3(-(tc >> 1), tc >> 1, (((q2 + q0 + 1) >> 1) - q1 - delta) >> 1));
                }
            }
            if (!keep_p) { P_(i, 0) = np[0]; P_(i, 1) = np[1]; P_(i, 2) = np[2]; }
            if (!keep_q) { Q_(i, 0) = nq[0]; Q_(i, 1) = nq[1]; Q_(i, 2) = nq[2]; }
        }
    } else {
        if (bs != 2) return; /* 8.7.2.5: chroma edges only where bS == 2 */
        const int qpc = chroma_qp(qpl + p->prm.c_qp_offset); /* 8.7.2.5.5 */
        const int tc = dbko_h265_tc_table[clip3(0, 53, qpc + 2 * (bs - 1) + (p->prm.tc_offset_div2 << 1))] << shift;
        for (int i = 0; i < 4; i++) {
            const int p0 = P_(i, 0), p1 = P_(i, 1), q0 = Q_(i, 0), q1 = Q_(i, 1);
            const int delta = clip3(-tc, tc, ((((q0 - p0) << 2) + p1 - q1 + 4) >> 3)); /* 8.7.2.5.8 */
            if (!(b & DBKO_H265_KEEP_P)) P_(i, 0) = clip3(0, max_v, p0 + delta);
            if (!(b & DBKO_H265_KEEP_Q)) Q_(i, 0) = clip3(0, max_v, q0 - delta);
        }
    }
#undef P_
#undef Q_
}

int dbko_h265_filter_plane(void *plane, unsigned plane_w, unsigned plane_h, size_t pitch_bytes, unsigned bit_depth,
                           unsigned sample_bytes, int c_idx, const uint8_t *vert_bs4, const uint8_t *hor_bs4,
                           unsigned qp, const uint8_t *qp_map, unsigned map_stride, unsigned unit_log2,
                           const dbko_h265_params *prm)
{
    if (!plane || !vert_bs4 || !hor_bs4 || plane_w == 0 || plane_h == 0 || plane_w % 8 || plane_h % 8) return -2;
    if (bit_depth < 8 || bit_depth > 16 || (sample_bytes != 1 && sample_bytes != 2) || (sample_bytes == 1 && bit_depth != 8)) return -5;
    plane_t p;
    memset(&p, 0, sizeof(p));
    p.w = (int)plane_w; p.h = (int)plane_h; p.bit_depth = (int)bit_depth; p.c_idx = c_idx;
    p.qp = qp; p.qp_map = qp_map; p.map_stride = map_stride; p.unit_log2 = unit_log2;
    if (prm) p.prm = *prm;
    p.s = (int *)malloc(sizeof(int) * plane_w * plane_h);
    if (!p.s) return -6;
    for (unsigned y = 0; y < plane_h; y++)
        for (unsigned x = 0; x < plane_w; x++)
            p.s[y * plane_w + x] = sample_bytes == 1 ? ((const uint8_t *)plane)[y * pitch_bytes + x]
                                                      : ((const uint16_t *)((const uint8_t *)plane + y * pitch_bytes))[x];
    const int vstride = (int)(plane_w / 8 + 1), hstride = (int)(plane_w / 4);
    /* 8.7.2.1: the vertical edges of the whole picture first ... */
    for (int y4 = 0; y4 < (int)plane_h / 4; y4++)
        for (int bx = 1; bx < (int)plane_w / 8; bx++)
            filter_segment(&p, vert_bs4[y4 * vstride + bx], 8 * bx, 4 * y4, 1, 0, 0, 1);
    /* ... then the horizontal edges, with the samples modified by the vertical edge filtering as input */
    for (int by = 1; by < (int)plane_h / 8; by++)
        for (int x4 = 0; x4 < (int)plane_w / 4; x4++)
            filter_segment(&p, hor_bs4[by * hstride + x4], 4 * x4, 8 * by, 0, 1, 1, 0);
    for (unsigned y = 0; y < plane_h; y++)
        for (unsigned x = 0; x < plane_w; x++) {
            if (sample_bytes == 1) ((uint8_t *)plane)[y * pitch_bytes + x] = (uint8_t)p.s[y * plane_w + x];
            else ((uint16_t *)((uint8_t *)plane + y * pitch_bytes))[x] = (uint16_t)p.s[y * plane_w + x];
        }
    free(p.s);
    return 0;
}

/* ---- 8.7.2.4 ---- */

typedef struct {
    int n;          /* number of motion vectors */
    int32_t ref[2]; /* reference pictures */
    int mv[2][2];
} motion_t;

static motion_t motion_of(const dbko_h265_units *u, size_t i)
{
    motion_t m;
    memset(&m, 0, sizeof(m));
    const unsigned f = u->flags[i];
    if (f & DBKO_U_PRED_L0) { m.ref[m.n] = u->ref0[i]; m.mv[m.n][0] = u->mv0[2 * i]; m.mv[m.n][1] = u->mv0[2 * i + 1]; m.n++; }
    if (f & DBKO_U_PRED_L1) { m.ref[m.n] = u->ref1[i]; m.mv[m.n][0] = u->mv1[2 * i]; m.mv[m.n][1] = u->mv1[2 * i + 1]; m.n++; }
    return m;
}

/* "the absolute difference between the horizontal or vertical component ... is greater than or equal to 4" */
static int mv_far(const int a[2], const int b[2]) { return iabs(a[0] - b[0]) >= 4 || iabs(a[1] - b[1]) >= 4; }

static unsigned bs_of_edge(const dbko_h265_units *u, size_t ip, size_t iq, int left)
{
    const unsigned fp = u->flags[ip], fq = u->flags[iq];
    const unsigned tu = left ? DBKO_U_TU_LEFT : DBKO_U_TU_TOP, pu = left ? DBKO_U_PU_LEFT : DBKO_U_PU_TOP;
    const unsigned nox = left ? DBKO_U_NOX_LEFT : DBKO_U_NOX_TOP;
    if (!(fq & (tu | pu))) return 0;               /* neither a transform nor a prediction block edge */
    if (fq & (DBKO_U_DBK_OFF | nox)) return 0;     /* filterEdgeFlag = 0 */
    unsigned bs;
    if ((fp | fq) & DBKO_U_INTRA) {
        bs = 2;
    } else if ((fq & tu) && ((fp | fq) & DBKO_U_CBF)) {
        bs = 1;
    } else {
        const motion_t a = motion_of(u, ip), b = motion_of(u, iq);
        if (a.n != b.n) {
            bs = 1;
        } else if (a.n == 1) {
            bs = (a.ref[0] != b.ref[0]) || mv_far(a.mv[0], b.mv[0]);
        } else if (a.n == 2) {
            const int same_set = (a.ref[0] == b.ref[0] && a.ref[1] == b.ref[1]) || (a.ref[0] == b.ref[1] && a.ref[1] == b.ref[0]);
            if (!same_set) {
                bs = 1;
            } else if (a.ref[0] != a.ref[1]) {
                /* two different reference pictures: compare the vectors that refer to the same picture */
                if (a.ref[0] == b.ref[0]) bs = mv_far(a.mv[0], b.mv[0]) || mv_far(a.mv[1], b.mv[1]);
                else bs = mv_far(a.mv[0], b.mv[1]) || mv_far(a.mv[1], b.mv[0]);
            } else {
                /* both vectors of both blocks refer to the same picture */
                bs = (mv_far(a.mv[0], b.mv[0]) || mv_far(a.mv[1], b.mv[1])) &&
                     (mv_far(a.mv[0], b.mv[1]) || mv_far(a.mv[1], b.mv[0]));
            }
        } else {
            bs = 0; /* no motion on either side and not intra: nothing to compare */
        }
    }
    if (bs == 0) return 0;
    return bs | ((fp & DBKO_U_KEEP) ? DBKO_H265_KEEP_P : 0u) | ((fq & DBKO_U_KEEP) ? DBKO_H265_KEEP_Q : 0u);
}

int dbko_h265_derive_bs(const dbko_h265_units *u, unsigned w, unsigned h, uint8_t *vert_bs4, uint8_t *hor_bs4)
{
    if (!u || !vert_bs4 || !hor_bs4 || w % 8 || h % 8 || !w || !h) return -2;
    const size_t uw = w / 4, uh = h / 4;
    const size_t vstride = w / 8 + 1, hstride = w / 4;
    memset(vert_bs4, 0, dbko_h265_num_vert_bs(w, h));
    memset(hor_bs4, 0, dbko_h265_num_hor_bs(w, h));
    for (size_t y4 = 0; y4 < uh; y4++)
        for (size_t bx = 1; bx < w / 8; bx++) /* x = 0 and x = w are the picture boundary */
            vert_bs4[y4 * vstride + bx] = (uint8_t)bs_of_edge(u, y4 * uw + 2 * bx - 1, y4 * uw + 2 * bx, 1);
    for (size_t by = 1; by < h / 8; by++)
        for (size_t x4 = 0; x4 < uw; x4++)
            hor_bs4[by * hstride + x4] = (uint8_t)bs_of_edge(u, (2 * by - 1) * uw + x4, (2 * by) * uw + x4, 0);
    return 0;
}

void dbko_h265_chroma_bs(const uint8_t *vert_bs4, const uint8_t *hor_bs4, unsigned w, unsigned h,
                         uint8_t *c_vert_bs4, uint8_t *c_hor_bs4)
{
    const unsigned cw = w / 2, ch = h / 2;
    const size_t vstride = w / 8 + 1, hstride = w / 4, cvstride = cw / 8 + 1, chstride = cw / 4;
    for (unsigned m = 0; m < ch / 4; m++)
        for (unsigned bx = 0; bx <= cw / 8; bx++)
            c_vert_bs4[m * cvstride + bx] = vert_bs4[(size_t)(2 * m) * vstride + 2 * bx];
    for (unsigned by = 0; by <= ch / 8; by++)
        for (unsigned m = 0; m < cw / 4; m++)
            c_hor_bs4[by * chstride + m] = hor_bs4[(size_t)(2 * by) * hstride + 2 * m];
}

/* ---- 8.7.3 ---- */

static int sgn(int v) { return (v > 0) - (v < 0); }

int dbko_h265_sao_plane(const void *src, void *dst, unsigned plane_w, unsigned plane_h, size_t pitch_bytes, unsigned bit_depth,
                        unsigned sample_bytes, const dbko_sao_ctb *params, unsigned params_stride, unsigned ctb_log2,
                        const uint8_t *keep, unsigned keep_stride)
{
    if (!src || !dst || !params || src == dst || plane_w == 0 || plane_h == 0) return -5;
    if (bit_depth < 8 || bit_depth > 16 || (sample_bytes != 1 && sample_bytes != 2) || (sample_bytes == 1 && bit_depth != 8)) return -5;
    static const int hpos[4][2] = {{-1, 1}, {0, 0}, {-1, 1}, {1, -1}};   /* 8.7.3.2, Table 8-13 */
    static const int vpos[4][2] = {{0, 0}, {-1, 1}, {-1, 1}, {-1, 1}};
    const int max_v = (1 << bit_depth) - 1, band_shift = (int)bit_depth - 5;
#define S_(x, y) (sample_bytes == 1 ? (int)((const uint8_t *)src)[(size_t)(y) * pitch_bytes + (x)] \
                                    : (int)((const uint16_t *)((const uint8_t *)src + (size_t)(y) * pitch_bytes))[(x)])
    for (int y = 0; y < (int)plane_h; y++)
        for (int x = 0; x < (int)plane_w; x++) {
            const dbko_sao_ctb *c = &params[((unsigned)y >> ctb_log2) * params_stride + ((unsigned)x >> ctb_log2)];
            const int rec = S_(x, y);
            int out = rec;
            const int kept = keep && keep[((unsigned)y >> 3) * keep_stride + ((unsigned)x >> 3)];
            if (!kept && c->type == 1) {
                /* bandTable[(k + sao_band_position) & 31] = k + 1 for k = 0..3 */
                const int k = ((rec >> band_shift) - (int)c->cls) & 31;
                if (k < 4) out = clip3(0, max_v, rec + c->offset[k]);
            } else if (!kept && c->type == 2) {
                const int xa = x + hpos[c->cls & 3][0], ya = y + vpos[c->cls & 3][0];
                const int xb = x + hpos[c->cls & 3][1], yb = y + vpos[c->cls & 3][1];
                if (xa >= 0 && xa < (int)plane_w && ya >= 0 && ya < (int)plane_h && xb >= 0 && xb < (int)plane_w && yb >= 0 &&
                    yb < (int)plane_h) {
                    int e = 2 + sgn(rec - S_(xa, ya)) + sgn(rec - S_(xb, yb));
                    if (e <= 2) e = (e == 2) ? 0 : e + 1; /* 0 -> 1, 1 -> 2, 2 -> 0 */
                    if (e) out = clip3(0, max_v, rec + c->offset[e - 1]);
                }
            }
            if (sample_bytes == 1) ((uint8_t *)dst)[(size_t)y * pitch_bytes + x] = (uint8_t)out;
            else ((uint16_t *)((uint8_t *)dst + (size_t)y * pitch_bytes))[x] = (uint16_t)out;
        }
#undef S_
    return 0;
}
