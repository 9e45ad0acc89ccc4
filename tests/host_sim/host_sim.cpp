/*
 * host_sim.cpp -- runs the KERNEL's per-block arithmetic (gpu_video_codec_amd/csrc/deblock_core.h,
 * and deblock_packed.h when present) on the CPU over a whole plane, with the same virtual zero
 * padding and bS guards the kernels use.  TEST-ONLY: lets the CPU test-suite compare the kernel
 * arithmetic with the oracle bit-for-bit without a GPU.  It is not part of the product library and
 * nothing in the product can reach it.
 */
#include <cstdint>
#include <cstring>

#include "../../gpu_video_codec_amd/csrc/deblock_core.h"
#include "../../gpu_video_codec_amd/csrc/deblock_h265.h"
#if __has_include("../../gpu_video_codec_amd/csrc/deblock_packed.h")
#define DBK_HOST_SIM 1
#include "../../gpu_video_codec_amd/csrc/deblock_packed.h"
#include "../../gpu_video_codec_amd/csrc/deblock_packed_h265.h"
#include "../../gpu_video_codec_amd/csrc/deblock_packed16.h"
#define HAVE_PACKED 1
#else
#define HAVE_PACKED 0
#endif

template <typename T>
static void load_block(const T *plane, long pitch_s, int w, int h, int bx, int by, int (&v)[8][8])
{
    for (int r = 0; r < 8; r++)
        for (int c = 0; c < 8; c++) {
            const int x = bx * 8 - 4 + c, y = by * 8 - 4 + r;
            v[r][c] = (x >= 0 && x < w && y >= 0 && y < h) ? plane[(long)y * pitch_s + x] : 0;
        }
}
template <typename T>
static void store_block(T *plane, long pitch_s, int w, int h, int bx, int by, const int (&v)[8][8])
{
    for (int r = 0; r < 8; r++)
        for (int c = 0; c < 8; c++) {
            const int x = bx * 8 - 4 + c, y = by * 8 - 4 + r;
            if (x >= 0 && x < w && y >= 0 && y < h) plane[(long)y * pitch_s + x] = (T)v[r][c];
        }
}

template <typename T>
static void run(T *plane, int w, int h, long pitch_s, int is_chroma, const uint8_t *vbs, const uint8_t *hbs,
                int tc, int beta, int max_v, const uint8_t *map, int map_stride, int ctu_log2,
                const uint8_t *tc_tab, const uint8_t *beta_tab, int shift, int packed)
{
    const int nbx = w / 8 + 1, nby = h / 8 + 1;
    const int limit_bx = is_chroma ? 2 * w / 8 : nbx - 1, limit_by = is_chroma ? 2 * h / 8 : nby - 1;
    const int n_vert = (w / 8 + 1) * h / 8, n_hor = (h / 8 + 1) * w / 8;
#if HAVE_PACKED
    /* QP-map launches of the packed luma kernels take a segment's operands from the workgroup's table (deblock_packed.h,
     * ktab_build), indexed by the segment's QP: built here the way ktab_setup of deblock_kernels.hip builds it */
    uint32_t ktab[dbk::kKTabDwords];
    if (map && packed && !is_chroma)
        dbk::ktab_build<false>(ktab, 0, 1, [&](int i) { return i < 52 ? (int)beta_tab[i] << shift : 0; },
                               [&](int i) { return i < 52 ? (int)tc_tab[i] << shift : 0; });
#endif
    for (int by = 0; by < nby; by++)
        for (int bx = 0; bx < nbx; bx++) {
            int v[8][8];
            load_block(plane, pitch_s, w, h, bx, by, v);
            dbk::BlockBs bs = dbk::load_block_bs(vbs, hbs, bx, by, w / 8 + 1, w / 8, limit_bx, limit_by, n_vert, n_hor);
            dbk::BlockQp q;
#if HAVE_PACKED
            dbk::QsTable qt;
            qt.tab = ktab;
            if (map && packed && !is_chroma) { /* the kernels' four map units (block_unit_qps) and their pairing */
                int u[4];
                dbk::block_unit_qps(map, map_stride, ctu_log2, 1, w, h, bx * 8 - 4, by * 8 - 4, u);
                qt.ib[0] = qt.it[0] = dbk::seg_qp_avg(u[0], u[1]);
                qt.ib[1] = qt.it[1] = dbk::seg_qp_avg(u[2], u[3]);
                qt.ib[2] = qt.it[2] = dbk::seg_qp_avg(u[0], u[2]);
                qt.ib[3] = qt.it[3] = dbk::seg_qp_avg(u[1], u[2]);
            }
#endif
            if (map) {
                const int sc = is_chroma ? 2 : 1, lw = w * sc, lh = h * sc, x0 = bx * 8 - 4, y0 = by * 8 - 4;
                const int qs[4] = {
                    dbk::seg_qp_from_map(map, map_stride, ctu_log2, sc, lw, lh, x0 + 3, y0 + 0, x0 + 4, y0 + 0),
                    dbk::seg_qp_from_map(map, map_stride, ctu_log2, sc, lw, lh, x0 + 3, y0 + 4, x0 + 4, y0 + 4),
                    dbk::seg_qp_from_map(map, map_stride, ctu_log2, sc, lw, lh, x0 + 0, y0 + 3, x0 + 0, y0 + 4),
                    dbk::seg_qp_from_map(map, map_stride, ctu_log2, sc, lw, lh, x0 + 4, y0 + 3, x0 + 0, y0 + 4)};
                for (int s = 0; s < 4; s++) { q.tc[s] = tc_tab[qs[s]] << shift; q.beta[s] = beta_tab[qs[s]] << shift; }
            } else {
                for (int s = 0; s < 4; s++) { q.tc[s] = tc; q.beta[s] = beta; }
            }
#if HAVE_PACKED
            if (packed && sizeof(T) == 2 && !is_chroma) {
                /* 16-bit containers through the packed core (luma, scalar QP) */
                uint32_t W[8][4];
                for (int r = 0; r < 8; r++)
                    for (int j = 0; j < 4; j++) W[r][j] = (uint32_t)v[r][2 * j] | ((uint32_t)v[r][2 * j + 1] << 16);
                /* UNI (second template argument) exactly as the kernels instantiate it: one QP => true, QP map => false */
                if (max_v > 2047) { /* 12 bit: wide sums */
                    if (map) dbk::packed_filter_luma_block16_src<true>(W, bs, qt, max_v);
                    else dbk::packed_filter_luma_block16<true, true>(W, bs, q, max_v);
                } else {
                    if (map) dbk::packed_filter_luma_block16_src<false>(W, bs, qt, max_v);
                    else dbk::packed_filter_luma_block16<false, true>(W, bs, q, max_v);
                }
                for (int r = 0; r < 8; r++)
                    for (int j = 0; j < 4; j++) {
                        v[r][2 * j] = W[r][j] & 0xffff;
                        v[r][2 * j + 1] = W[r][j] >> 16;
                    }
                store_block(plane, pitch_s, w, h, bx, by, v);
                continue;
            }
            if (packed && sizeof(T) == 2 && is_chroma) {
                uint32_t W[8][4];
                for (int r = 0; r < 8; r++)
                    for (int j = 0; j < 4; j++) W[r][j] = (uint32_t)v[r][2 * j] | ((uint32_t)v[r][2 * j + 1] << 16);
                dbk::packed_filter_chroma_block16(W, bs, q, max_v);
                for (int r = 0; r < 8; r++)
                    for (int j = 0; j < 4; j++) {
                        v[r][2 * j] = W[r][j] & 0xffff;
                        v[r][2 * j + 1] = W[r][j] >> 16;
                    }
                store_block(plane, pitch_s, w, h, bx, by, v);
                continue;
            }
            if (packed && sizeof(T) == 1) {
                uint32_t L[8], R[8];
                for (int r = 0; r < 8; r++) {
                    L[r] = (uint32_t)v[r][0] | ((uint32_t)v[r][1] << 8) | ((uint32_t)v[r][2] << 16) | ((uint32_t)v[r][3] << 24);
                    R[r] = (uint32_t)v[r][4] | ((uint32_t)v[r][5] << 8) | ((uint32_t)v[r][6] << 16) | ((uint32_t)v[r][7] << 24);
                }
                if (is_chroma) dbk::packed_filter_block<true>(L, R, bs, q);
                else if (map) dbk::packed_filter_luma_block_src(L, R, bs, qt);
                else dbk::packed_filter_block<false, true>(L, R, bs, q);
                for (int r = 0; r < 8; r++)
                    for (int c = 0; c < 4; c++) {
                        v[r][c] = (L[r] >> (8 * c)) & 0xff;
                        v[r][4 + c] = (R[r] >> (8 * c)) & 0xff;
                    }
                store_block(plane, pitch_s, w, h, bx, by, v);
                continue;
            }
#else
            (void)packed;
#endif
            if (is_chroma) dbk::filter_block_generic<true>(v, bs, q, max_v);
            else dbk::filter_block_generic<false>(v, bs, q, max_v);
            store_block(plane, pitch_s, w, h, bx, by, v);
        }
}

extern "C" int host_sim_have_packed(void) { return HAVE_PACKED; }
/* the spec-exact one-QP kernels' form for waves that hold bS 1 next to bS 2 (LumaKSel), for every block */
extern "C" void host_sim_h265_force_mixed(int on)
{
#if HAVE_PACKED
    dbk::h265_sim_force_mixed_flag() = on;
#else
    (void)on;
#endif
}
/* the launcher's operand-range predicate for the packed luma core (deblock_packed.h) */
extern "C" int host_sim_packed_luma_tc_fits(int max_v, int tc_max)
{
#if HAVE_PACKED
    return dbk::packed_luma_tc_fits(max_v, tc_max) ? 1 : 0;
#else
    (void)max_v; (void)tc_max;
    return 0;
#endif
}

extern "C" void host_sim_filter_plane(void *plane, int w, int h, long pitch_bytes, int sample_bytes, int is_chroma,
                                      const uint8_t *vbs, const uint8_t *hbs, int tc, int beta, int max_v,
                                      const uint8_t *map, int map_stride, int ctu_log2,
                                      const uint8_t *tc_tab, const uint8_t *beta_tab, int shift, int packed)
{
    if (sample_bytes == 1)
        run((uint8_t *)plane, w, h, pitch_bytes, is_chroma, vbs, hbs, tc, beta, max_v, map, map_stride, ctu_log2, tc_tab, beta_tab, shift, packed);
    else
        run((uint16_t *)plane, w, h, pitch_bytes / 2, is_chroma, vbs, hbs, tc, beta, max_v, map, map_stride, ctu_log2, tc_tab, beta_tab, shift, packed);
}

/* ---- spec-exact mode (deblock_h265.h) ---------------------------------------------------------------- */

template <typename T>
static void run_h265(T *plane, int w, int h, long pitch_s, int c_idx, const uint8_t *vbs4, const uint8_t *hbs4, int qp,
                     const uint8_t *map, int map_stride, int unit_log2, const dbk::H265Prm &prm, int packed)
{
    const int nbx = w / 8 + 1, nby = h / 8 + 1, sc = c_idx ? 2 : 1;
#if HAVE_PACKED
    /* the one-QP kernels' per-block scalars, derived as dbk_launch_packed_h265 derives them (luma); a block is a "wave" of
     * one lane here, so every block without keep flags takes the uniform path and every block with one the general path */
    const dbk::H265Uni uni = dbk::h265_uni(dbk::h265_beta(dbk::clampi(qp + prm.beta_off, 0, 51)) << prm.shift,
                                           dbk::h265_tc(dbk::clampi(qp + prm.tc_off, 0, 53)) << prm.shift,
                                           dbk::h265_tc(dbk::clampi(qp + 2 + prm.tc_off, 0, 53)) << prm.shift);
    const dbk::H265Uni *const u = (map || c_idx) ? nullptr : &uni;
    /* QP-map luma launches of the packed kernels: operands from the workgroup's table, as ktab_setup_h265 builds it */
    const bool tabbed = map && !c_idx;
    uint32_t ktab[dbk::kKTabDwords];
    if (tabbed)
        dbk::ktab_build<true>(ktab, 0, 1, [&](int i) { return dbk::h265_beta(i < 52 ? i : 51) << prm.shift; },
                              [&](int i) { return dbk::h265_tc(i) << prm.shift; });
#endif
    for (int by = 0; by < nby; by++)
        for (int bx = 0; bx < nbx; bx++) {
            int v[8][8], entry[4], qpl[4];
            load_block(plane, pitch_s, w, h, bx, by, v);
            dbk::load_block_bs_h265(vbs4, hbs4, bx, by, nbx, nby, w / 8 + 1, w / 4, entry);
            dbk::h265_block_qpl(map, map_stride, unit_log2, sc, w * sc, h * sc, bx * 8 - 4, by * 8 - 4, qp, qpl);
#if HAVE_PACKED
            if (packed && sizeof(T) == 2) {
                uint32_t W[8][4];
                for (int r = 0; r < 8; r++)
                    for (int j = 0; j < 4; j++) W[r][j] = (uint32_t)v[r][2 * j] | ((uint32_t)v[r][2 * j + 1] << 16);
                dbk::H265Seg sg;
                if (c_idx) { dbk::h265_seg_params<true>(entry, qpl, prm, sg); dbk::packed_filter_block16_h265<true>(W, sg, prm.max_v); }
                else if (tabbed) {
                    dbk::h265_seg_rows(entry, qpl, prm, ktab, sg);
                    if (prm.max_v > 2047) dbk::packed_filter_block16_h265<false, true, true>(W, sg, prm.max_v);
                    else dbk::packed_filter_block16_h265<false, false, true>(W, sg, prm.max_v);
                } else {
                    dbk::h265_seg_params<false>(entry, qpl, prm, sg);
                    if (prm.max_v > 2047) dbk::packed_filter_block16_h265<false, true>(W, sg, prm.max_v, u);
                    else dbk::packed_filter_block16_h265<false>(W, sg, prm.max_v, u);
                }
                for (int r = 0; r < 8; r++)
                    for (int j = 0; j < 4; j++) {
                        v[r][2 * j] = W[r][j] & 0xffff;
                        v[r][2 * j + 1] = W[r][j] >> 16;
                    }
                store_block(plane, pitch_s, w, h, bx, by, v);
                continue;
            }
            if (packed && sizeof(T) == 1) {
                uint32_t L[8], R[8];
                for (int r = 0; r < 8; r++) {
                    L[r] = (uint32_t)v[r][0] | ((uint32_t)v[r][1] << 8) | ((uint32_t)v[r][2] << 16) | ((uint32_t)v[r][3] << 24);
                    R[r] = (uint32_t)v[r][4] | ((uint32_t)v[r][5] << 8) | ((uint32_t)v[r][6] << 16) | ((uint32_t)v[r][7] << 24);
                }
                dbk::H265Seg sg;
                if (c_idx) { dbk::h265_seg_params<true>(entry, qpl, prm, sg); dbk::packed_filter_block_h265<true>(L, R, sg); }
                else if (tabbed) { dbk::h265_seg_rows(entry, qpl, prm, ktab, sg); dbk::packed_filter_block_h265<false, true>(L, R, sg); }
                else { dbk::h265_seg_params<false>(entry, qpl, prm, sg); dbk::packed_filter_block_h265<false>(L, R, sg, u); }
                for (int r = 0; r < 8; r++)
                    for (int c = 0; c < 4; c++) {
                        v[r][c] = (L[r] >> (8 * c)) & 0xff;
                        v[r][4 + c] = (R[r] >> (8 * c)) & 0xff;
                    }
                store_block(plane, pitch_s, w, h, bx, by, v);
                continue;
            }
#else
            (void)packed;
#endif
            if (c_idx) dbk::filter_block_h265<true>(v, entry, qpl, prm);
            else dbk::filter_block_h265<false>(v, entry, qpl, prm);
            store_block(plane, pitch_s, w, h, bx, by, v);
        }
}

extern "C" void host_sim_h265_filter_plane(void *plane, int w, int h, long pitch_bytes, int sample_bytes, int bit_depth,
                                           int c_idx, const uint8_t *vbs4, const uint8_t *hbs4, int qp, const uint8_t *map,
                                           int map_stride, int unit_log2, int tc_offset_div2, int beta_offset_div2,
                                           int c_qp_offset, int packed)
{
    const dbk::H265Prm prm = {tc_offset_div2 * 2, beta_offset_div2 * 2, c_qp_offset, bit_depth - 8, (1 << bit_depth) - 1};
    if (sample_bytes == 1) run_h265((uint8_t *)plane, w, h, pitch_bytes, c_idx, vbs4, hbs4, qp, map, map_stride, unit_log2, prm, packed);
    else run_h265((uint16_t *)plane, w, h, pitch_bytes / 2, c_idx, vbs4, hbs4, qp, map, map_stride, unit_log2, prm, packed);
}

extern "C" void host_sim_h265_derive_bs(const uint16_t *flags, const int16_t *mv0, const int16_t *mv1, const int32_t *ref0,
                                        const int32_t *ref1, int w, int h, uint8_t *vbs4, uint8_t *hbs4)
{
    const dbk::H265Units u = {flags, mv0, mv1, ref0, ref1};
    const long long uw = w / 4;
    const int vstride = w / 8 + 1, hstride = w / 4;
    std::memset(vbs4, 0, (size_t)vstride * (h / 4));
    std::memset(hbs4, 0, (size_t)(h / 8 + 1) * hstride);
    for (int y4 = 0; y4 < h / 4; y4++)
        for (int bx = 1; bx < w / 8; bx++)
            vbs4[y4 * vstride + bx] = (uint8_t)dbk::h265_bs_of_edge(u, y4 * uw + 2 * bx - 1, y4 * uw + 2 * bx, true);
    for (int by = 1; by < h / 8; by++)
        for (int x4 = 0; x4 < w / 4; x4++)
            hbs4[by * hstride + x4] = (uint8_t)dbk::h265_bs_of_edge(u, (2 * by - 1) * uw + x4, (2 * by) * uw + x4, false);
}
