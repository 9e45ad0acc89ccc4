/*
 * deblock_host_h265.cpp -- C-ABI entries of the spec-exact mode (ITU-T H.265 clause 8.7.2: bS derivation, block filter,
 * host-frame operator) and of sample adaptive offset (clause 8.7.3).  Context and shared helpers: deblock_ctx.h.
 */
#include <chrono>
#include <cstring>

#include "deblock_ctx.h"
#ifdef HEVCDBK_DIAG
#include "hevcdbk_diag.h"
#endif

using namespace dbkh;

extern "C" {

/* ---- spec-exact mode (H.265 clause 8.7.2) ------------------------------------------------------------------ */

size_t hevcdbk_h265_num_vert_bs(unsigned w, unsigned h) { return (size_t)(w / 8 + 1) * (h / 4); }
size_t hevcdbk_h265_num_hor_bs(unsigned w, unsigned h) { return (size_t)(h / 8 + 1) * (w / 4); }

int hevcdbk_h265_derive_bs_device(hevcdbk_context *ctx, const hevcdbk_h265_units *u, unsigned width, unsigned height,
                                  uint8_t *vert_bs4, uint8_t *hor_bs4, uint8_t *chroma_vert_bs4, uint8_t *chroma_hor_bs4,
                                  void *hip_stream)
{
    if (!ctx || !u || !u->flags || !u->mv0 || !u->mv1 || !u->ref0 || !u->ref1 || !vert_bs4 || !hor_bs4) return HEVCDBK_ERR_ARG;
    if ((chroma_vert_bs4 != nullptr) != (chroma_hor_bs4 != nullptr)) return HEVCDBK_ERR_ARG;
    if (width == 0 || height == 0 || width % 8 != 0 || height % 8 != 0) return HEVCDBK_ERR_DIMENSIONS;
    if (chroma_vert_bs4 && ((width / 2) % 8 != 0 || (height / 2) % 8 != 0)) return HEVCDBK_ERR_DIMENSIONS;
    if (int rc = bind(ctx)) return rc;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->compute;
    const hipError_t e = dbk_launch_h265_bs(u->flags, u->mv0, u->mv1, u->ref0, u->ref1, (int)width, (int)height, vert_bs4,
                                            hor_bs4, chroma_vert_bs4, chroma_hor_bs4, s);
    return hip_ok(ctx, e, "bS derivation launch") ? HEVCDBK_OK : HEVCDBK_ERR_HIP;
}

namespace {

int h265_args(const hevcdbk_device_planes *planes, int c_idx, unsigned qp, const hevcdbk_h265_params *prm, DbkH265Args &h)
{
    if (!planes || c_idx < 0 || c_idx > 2 || (c_idx != 0) != (planes->is_chroma != 0)) return HEVCDBK_ERR_ARG;
    const hevcdbk_h265_params zero = {0, 0, 0, 0};
    if (!prm) prm = &zero;
    if (prm->tc_offset_div2 < -6 || prm->tc_offset_div2 > 6 || prm->beta_offset_div2 < -6 || prm->beta_offset_div2 > 6 ||
        prm->cb_qp_offset < -12 || prm->cb_qp_offset > 12 || prm->cr_qp_offset < -12 || prm->cr_qp_offset > 12)
        return HEVCDBK_ERR_ARG;
    if (planes->qp_map && (planes->ctu_log2 < 3 || planes->ctu_log2 > 8)) return HEVCDBK_ERR_ARG;
    if (int rc = planes_to_args(planes, qp, nullptr, h.base)) return rc;
    h.base.hstride = (int)(planes->plane_w / 4);
    h.base.n_vert = (int)hevcdbk_h265_num_vert_bs(planes->plane_w, planes->plane_h);
    h.base.n_hor = (int)hevcdbk_h265_num_hor_bs(planes->plane_w, planes->plane_h);
    h.qp = (int)(qp > 51 ? 51 : qp);
    h.tc_off = prm->tc_offset_div2 * 2;
    h.beta_off = prm->beta_offset_div2 * 2;
    h.c_qp_offset = c_idx == 1 ? prm->cb_qp_offset : (c_idx == 2 ? prm->cr_qp_offset : 0);
    return HEVCDBK_OK;
}

int launch_h265(hevcdbk_context *ctx, const DbkH265Args &h0, int sample_bytes, bool chroma, int variant, hipStream_t s)
{
    const int map = variant & HEVCDBK_MAP_MASK; /* as in dbkh::launch */
    variant &= ~HEVCDBK_MAP_MASK;
    #ifdef HEVCDBK_DIAG
    if (map != HEVCDBK_MAP_AUTO && map != HEVCDBK_MAP_ROWS && map != HEVCDBK_MAP_LINEAR && map != HEVCDBK_DIAG_MAP_TILES && map != HEVCDBK_DIAG_MAP_STRIPE && map != HEVCDBK_DIAG_MAP_PIPE && map != HEVCDBK_DIAG_MAP_GROUP) return HEVCDBK_ERR_ARG;
#else
    if (map != HEVCDBK_MAP_AUTO && map != HEVCDBK_MAP_ROWS && map != HEVCDBK_MAP_LINEAR) return HEVCDBK_ERR_ARG;
#endif
    DbkH265Args h = h0;
    h.base.map_override = map == HEVCDBK_MAP_ROWS ? 1 : (map == HEVCDBK_MAP_LINEAR ? 2 : (map == 0x300 ? 3 : (map == 0x400 ? 4 : (map == 0x500 ? 5 : (map == 0x600 ? 6 : 0)))));
    const bool can_pack = dbk_packed_h265_supports(h, sample_bytes, chroma);
    hipError_t e;
    if (variant == HEVCDBK_KERNEL_PACKED) {
        if (!can_pack) return HEVCDBK_ERR_UNSUPPORTED;
        e = dbk_launch_packed_h265(h, sample_bytes, chroma, s);
    } else if (variant == HEVCDBK_KERNEL_GENERIC) {
        e = dbk_launch_h265(h, sample_bytes, chroma, s);
    } else if (variant == HEVCDBK_KERNEL_AUTO) {
        e = can_pack ? dbk_launch_packed_h265(h, sample_bytes, chroma, s) : dbk_launch_h265(h, sample_bytes, chroma, s);
    } else {
        return HEVCDBK_ERR_ARG;
    }
    return hip_ok(ctx, e, "kernel launch") ? HEVCDBK_OK : HEVCDBK_ERR_HIP;
}

} /* namespace */

int hevc_deblocking_filter_h265_device(hevcdbk_context *ctx, const hevcdbk_device_planes *planes, int c_idx, unsigned qp,
                                       const hevcdbk_h265_params *params, int kernel_variant, void *hip_stream)
{
    if (!ctx) return HEVCDBK_ERR_ARG;
    DbkH265Args h;
    if (int rc = h265_args(planes, c_idx, qp, params, h)) return rc;
    if (int rc = bind(ctx)) return rc;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->compute;
    return launch_h265(ctx, h, (int)planes->sample_bytes, c_idx != 0, kernel_variant, s);
}

int hevc_deblocking_filter_h265(hevcdbk_context *ctx, hevcdbk_frame *frame, const hevcdbk_h265_units *units,
                                const hevcdbk_bs *bs4, const hevcdbk_qp *qp, const hevcdbk_h265_params *params,
                                hevcdbk_timing *timing)
{
    if (!ctx || !frame || !qp || !frame->plane[0] || (units != nullptr) == (bs4 != nullptr)) return HEVCDBK_ERR_ARG;
    if (bad_depth(frame->bit_depth, frame->sample_bytes)) return HEVCDBK_ERR_ARG;
    const unsigned W = frame->width, H = frame->height, sb = frame->sample_bytes;
    if (W == 0 || H == 0 || W % 8 != 0 || H % 8 != 0) return HEVCDBK_ERR_DIMENSIONS;
    const bool chroma = frame->plane[1] && frame->plane[2];
    if (chroma && ((W / 2) % 8 != 0 || (H / 2) % 8 != 0)) return HEVCDBK_ERR_DIMENSIONS;
    const int npl = chroma ? 3 : 1;
    const unsigned pw[3] = {W, W / 2, W / 2}, ph[3] = {H, H / 2, H / 2};
    for (int i = 0; i < npl; i++)
        if (frame->pitch[i] < (size_t)pw[i] * sb) return HEVCDBK_ERR_ARG;
    const size_t nv = hevcdbk_h265_num_vert_bs(W, H), nh = hevcdbk_h265_num_hor_bs(W, H);
    const size_t ncv = chroma ? hevcdbk_h265_num_vert_bs(W / 2, H / 2) : 0, nch = chroma ? hevcdbk_h265_num_hor_bs(W / 2, H / 2) : 0;
    if (bs4) {
        if (!bs4->vert || !bs4->hor) return HEVCDBK_ERR_ARG;
        if (bs4->n_vert != nv || bs4->n_hor != nh) return HEVCDBK_ERR_BS_SIZE;
    } else if (!units->flags || !units->mv0 || !units->mv1 || !units->ref0 || !units->ref1) {
        return HEVCDBK_ERR_ARG;
    }
    size_t map_rows = 0;
    if (qp->map) {
        if (qp->ctu_log2 < 3 || qp->ctu_log2 > 8 || qp->map_stride < ((W + (1u << qp->ctu_log2) - 1) >> qp->ctu_log2))
            return HEVCDBK_ERR_ARG;
        map_rows = (H + (1u << qp->ctu_log2) - 1) >> qp->ctu_log2;
    }
    if (int rc = bind(ctx)) return rc;

    size_t plane_bytes[3] = {0, 0, 0}, plane_off[3] = {0, 0, 0}, frame_bytes = 0;
    for (int i = 0; i < npl; i++) {
        plane_bytes[i] = (size_t)pw[i] * ph[i] * sb;
        plane_off[i] = frame_bytes;
        frame_bytes = (frame_bytes + plane_bytes[i] + 255) & ~(size_t)255;
    }
    if (int rc = grow_pinned(ctx, ctx->pin[0], frame_bytes)) return rc;
    if (int rc = grow_device(ctx, ctx->dev[0], frame_bytes)) return rc;
    if (int rc = grow_device(ctx, ctx->dev_bs, nv + nh + ncv + nch)) return rc;
    ctx->bs_default_at = nullptr; /* dev_bs no longer holds the reference's default pattern */
    if (qp->map)
        if (int rc = grow_device(ctx, ctx->dev_map, map_rows * qp->map_stride)) return rc;
    const size_t U = (size_t)(W / 4) * (H / 4);
    if (units)
        if (int rc = grow_device(ctx, ctx->dev_units, 18 * U)) return rc;
    uint8_t *dbs = (uint8_t *)ctx->dev_bs.p, *dun = (uint8_t *)ctx->dev_units.p;
    uint8_t *dmap = qp->map ? (uint8_t *)ctx->dev_map.p : nullptr;
    hipStream_t s = ctx->compute;
    hipEvent_t *ev = ctx->ev;

    /* The frame's way in and out (round 4, as hevc_deblocking_filter's large-frame path): on a large-BAR device the staging crew
     * writes the caller's rows straight into fine-grained HBM through the BAR and the kernels store their results straight into
     * the page-locked ring, from which the crew copies them back -- no DMA in either direction; elsewhere ring + DMA both ways,
     * the ring filled and emptied by the crew.  Not cut into strips: the bS derivation wants the whole picture's units first. */
    const auto wall0 = std::chrono::steady_clock::now();
    uint8_t *const ring = (uint8_t *)ctx->pin[0].p;
    uint8_t *const dpush = push_buffer(ctx, frame_bytes);
    uint8_t *const din = dpush ? dpush : (uint8_t *)ctx->dev[0].p;    /* what the kernels read */
    uint8_t *const dout = dpush ? ring : (uint8_t *)ctx->dev[0].p;    /* what they write */
    if (int rc = crew_copy_frame(ctx, frame, npl, pw, ph, sb, dpush ? dpush : ring, plane_off, false, dpush != nullptr)) return rc;
    const double push_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
    HIP_TRY(ctx, hipEventRecord(ev[0], s));
    if (!dpush) HIP_TRY(ctx, hipMemcpyAsync(ctx->dev[0].p, ring, frame_bytes, hipMemcpyHostToDevice, s));
    if (dmap) HIP_TRY(ctx, hipMemcpyAsync(dmap, qp->map, map_rows * qp->map_stride, hipMemcpyHostToDevice, s));
    if (units) {
        /* device layout: ref0 | ref1 | mv0 | mv1 | flags (descending alignment) */
        HIP_TRY(ctx, hipMemcpyAsync(dun, units->ref0, 4 * U, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpyAsync(dun + 4 * U, units->ref1, 4 * U, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpyAsync(dun + 8 * U, units->mv0, 4 * U, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpyAsync(dun + 12 * U, units->mv1, 4 * U, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpyAsync(dun + 16 * U, units->flags, 2 * U, hipMemcpyHostToDevice, s));
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(dbs, bs4->vert, nv, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpyAsync(dbs + nv, bs4->hor, nh, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(ctx, hipEventRecord(ev[1], s));
    uint8_t *dcv = chroma ? dbs + nv + nh : nullptr, *dch = chroma ? dbs + nv + nh + ncv : nullptr;
    hipError_t e;
    if (units) e = dbk_launch_h265_bs(dun + 16 * U, dun + 8 * U, dun + 12 * U, dun, dun + 4 * U, (int)W, (int)H, dbs, dbs + nv, dcv, dch, s);
    else e = chroma ? dbk_launch_h265_chroma_bs(dbs, dbs + nv, (int)W, (int)H, dcv, dch, s) : hipSuccess;
    if (!hip_ok(ctx, e, "bS derivation launch")) return HEVCDBK_ERR_HIP;
    for (int i = 0; i < npl; i++) {
        hevcdbk_device_planes p;
        std::memset(&p, 0, sizeof(p));
        p.src = din + plane_off[i];
        p.dst = dout + plane_off[i];
        p.pitch = (size_t)pw[i] * sb; p.frame_stride = plane_bytes[i]; p.n_frames = 1;
        p.plane_w = pw[i]; p.plane_h = ph[i]; p.bit_depth = frame->bit_depth; p.sample_bytes = sb;
        p.is_chroma = i != 0;
        p.vert_bs = i == 0 ? dbs : dcv;
        p.hor_bs = i == 0 ? dbs + nv : dch;
        p.qp_map = dmap; p.qp_map_stride = qp->map_stride; p.ctu_log2 = qp->ctu_log2;
        DbkH265Args h;
        if (int rc = h265_args(&p, i, qp->qp, params, h)) return rc;
        if (int rc = launch_h265(ctx, h, (int)sb, i != 0, HEVCDBK_KERNEL_AUTO, s)) return rc;
    }
    HIP_TRY(ctx, hipEventRecord(ev[2], s));
    if (!dpush) HIP_TRY(ctx, hipMemcpyAsync(ring, ctx->dev[0].p, frame_bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipEventRecord(ev[3], s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    if (int rc = crew_copy_frame(ctx, frame, npl, pw, ph, sb, ring, plane_off, true, false)) return rc;
    const auto wall1 = std::chrono::steady_clock::now();
    if (timing) {
        float a = 0.f, b = 0.f, c = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&a, ev[0], ev[1]));
        HIP_TRY(ctx, hipEventElapsedTime(&b, ev[1], ev[2]));
        HIP_TRY(ctx, hipEventElapsedTime(&c, ev[2], ev[3]));
        timing->exec_s = b * 1e-3;   /* with the BAR path the download is the kernels' own stores: inside exec_s */
        timing->copy_s = (a + c) * 1e-3 + (dpush ? push_s : 0.0); /* uploads of bS / units / map (+ the frame: DMA, or the crew's writes) */
        timing->total_s = timing->exec_s + timing->copy_s;
        timing->pipelined_s = std::chrono::duration<double>(wall1 - wall0).count();
    }
    return HEVCDBK_OK;
}

/* ---- sample adaptive offset (H.265 clause 8.7.3) ------------------------------------------------------------- */

static_assert(sizeof(hevcdbk_sao_ctb) == sizeof(DbkSaoCtb) && sizeof(DbkSaoCtb) == 6, "SAO CTB entry layout");

namespace {

/* validates the SAO operands of `p` and fills `a` (src / dst as in `p`) */
int sao_args(const hevcdbk_device_planes *p, const hevcdbk_sao_ctb *params, unsigned params_stride, size_t params_frame_stride,
             unsigned ctb_log2, const uint8_t *keep, unsigned keep_stride, size_t keep_frame_stride, DbkSaoArgs &a)
{
    if (!p || !p->src || !p->dst || p->src == p->dst || !params) return HEVCDBK_ERR_ARG;
    if (bad_depth(p->bit_depth, p->sample_bytes)) return HEVCDBK_ERR_ARG;
    if (p->plane_w == 0 || p->plane_h == 0 || p->plane_w % 8 != 0 || p->plane_h % 8 != 0) return HEVCDBK_ERR_DIMENSIONS;
    if (ctb_log2 < 3 || ctb_log2 > 6) return HEVCDBK_ERR_ARG;
    if (params_stride < ((p->plane_w + (1u << ctb_log2) - 1) >> ctb_log2)) return HEVCDBK_ERR_ARG;
    if (keep && keep_stride < p->plane_w / 8) return HEVCDBK_ERR_ARG;
    const size_t align = 4 * p->sample_bytes;
    if (p->pitch % align != 0 || p->frame_stride % align != 0 || (uintptr_t)p->src % align != 0 || (uintptr_t)p->dst % align != 0)
        return HEVCDBK_ERR_UNSUPPORTED;
    if (p->pitch < (size_t)p->plane_w * p->sample_bytes || p->n_frames > 65535 || p->plane_h > 65535) return HEVCDBK_ERR_ARG;
    std::memset(&a, 0, sizeof(a));
    a.src = (const uint8_t *)p->src; a.dst = (uint8_t *)p->dst;
    a.pitch = (long long)p->pitch; a.frame_stride = (long long)p->frame_stride;
    a.plane_w = (int)p->plane_w; a.plane_h = (int)p->plane_h; a.n_frames = (int)p->n_frames;
    a.max_v = (1 << p->bit_depth) - 1; a.band_shift = (int)p->bit_depth - 5;
    a.params = reinterpret_cast<const DbkSaoCtb *>(params);
    a.params_stride = (int)params_stride; a.params_frame_stride = (long long)params_frame_stride;
    a.ctb_log2 = (int)ctb_log2;
    a.keep = keep; a.keep_stride = (int)keep_stride; a.keep_frame_stride = (long long)keep_frame_stride;
    return HEVCDBK_OK;
}

/* the two-launch form of deblocking + SAO: the deblocked planes go through ctx->dev_tmp (same pitch and frame stride).
 * The scratch plane is ONE buffer per context while the caller may hand in any stream: the previous user's SAO launch is
 * fenced by an event (ctx->tmp_ev) that the next user's stream waits for before it overwrites the scratch, and the buffer
 * is only re-allocated after that event has completed (two calls on different streams raced on it) */
int tmp_planes(hevcdbk_context *ctx, const hevcdbk_device_planes *p, hipStream_t s, hevcdbk_device_planes &first,
               hevcdbk_device_planes &second)
{
    const size_t bytes = (size_t)p->frame_stride * (p->n_frames ? p->n_frames - 1 : 0) + (size_t)p->pitch * p->plane_h;
    if (!ctx->tmp_ev) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->tmp_ev, hipEventDisableTiming));
    if (ctx->tmp_used) {
        if (ctx->dev_tmp.cap < bytes) HIP_TRY(ctx, hipEventSynchronize(ctx->tmp_ev)); /* about to free it */
        else HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->tmp_ev, 0));
    }
    if (int rc = grow_device(ctx, ctx->dev_tmp, bytes)) return rc;
    first = *p;
    first.dst = ctx->dev_tmp.p;
    second = *p;
    second.src = ctx->dev_tmp.p;
    return HEVCDBK_OK;
}
int tmp_done(hevcdbk_context *ctx, hipStream_t s)
{
    HIP_TRY(ctx, hipEventRecord(ctx->tmp_ev, s));
    ctx->tmp_used = true;
    return HEVCDBK_OK;
}

/* one plane: the fused kernel where it applies (and is not switched off), else the two launches */
int deblock_sao_plane(hevcdbk_context *ctx, const hevcdbk_device_planes *p, unsigned qp, const hevcdbk_tables *tables, DbkArgs &da,
                      DbkSaoArgs &sa, int fused, hipStream_t s)
{
    const bool can = dbk_deblock_sao_supports(da, sa, (int)p->sample_bytes, p->is_chroma != 0);
    if (fused == HEVCDBK_FUSED_ON && !can) return HEVCDBK_ERR_UNSUPPORTED;
    if (can && fused != HEVCDBK_FUSED_OFF)
        return hip_ok(ctx, dbk_launch_deblock_sao(da, sa, (int)p->sample_bytes, p->is_chroma != 0, s), "fused deblocking + SAO launch") ? HEVCDBK_OK : HEVCDBK_ERR_HIP;
    hevcdbk_device_planes first, second;
    if (int rc = tmp_planes(ctx, p, s, first, second)) return rc;
    if (int rc = planes_to_args(&first, qp, tables, da)) return rc;
    if (int rc = launch(ctx, da, (int)p->sample_bytes, p->is_chroma != 0, HEVCDBK_KERNEL_AUTO, s)) return rc;
    sa.src = (const uint8_t *)second.src;
    if (!hip_ok(ctx, dbk_launch_sao(sa, (int)p->sample_bytes, s), "SAO launch")) return HEVCDBK_ERR_HIP;
    return tmp_done(ctx, s);
}
int deblock_sao_plane_h265(hevcdbk_context *ctx, const hevcdbk_device_planes *p, int c_idx, unsigned qp, const hevcdbk_h265_params *prm,
                           DbkH265Args &h, DbkSaoArgs &sa, int fused, hipStream_t s)
{
    const bool can = dbk_packed_h265_supports(h, (int)p->sample_bytes, c_idx != 0) &&
                     dbk_deblock_sao_supports(h.base, sa, (int)p->sample_bytes, c_idx != 0);
    if (fused == HEVCDBK_FUSED_ON && !can) return HEVCDBK_ERR_UNSUPPORTED;
    if (can && fused != HEVCDBK_FUSED_OFF)
        return hip_ok(ctx, dbk_launch_deblock_sao_h265(h, sa, (int)p->sample_bytes, c_idx != 0, s), "fused deblocking + SAO launch") ? HEVCDBK_OK : HEVCDBK_ERR_HIP;
    hevcdbk_device_planes first, second;
    if (int rc = tmp_planes(ctx, p, s, first, second)) return rc;
    if (int rc = h265_args(&first, c_idx, qp, prm, h)) return rc;
    if (int rc = launch_h265(ctx, h, (int)p->sample_bytes, c_idx != 0, HEVCDBK_KERNEL_AUTO, s)) return rc;
    sa.src = (const uint8_t *)second.src;
    if (!hip_ok(ctx, dbk_launch_sao(sa, (int)p->sample_bytes, s), "SAO launch")) return HEVCDBK_ERR_HIP;
    return tmp_done(ctx, s);
}

} /* namespace */

int hevc_sao_filter_device(hevcdbk_context *ctx, const hevcdbk_device_planes *p, const hevcdbk_sao_ctb *params,
                           unsigned params_stride, size_t params_frame_stride, unsigned ctb_log2, const uint8_t *keep,
                           unsigned keep_stride, size_t keep_frame_stride, void *hip_stream)
{
    if (!ctx) return HEVCDBK_ERR_ARG;
    DbkSaoArgs a;
    if (int rc = sao_args(p, params, params_stride, params_frame_stride, ctb_log2, keep, keep_stride, keep_frame_stride, a)) return rc;
    if (int rc = bind(ctx)) return rc;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->compute;
    return hip_ok(ctx, dbk_launch_sao(a, (int)p->sample_bytes, s), "SAO launch") ? HEVCDBK_OK : HEVCDBK_ERR_HIP;
}

/* ---- deblocking followed by SAO in one call ---------------------------------------------- */

int hevc_deblock_sao_device(hevcdbk_context *ctx, const hevcdbk_device_planes *p, unsigned qp, const hevcdbk_tables *tables,
                            const hevcdbk_sao_ctb *params, unsigned params_stride, size_t params_frame_stride, unsigned ctb_log2,
                            const uint8_t *keep, unsigned keep_stride, size_t keep_frame_stride, int fused, void *hip_stream)
{
    if (!ctx || (fused != HEVCDBK_FUSED_AUTO && fused != HEVCDBK_FUSED_OFF && fused != HEVCDBK_FUSED_ON)) return HEVCDBK_ERR_ARG;
    DbkSaoArgs sa;
    if (int rc = sao_args(p, params, params_stride, params_frame_stride, ctb_log2, keep, keep_stride, keep_frame_stride, sa)) return rc;
    DbkArgs da;
    if (int rc = planes_to_args(p, qp, tables, da)) return rc;
    if (int rc = bind(ctx)) return rc;
    return deblock_sao_plane(ctx, p, qp, tables, da, sa, fused, hip_stream ? (hipStream_t)hip_stream : ctx->compute);
}

int hevc_deblock_sao_h265_device(hevcdbk_context *ctx, const hevcdbk_device_planes *p, int c_idx, unsigned qp,
                                 const hevcdbk_h265_params *prm, const hevcdbk_sao_ctb *params, unsigned params_stride,
                                 size_t params_frame_stride, unsigned ctb_log2, const uint8_t *keep, unsigned keep_stride,
                                 size_t keep_frame_stride, int fused, void *hip_stream)
{
    if (!ctx || (fused != HEVCDBK_FUSED_AUTO && fused != HEVCDBK_FUSED_OFF && fused != HEVCDBK_FUSED_ON)) return HEVCDBK_ERR_ARG;
    DbkSaoArgs sa;
    if (int rc = sao_args(p, params, params_stride, params_frame_stride, ctb_log2, keep, keep_stride, keep_frame_stride, sa)) return rc;
    DbkH265Args h;
    if (int rc = h265_args(p, c_idx, qp, prm, h)) return rc;
    if (int rc = bind(ctx)) return rc;
    return deblock_sao_plane_h265(ctx, p, c_idx, qp, prm, h, sa, fused, hip_stream ? (hipStream_t)hip_stream : ctx->compute);
}

/* ---- Y, U, V of a batch: deblocking + SAO of all planes in ONE launch where the fused kernel takes every plane ---- */

int hevc_deblock_sao_device_planes(hevcdbk_context *ctx, const hevcdbk_device_planes *planes, unsigned n_planes, unsigned qp,
                                   const hevcdbk_tables *tables, const hevcdbk_sao_plane *sao, int fused, void *hip_stream)
{
    if (!ctx || !planes || !sao || n_planes == 0 || n_planes > 3 ||
        (fused != HEVCDBK_FUSED_AUTO && fused != HEVCDBK_FUSED_OFF && fused != HEVCDBK_FUSED_ON))
        return HEVCDBK_ERR_ARG;
    DbkArgs da[3];
    DbkSaoArgs sa[3];
    bool one = n_planes >= 2 && fused != HEVCDBK_FUSED_OFF && !planes[0].is_chroma;
    for (unsigned i = 0; i < n_planes; i++) {
        if (int rc = sao_args(&planes[i], sao[i].params, sao[i].params_stride, sao[i].params_frame_stride, sao[i].ctb_log2, sao[i].keep,
                              sao[i].keep_stride, sao[i].keep_frame_stride, sa[i]))
            return rc;
        if (int rc = planes_to_args(&planes[i], qp, tables, da[i])) return rc;
        if (planes[i].n_frames != planes[0].n_frames) return HEVCDBK_ERR_ARG;
        one = one && (i == 0 || planes[i].is_chroma) && planes[i].sample_bytes == planes[0].sample_bytes &&
              planes[i].bit_depth == planes[0].bit_depth && (da[i].qp_map != nullptr) == (da[0].qp_map != nullptr) &&
              dbk_deblock_sao_supports(da[i], sa[i], (int)planes[i].sample_bytes, planes[i].is_chroma != 0);
    }
    if (int rc = bind(ctx)) return rc;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->compute;
    if (one)
        return hip_ok(ctx, dbk_launch_deblock_sao_multi(da, sa, (int)n_planes, (int)planes[0].sample_bytes, s), "fused deblocking + SAO launch")
                   ? HEVCDBK_OK : HEVCDBK_ERR_HIP;
    /* plane by plane; every plane is checked before the first launch */
    if (fused == HEVCDBK_FUSED_ON)
        for (unsigned i = 0; i < n_planes; i++)
            if (!dbk_deblock_sao_supports(da[i], sa[i], (int)planes[i].sample_bytes, planes[i].is_chroma != 0)) return HEVCDBK_ERR_UNSUPPORTED;
    for (unsigned i = 0; i < n_planes; i++)
        if (int rc = deblock_sao_plane(ctx, &planes[i], qp, tables, da[i], sa[i], fused, s)) return rc;
    return HEVCDBK_OK;
}

int hevc_deblock_sao_h265_device_planes(hevcdbk_context *ctx, const hevcdbk_device_planes *planes, unsigned n_planes, unsigned qp,
                                        const hevcdbk_h265_params *prm, const hevcdbk_sao_plane *sao, int fused, void *hip_stream)
{
    if (!ctx || !planes || !sao || n_planes == 0 || n_planes > 3 ||
        (fused != HEVCDBK_FUSED_AUTO && fused != HEVCDBK_FUSED_OFF && fused != HEVCDBK_FUSED_ON))
        return HEVCDBK_ERR_ARG;
    DbkH265Args h[3];
    DbkSaoArgs sa[3];
    bool can[3] = {false, false, false};
    bool one = n_planes >= 2 && fused != HEVCDBK_FUSED_OFF && !planes[0].is_chroma;
    for (unsigned i = 0; i < n_planes; i++) {
        if (int rc = sao_args(&planes[i], sao[i].params, sao[i].params_stride, sao[i].params_frame_stride, sao[i].ctb_log2, sao[i].keep,
                              sao[i].keep_stride, sao[i].keep_frame_stride, sa[i]))
            return rc;
        if (int rc = h265_args(&planes[i], (int)i, qp, prm, h[i])) return rc; /* c_idx = plane index: 0 Y, 1 Cb, 2 Cr */
        if (planes[i].n_frames != planes[0].n_frames) return HEVCDBK_ERR_ARG;
        can[i] = dbk_packed_h265_supports(h[i], (int)planes[i].sample_bytes, i != 0) &&
                 dbk_deblock_sao_supports(h[i].base, sa[i], (int)planes[i].sample_bytes, i != 0);
        one = one && (i == 0 || planes[i].is_chroma) && planes[i].sample_bytes == planes[0].sample_bytes &&
              planes[i].bit_depth == planes[0].bit_depth && (h[i].base.qp_map != nullptr) == (h[0].base.qp_map != nullptr) && can[i];
    }
    if (int rc = bind(ctx)) return rc;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->compute;
    if (one)
        return hip_ok(ctx, dbk_launch_deblock_sao_multi_h265(h, sa, (int)n_planes, (int)planes[0].sample_bytes, s), "fused deblocking + SAO launch")
                   ? HEVCDBK_OK : HEVCDBK_ERR_HIP;
    if (fused == HEVCDBK_FUSED_ON)
        for (unsigned i = 0; i < n_planes; i++)
            if (!can[i]) return HEVCDBK_ERR_UNSUPPORTED;
    for (unsigned i = 0; i < n_planes; i++)
        if (int rc = deblock_sao_plane_h265(ctx, &planes[i], (int)i, qp, prm, h[i], sa[i], fused, s)) return rc;
    return HEVCDBK_OK;
}

} /* extern "C" */
