/*
 * decoder_loop.c -- plain-C sketch of a decoder's in-loop stage on top of include/hevc_deblock.h: per picture, derive bS
 * from the prediction data on the GPU (H.265 8.7.2.4), then deblock + SAO all three planes of the picture into the output
 * picture in ONE launch (8.7.2 + 8.7.3: hevc_deblock_sao_h265_device_planes).  Everything stays in HBM; the caller owns all
 * buffers.  Built by the CPU test-suite with
 * `gcc -std=c99 -pedantic -Wall -Werror` to prove that the header is a C header; run it on a machine with an MI355X:
 *
 *   gcc -std=c99 -Iinclude examples/decoder_loop.c -Lgpu_video_codec_amd -lhevcdbk -Wl,-rpath,$PWD/gpu_video_codec_amd -o decoder_loop
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hevc_deblock.h"

#define CHECK(call)                                                                                      \
    do {                                                                                                 \
        int rc_ = (call);                                                                                \
        if (rc_ != HEVCDBK_OK) {                                                                         \
            fprintf(stderr, "%s -> %s (%s)\n", #call, hevcdbk_strerror(rc_), hevcdbk_last_error(ctx));  \
            return 1;                                                                                    \
        }                                                                                                \
    } while (0)

int main(void)
{
    const unsigned W = 1920, H = 1088, CW = W / 2, CH = H / 2;
    const size_t units = (size_t)(W / 4) * (H / 4);
    hevcdbk_context *ctx = NULL;
    if (hevcdbk_create(0, &ctx) != HEVCDBK_OK) {
        fprintf(stderr, "no HIP device: this library has no CPU path\n");
        return 2;
    }

    /* picture planes, the output picture's planes, prediction data, bS arrays, SAO parameters: all in HBM */
    void *y, *cb, *cr, *y_out, *cb_out, *cr_out, *flags, *mv0, *mv1, *ref0, *ref1, *vbs, *hbs, *cvbs, *chbs, *sao, *sao_cb, *sao_cr, *qpy;
    CHECK(hevcdbk_device_malloc(ctx, (size_t)W * H, &y));
    CHECK(hevcdbk_device_malloc(ctx, (size_t)CW * CH, &cb));
    CHECK(hevcdbk_device_malloc(ctx, (size_t)CW * CH, &cr));
    CHECK(hevcdbk_device_malloc(ctx, (size_t)W * H, &y_out));
    CHECK(hevcdbk_device_malloc(ctx, (size_t)CW * CH, &cb_out));
    CHECK(hevcdbk_device_malloc(ctx, (size_t)CW * CH, &cr_out));
    CHECK(hevcdbk_device_malloc(ctx, units * 2, &flags));
    CHECK(hevcdbk_device_malloc(ctx, units * 4, &mv0));
    CHECK(hevcdbk_device_malloc(ctx, units * 4, &mv1));
    CHECK(hevcdbk_device_malloc(ctx, units * 4, &ref0));
    CHECK(hevcdbk_device_malloc(ctx, units * 4, &ref1));
    CHECK(hevcdbk_device_malloc(ctx, hevcdbk_h265_num_vert_bs(W, H), &vbs));
    CHECK(hevcdbk_device_malloc(ctx, hevcdbk_h265_num_hor_bs(W, H), &hbs));
    CHECK(hevcdbk_device_malloc(ctx, hevcdbk_h265_num_vert_bs(CW, CH), &cvbs));
    CHECK(hevcdbk_device_malloc(ctx, hevcdbk_h265_num_hor_bs(CW, CH), &chbs));
    /* one SAO entry per CTB and plane: 64-sample luma CTBs are 32-sample CTBs of the 4:2:0 chroma planes, same grid */
    const unsigned ctbs_x = (W + 63) / 64, ctbs_y = (H + 63) / 64;
    const size_t sao_bytes = sizeof(hevcdbk_sao_ctb) * ctbs_x * ctbs_y;
    CHECK(hevcdbk_device_malloc(ctx, sao_bytes, &sao));
    CHECK(hevcdbk_device_malloc(ctx, sao_bytes, &sao_cb));
    CHECK(hevcdbk_device_malloc(ctx, sao_bytes, &sao_cr));
    /* QpY per 16x16 quantization group (cu_qp_delta): one map in luma units, read by all three planes */
    const unsigned qg_x = (W + 15) / 16, qg_y = (H + 15) / 16;
    CHECK(hevcdbk_device_malloc(ctx, (size_t)qg_x * qg_y, &qpy));
    /* a decoder's reconstruction kernels would have written all of these; fill them so the sketch runs */
    CHECK(hevcdbk_memset_d(ctx, y, 128, (size_t)W * H));
    CHECK(hevcdbk_memset_d(ctx, cb, 128, (size_t)CW * CH));
    CHECK(hevcdbk_memset_d(ctx, cr, 128, (size_t)CW * CH));
    CHECK(hevcdbk_memset_d(ctx, flags, 0, units * 2));
    CHECK(hevcdbk_memset_d(ctx, mv0, 0, units * 4));
    CHECK(hevcdbk_memset_d(ctx, mv1, 0, units * 4));
    CHECK(hevcdbk_memset_d(ctx, ref0, 0, units * 4));
    CHECK(hevcdbk_memset_d(ctx, ref1, 0, units * 4));
    CHECK(hevcdbk_memset_d(ctx, sao, 0, sao_bytes));
    CHECK(hevcdbk_memset_d(ctx, sao_cb, 0, sao_bytes));
    CHECK(hevcdbk_memset_d(ctx, sao_cr, 0, sao_bytes));
    CHECK(hevcdbk_memset_d(ctx, qpy, 32, (size_t)qg_x * qg_y));

    /* 8.7.2.4 */
    hevcdbk_h265_units u;
    u.flags = (const uint16_t *)flags; u.mv0 = (const int16_t *)mv0; u.mv1 = (const int16_t *)mv1;
    u.ref0 = (const int32_t *)ref0; u.ref1 = (const int32_t *)ref1;
    CHECK(hevcdbk_h265_derive_bs_device(ctx, &u, W, H, (uint8_t *)vbs, (uint8_t *)hbs, (uint8_t *)cvbs, (uint8_t *)chbs, NULL));

    /* 8.7.2 + 8.7.3 of Y, Cb, Cr: reconstruction -> output picture in ONE launch (a workgroup deblocks a tile into LDS and
     * applies SAO from there: the deblocked picture never exists in memory; the planes' tiles follow each other in the grid) */
    hevcdbk_h265_params prm;
    memset(&prm, 0, sizeof(prm));
    hevcdbk_device_planes p[3];
    hevcdbk_sao_plane so[3];
    memset(p, 0, sizeof(p));
    memset(so, 0, sizeof(so));
    void *const src[3] = {y, cb, cr}, *const dst[3] = {y_out, cb_out, cr_out}, *const sp[3] = {sao, sao_cb, sao_cr};
    int i;
    for (i = 0; i < 3; i++) {
        const unsigned pw = i ? CW : W, ph = i ? CH : H;
        p[i].n_frames = 1; p[i].bit_depth = 8; p[i].sample_bytes = 1; p[i].is_chroma = i != 0;
        p[i].src = src[i]; p[i].dst = dst[i]; p[i].pitch = pw; p[i].frame_stride = (size_t)pw * ph; p[i].plane_w = pw; p[i].plane_h = ph;
        p[i].vert_bs = (const uint8_t *)(i ? cvbs : vbs); p[i].hor_bs = (const uint8_t *)(i ? chbs : hbs);
        p[i].qp_map = (const uint8_t *)qpy; p[i].qp_map_stride = qg_x; p[i].ctu_log2 = 4; /* the map unit: 16 luma samples */
        so[i].params = (const hevcdbk_sao_ctb *)sp[i]; so[i].params_stride = ctbs_x; so[i].ctb_log2 = i ? 5 : 6;
    }
    CHECK(hevc_deblock_sao_h265_device_planes(ctx, p, 3, /* qp: unused with a map */ 0, &prm, so, HEVCDBK_FUSED_AUTO, NULL));
    /* a picture whose chroma SAO is switched off deblocks Cb / Cr in place instead:
     *   hevc_deblocking_filter_h265_device(ctx, &chroma_plane, c_idx, qp, &prm, HEVCDBK_KERNEL_AUTO, NULL) */
    CHECK(hevcdbk_synchronize(ctx));
    printf("one %ux%u picture through bS derivation, deblocking and SAO on the GPU\n", W, H);

    hevcdbk_device_free(ctx, y); hevcdbk_device_free(ctx, cb); hevcdbk_device_free(ctx, cr); hevcdbk_device_free(ctx, y_out);
    hevcdbk_device_free(ctx, cb_out); hevcdbk_device_free(ctx, cr_out); hevcdbk_device_free(ctx, sao_cb); hevcdbk_device_free(ctx, sao_cr);
    hevcdbk_device_free(ctx, flags); hevcdbk_device_free(ctx, mv0); hevcdbk_device_free(ctx, mv1);
    hevcdbk_device_free(ctx, ref0); hevcdbk_device_free(ctx, ref1);
    hevcdbk_device_free(ctx, vbs); hevcdbk_device_free(ctx, hbs); hevcdbk_device_free(ctx, cvbs); hevcdbk_device_free(ctx, chbs);
    hevcdbk_device_free(ctx, sao); hevcdbk_device_free(ctx, qpy);
    hevcdbk_destroy(ctx);
    return 0;
}
