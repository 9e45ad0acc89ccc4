/*
 * decoder_loop.c -- plain-C sketch of a decoder's in-loop stage on top of include/hevc_deblock.h: per picture, derive bS
 * from the prediction data on the GPU (H.265 8.7.2.4), deblock + SAO the luma plane into the output picture in one kernel
 * (8.7.2 + 8.7.3), deblock Cb / Cr in place.  Everything stays in HBM; the caller owns all buffers.  Built by the CPU test-suite with
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

    /* picture planes, the luma output plane, prediction data, bS arrays, SAO parameters: all in HBM */
    void *y, *cb, *cr, *y_out, *flags, *mv0, *mv1, *ref0, *ref1, *vbs, *hbs, *cvbs, *chbs, *sao;
    CHECK(hevcdbk_device_malloc(ctx, (size_t)W * H, &y));
    CHECK(hevcdbk_device_malloc(ctx, (size_t)CW * CH, &cb));
    CHECK(hevcdbk_device_malloc(ctx, (size_t)CW * CH, &cr));
    CHECK(hevcdbk_device_malloc(ctx, (size_t)W * H, &y_out));
    CHECK(hevcdbk_device_malloc(ctx, units * 2, &flags));
    CHECK(hevcdbk_device_malloc(ctx, units * 4, &mv0));
    CHECK(hevcdbk_device_malloc(ctx, units * 4, &mv1));
    CHECK(hevcdbk_device_malloc(ctx, units * 4, &ref0));
    CHECK(hevcdbk_device_malloc(ctx, units * 4, &ref1));
    CHECK(hevcdbk_device_malloc(ctx, hevcdbk_h265_num_vert_bs(W, H), &vbs));
    CHECK(hevcdbk_device_malloc(ctx, hevcdbk_h265_num_hor_bs(W, H), &hbs));
    CHECK(hevcdbk_device_malloc(ctx, hevcdbk_h265_num_vert_bs(CW, CH), &cvbs));
    CHECK(hevcdbk_device_malloc(ctx, hevcdbk_h265_num_hor_bs(CW, CH), &chbs));
    const unsigned ctbs_x = (W + 63) / 64, ctbs_y = (H + 63) / 64;
    CHECK(hevcdbk_device_malloc(ctx, sizeof(hevcdbk_sao_ctb) * ctbs_x * ctbs_y, &sao));
    /* a decoder's reconstruction kernels would have written all of these; zero them so the sketch runs */
    CHECK(hevcdbk_memset_d(ctx, y, 128, (size_t)W * H));
    CHECK(hevcdbk_memset_d(ctx, cb, 128, (size_t)CW * CH));
    CHECK(hevcdbk_memset_d(ctx, cr, 128, (size_t)CW * CH));
    CHECK(hevcdbk_memset_d(ctx, flags, 0, units * 2));
    CHECK(hevcdbk_memset_d(ctx, mv0, 0, units * 4));
    CHECK(hevcdbk_memset_d(ctx, mv1, 0, units * 4));
    CHECK(hevcdbk_memset_d(ctx, ref0, 0, units * 4));
    CHECK(hevcdbk_memset_d(ctx, ref1, 0, units * 4));
    CHECK(hevcdbk_memset_d(ctx, sao, 0, sizeof(hevcdbk_sao_ctb) * ctbs_x * ctbs_y));

    /* 8.7.2.4 */
    hevcdbk_h265_units u;
    u.flags = (const uint16_t *)flags; u.mv0 = (const int16_t *)mv0; u.mv1 = (const int16_t *)mv1;
    u.ref0 = (const int32_t *)ref0; u.ref1 = (const int32_t *)ref1;
    CHECK(hevcdbk_h265_derive_bs_device(ctx, &u, W, H, (uint8_t *)vbs, (uint8_t *)hbs, (uint8_t *)cvbs, (uint8_t *)chbs, NULL));

    /* 8.7.2 + 8.7.3, luma: reconstruction -> output picture in ONE kernel (deblocked samples go from the first stage to the
     * second through LDS; the deblocked picture never exists in memory) */
    hevcdbk_h265_params prm;
    memset(&prm, 0, sizeof(prm));
    hevcdbk_device_planes p;
    memset(&p, 0, sizeof(p));
    p.n_frames = 1; p.bit_depth = 8; p.sample_bytes = 1;
    p.src = y; p.dst = y_out; p.pitch = W; p.frame_stride = (size_t)W * H; p.plane_w = W; p.plane_h = H;
    p.vert_bs = (const uint8_t *)vbs; p.hor_bs = (const uint8_t *)hbs;
    CHECK(hevc_deblock_sao_h265_device(ctx, &p, 0, 32, &prm, (const hevcdbk_sao_ctb *)sao, ctbs_x, 0, 6, NULL, 0, 0,
                                       HEVCDBK_FUSED_AUTO, NULL));
    /* 8.7.2, chroma: in place (a decoder with chroma SAO enabled would use the same fused entry with 32-sample CTBs) */
    p.is_chroma = 1; p.pitch = CW; p.frame_stride = (size_t)CW * CH; p.plane_w = CW; p.plane_h = CH;
    p.vert_bs = (const uint8_t *)cvbs; p.hor_bs = (const uint8_t *)chbs;
    p.src = p.dst = cb;
    CHECK(hevc_deblocking_filter_h265_device(ctx, &p, 1, 32, &prm, HEVCDBK_KERNEL_AUTO, NULL));
    p.src = p.dst = cr;
    CHECK(hevc_deblocking_filter_h265_device(ctx, &p, 2, 32, &prm, HEVCDBK_KERNEL_AUTO, NULL));
    CHECK(hevcdbk_synchronize(ctx));
    printf("one %ux%u picture through bS derivation, deblocking and SAO on the GPU\n", W, H);

    hevcdbk_device_free(ctx, y); hevcdbk_device_free(ctx, cb); hevcdbk_device_free(ctx, cr); hevcdbk_device_free(ctx, y_out);
    hevcdbk_device_free(ctx, flags); hevcdbk_device_free(ctx, mv0); hevcdbk_device_free(ctx, mv1);
    hevcdbk_device_free(ctx, ref0); hevcdbk_device_free(ctx, ref1);
    hevcdbk_device_free(ctx, vbs); hevcdbk_device_free(ctx, hbs); hevcdbk_device_free(ctx, cvbs); hevcdbk_device_free(ctx, chbs);
    hevcdbk_device_free(ctx, sao);
    hevcdbk_destroy(ctx);
    return 0;
}
