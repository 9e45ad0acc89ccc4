/*
 * host_frames.c -- the reference-shaped call on frames in HOST memory, plain C99 on include/hevc_deblock.h: what main.cu:128-138
 * does with ReadYuvFrame + ExecuteGpu, for 3840x2160 4:2:0 frames and with the three choices a caller has:
 *   1. frames in ordinary memory (malloc): the library's crew of host threads writes them into HBM through the PCIe BAR and
 *      copies the results back; nothing to do for the caller;
 *   2. the same buffers registered once (hevcdbk_host_register): the kernels store their results into the caller's planes;
 *   3. a sequence of frames through hevc_deblocking_filter_sequence (three frames in flight).
 * Every output is compared with the first one's (the three ways must agree byte for byte), and the per-strip record of the last
 * single-frame call is printed.  Built by the CPU test-suite with `gcc -std=c99 -pedantic -Wall -Werror`; run on an MI355X:
 *
 *   gcc -std=c99 -Iinclude examples/host_frames.c -Lgpu_video_codec_amd -lhevcdbk -Wl,-rpath,$PWD/gpu_video_codec_amd -o host_frames
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

enum { W = 3840, H = 2160, NSEQ = 6 };

/* blocky content: 8x8 blocks of a seeded level plus a little texture, so that the filter has edges to work on */
static void fill(unsigned char *p, unsigned w, unsigned h, unsigned seed)
{
    unsigned x, y, s = seed * 2654435761u + 12345u;
    for (y = 0; y < h; y++)
        for (x = 0; x < w; x++) {
            unsigned b = ((y >> 3) * 131u + (x >> 3) * 71u + seed * 977u) * 2246822519u;
            s = s * 1664525u + 1013904223u;
            p[(size_t)y * w + x] = (unsigned char)(96u + ((b >> 27) << 1) + ((s >> 30) & 1u));
        }
}

static void frame_of(hevcdbk_frame *f, unsigned char *base)
{
    memset(f, 0, sizeof *f);
    f->width = W; f->height = H; f->bit_depth = 8; f->sample_bytes = 1;
    f->plane[0] = base; f->pitch[0] = W;
    f->plane[1] = base + (size_t)W * H; f->pitch[1] = W / 2;
    f->plane[2] = base + (size_t)W * H + (size_t)(W / 2) * (H / 2); f->pitch[2] = W / 2;
}

int main(void)
{
    const size_t fb = (size_t)W * H * 3 / 2;
    hevcdbk_context *ctx = NULL;
    hevcdbk_qp qp;
    hevcdbk_timing t;
    hevcdbk_frame fr, seq[NSEQ];
    hevcdbk_strip_trace tr[64];
    unsigned char *src, *a, *b, *pool;
    unsigned i, n = 0;

    if (hevcdbk_create(0, &ctx) != HEVCDBK_OK) {
        fprintf(stderr, "no HIP device: this library has no CPU path\n");
        return 2;
    }
    memset(&qp, 0, sizeof qp);
    qp.qp = 35; qp.ctu_log2 = 6;
    src = malloc(fb); a = malloc(fb); b = malloc(fb); pool = malloc(fb * NSEQ);
    if (!src || !a || !b || !pool) return 3;
    fill(src, W, H, 1); fill(src + (size_t)W * H, W / 2, H / 2, 2); fill(src + (size_t)W * H * 5 / 4, W / 2, H / 2, 3);

    /* 1. ordinary memory */
    CHECK(hevcdbk_set_host_threads(ctx, 4));
    memcpy(a, src, fb);
    frame_of(&fr, a);
    CHECK(hevc_deblocking_filter(ctx, &fr, NULL, &qp, NULL, &t));
    printf("malloc'ed 4:2:0 4K frame: %.3f ms per call (exec %.3f + copy %.3f ms summed over strips), %u copying threads\n",
           t.pipelined_s * 1e3, t.exec_s * 1e3, t.copy_s * 1e3, hevcdbk_get_host_threads(ctx));
    CHECK(hevcdbk_last_frame_trace(ctx, tr, 64, &n));
    for (i = 0; i < n && i < 64; i++)
        printf("  strip %2u plane %d rows %4u..%4u: in %6.1f..%6.1f us, launched %6.1f, kernel ended (seen) %6.1f, out ..%6.1f us, kernel %.1f us\n", i,
               tr[i].plane, tr[i].row_begin, tr[i].row_end, tr[i].stage_begin_s * 1e6, tr[i].stage_end_s * 1e6, tr[i].enqueue_end_s * 1e6,
               tr[i].d2h_seen_s * 1e6, tr[i].unstage_end_s * 1e6, tr[i].kernel_ms * 1e3);

    /* 2. the same, with the buffer registered once (a decoder's picture pool) */
    CHECK(hevcdbk_host_register(ctx, b, fb));
    memcpy(b, src, fb);
    frame_of(&fr, b);
    CHECK(hevc_deblocking_filter(ctx, &fr, NULL, &qp, NULL, &t));
    printf("registered buffer:        %.3f ms per call\n", t.pipelined_s * 1e3);
    CHECK(hevcdbk_host_unregister(ctx, b));
    if (memcmp(a, b, fb) != 0) { fprintf(stderr, "registered result differs\n"); return 4; }

    /* 3. a sequence from ordinary memory */
    for (i = 0; i < NSEQ; i++) {
        memcpy(pool + i * fb, src, fb);
        frame_of(&seq[i], pool + i * fb);
    }
    CHECK(hevc_deblocking_filter_sequence(ctx, seq, NSEQ, NULL, &qp, NULL, &t));
    printf("sequence of %d frames:     %.3f ms per frame\n", (int)NSEQ, t.pipelined_s * 1e3 / NSEQ);
    for (i = 0; i < NSEQ; i++)
        if (memcmp(a, pool + i * fb, fb) != 0) { fprintf(stderr, "sequence frame %u differs\n", i); return 5; }
    printf("all three ways agree byte for byte\n");
    free(src); free(a); free(b); free(pool);
    hevcdbk_destroy(ctx);
    return 0;
}
