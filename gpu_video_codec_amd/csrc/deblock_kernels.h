/* deblock_kernels.h -- launch interface between the C-ABI host driver and the HIP kernels. */
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

struct DbkArgs {
    const uint8_t *src;
    uint8_t *dst;
    long long pitch;        /* bytes */
    long long frame_stride; /* bytes */
    int plane_w, plane_h;   /* samples */
    int nbx, nby;           /* offset blocks: plane_w/8+1, plane_h/8+1 (cpu.h:141-142) */
    int n_frames;
    const uint8_t *vert_bs, *hor_bs;
    long long vert_bs_stride, hor_bs_stride; /* bytes per frame, 0 = shared */
    int n_vert, n_hor;
    int vstride, hstride;   /* plane_w/8+1 (cpu.h:161), plane_w/8 (cpu.h:289) */
    int limit_bx, limit_by; /* guards of hor2 / ver2 (luma limits for chroma) */
    int tc, beta;           /* scalar-QP values, already << (bit_depth-8) */
    int max_v, shift;
    const uint8_t *qp_map;
    int map_stride, ctu_log2;
    long long map_frame_stride;
    uint8_t tc_tab[52], beta_tab[52];
    /* row-major block numbering of the packed kernels (filled by dbk_launch_packed) */
    int nb_total, wpf;              /* blocks per frame, workgroups per frame */
    uint32_t magic_wpf, magic_nbx;  /* floor(2^32/d)+1 reciprocals */
    int xcd_swizzle;                /* renumber workgroups so each XCD gets a contiguous range */
    int map_override;               /* block -> lane map of the packed kernels: 0 = chosen from the geometry, 1 = one
                                       workgroup per block row, 2 = row-major numbering (HEVCDBK_MAP_* of the C ABI); 3 = stripes, 4 = LDS tiles:
                                       diagnostic build only (HEVCDBK_DIAG_MAP_STRIPE / _TILES) */
    int n_cus;                      /* compute units of the device (0 = unknown: 256) */
#ifdef HEVCDBK_DIAG /* the diagnostic build only (libhevcdbk_diag.so, hevcdbk_diag.h): never in the product library */
    /* tile map of the 8-bit luma kernel (dbk_tile_kernel; filled by dbk_launch_packed): a workgroup stages k whole block
     * rows in LDS with naturally aligned 16-byte-per-lane accesses and filters the blocks bx = 1..nbx-1 out of LDS */
    int tl_k;                       /* block rows per workgroup */
    int tl_M;                       /* blocks per row handled: nbx - 1 */
    int tl_nw;                      /* waves per workgroup: ceil(k*M / 64) */
    int tl_cw;                      /* 16-byte pieces per pixel row: plane_w / 16 */
    int tl_ndma;                    /* 1-KiB transfers per tile: 8*k*plane_w / 1024 */
    uint32_t tl_magic_M, tl_magic_cw;
    /* stripe map of the 8-bit luma kernel (dbk_stripe_kernel; filled by dbk_launch_packed): the blocks bx = 1..nbx-1 of
     * the block rows 1..nby-2 are dealt to persistent one-wave workgroups that walk down the frames, the frame border
     * (row 0, row nby-1, column 0) goes to extra workgroups of the same launch */
    int st_M;                       /* blocks per row handled by the stripes: nbx - 1 (even) */
    int st_k;                       /* block rows per row group */
    int st_W;                       /* wave slots per row group: ceil(k*M / 64) */
    int st_Q;                       /* persistent workgroups (one per stripe) */
    int st_groups;                  /* row groups per frame: ceil((nby - 2) / k) */
    int st_items;                   /* n_frames * groups: one item = one row group of one frame */
    int st_border_wgs;              /* workgroups (W waves each) at the START of the grid that run the frame border */
    int st_lds_bytes;               /* dynamic LDS of a workgroup: W * (4096 + 256) */
    int st_border_wpf;              /* border waves per frame: ceil((2*nbx + nby - 2) / 64) */
    uint32_t st_magic_M, st_magic_groups, st_magic_bwpf; /* floor(2^32/d)+1 reciprocals */
    int use_queue;   /* 8-bit luma: strong segments scheduled through the workgroup's LDS queue */
    int diag_ablate; /* 1 = strong segments filtered as normal, 2 = normal filter skipped, 4 = barriers (WRONG pixels) */
    int diag_xshift; /* copy mode only: byte shift of every row span (alignment experiments) */
    int diag_prio;   /* bit 0: s_setprio 3 from wave start until the row loads are issued; bit 1: from the final pack on */
    int diag_dummy;  /* N extra VALU instructions per wave (sensitivity of the kernel time to VALU work) */
#endif
    /* launch only the block rows by_begin .. by_begin + by_count - 1 (by_count 0 = all): a frame's block rows are
     * independent, so a host pipeline can filter a frame strip by strip while the other strips are still on the bus */
    int by_begin, by_count;
};

/* the next launch made from this host thread through any dbk_launch_* of deblock_kernels.hip records the kernel's own begin
 * into `start` and its end into `stop` (either may be NULL); one-shot */
void dbk_set_next_launch_events(hipEvent_t start, hipEvent_t stop);

/* one lane per offset block, 32-bit arithmetic; every operand kind */
hipError_t dbk_launch_generic(const DbkArgs &a, int sample_bytes, bool chroma, hipStream_t stream);
/* packed int16 arithmetic kernels: scalar QP; 8-bit samples (luma or chroma) and 16-bit containers (luma) */
/* mode 0 = filter; 1 = diagnostic copy (same memory accesses, no arithmetic) exists in the HEVCDBK_DIAG build only */
hipError_t dbk_launch_packed(const DbkArgs &a, int sample_bytes, bool chroma, int mode, hipStream_t stream);
bool dbk_packed_supports(const DbkArgs &a, int sample_bytes, bool chroma);

/* the planes of one 4:2:0 frame (or batch of frames) in one launch: all 8-bit or all 16-bit containers (<= 12 bit), scalar QP,
 * plane 0 = luma */
struct DbkMultiArgs {
    DbkArgs p[3];
    int row_end[3]; /* cumulative block-row counts: blockIdx.x < row_end[i] belongs to plane <= i */
};
bool dbk_multi_supports(const DbkArgs *planes, int n, const int *sample_bytes);
hipError_t dbk_launch_packed_multi(const DbkArgs *planes, int n, int sample_bytes, hipStream_t stream);

#ifdef HEVCDBK_DIAG
/* knobs of the diagnostic build, set through hevcdbk_diag_set() (hevcdbk_diag.h); the product library has none of this and
 * reads no environment variable */
struct DbkDiag {
    int wg_cap = 512; /* workgroup width cap of the packed kernels (64..1024) */
    int noswz = 0;     /* row-major map without the per-XCD renumbering */
    int nofuse = 0;    /* no fused Y+U+V launch */
    int dmacopy = 0;   /* small frames through DMA copies instead of host-direct kernels */
    int ablate = 0;    /* 1 nostrong, 2 nonormal, 4 barriers: WRONG pixels, timing only */
    int queue = 0;     /* the LDS-queue kernel for 8-bit luma */
    int align = 0;     /* copy mode: shift the row spans onto their natural alignment */
    int prio = 0;      /* wave priority experiment, see DbkArgs::diag_prio */
    int dummy = 0;     /* extra VALU instructions per wave */
    int mode3 = 0;     /* run the instrumented (MODE 3) instantiation even with no ablation set: the A/B baseline */
    int rows = 0;      /* pipe map: block rows per workgroup (default 4) */
    int lds = 0;       /* dynamic LDS bytes per workgroup of the plain packed kernels: an occupancy limiter for A/B runs */
    int xpad = 0;      /* row-major map: N padding workgroups per XCD range (the XCDs' fronts move out of step with each other) */
};
extern DbkDiag g_dbk_diag;
#endif

/* ---- spec-exact mode (H.265 clause 8.7.2), deblock_h265.hip ---- */
struct DbkH265Args {
    DbkArgs base;    /* geometry, planes, bS arrays (4-sample granular: vstride = plane_w/8+1, hstride = plane_w/4), QP map */
    int qp;          /* scalar QpY when base.qp_map == NULL */
    int tc_off;      /* slice_tc_offset_div2 << 1 */
    int beta_off;    /* slice_beta_offset_div2 << 1 */
    int c_qp_offset; /* cQpPicOffset of a chroma plane */
    /* scalar-QP operands, filled by the launcher: beta, and tc for bS 1 / bS 2 (chroma: bS 2 through QpC) */
    int beta_s, tc_bs1, tc_bs2;
};
hipError_t dbk_launch_h265(const DbkH265Args &h, int sample_bytes, bool chroma, hipStream_t stream); /* 32-bit arithmetic, every operand kind */
/* packed-int16 arithmetic on the reference-mode kernels' memory path: 8-bit samples; 16-bit containers up to 11 bit
 * (luma) / 12 bit (chroma) */
bool dbk_packed_h265_supports(const DbkH265Args &h, int sample_bytes, bool chroma);
hipError_t dbk_launch_packed_h265(const DbkH265Args &h, int sample_bytes, bool chroma, hipStream_t stream);
/* 8.7.2.4 on per-4x4-unit arrays in device memory; cvert / chor (4:2:0 chroma arrays) may be NULL */
hipError_t dbk_launch_h265_bs(const void *flags, const void *mv0, const void *mv1, const void *ref0, const void *ref1, int w,
                              int h, uint8_t *vert, uint8_t *hor, uint8_t *cvert, uint8_t *chor, hipStream_t stream);
hipError_t dbk_launch_h265_chroma_bs(const uint8_t *vert, const uint8_t *hor, int w, int h, uint8_t *cvert, uint8_t *chor,
                                     hipStream_t stream);

/* ---- sample adaptive offset (H.265 clause 8.7.3), sao.hip ---- */
struct DbkSaoCtb {
    uint8_t type;     /* 0 off, 1 band, 2 edge */
    uint8_t cls;      /* band position / edge class */
    int8_t offset[4]; /* SaoOffsetVal[1..4] */
};
struct DbkSaoArgs {
    const uint8_t *src;
    uint8_t *dst;
    long long pitch, frame_stride;
    int plane_w, plane_h, n_frames;
    int max_v, band_shift; /* (1 << bit_depth) - 1, bit_depth - 5 */
    const DbkSaoCtb *params;
    int params_stride;
    long long params_frame_stride; /* entries */
    int ctb_log2;                  /* CTB size of this plane, in samples */
    const uint8_t *keep;           /* per 8x8 samples of this plane, may be NULL */
    int keep_stride;
    long long keep_frame_stride;
};
hipError_t dbk_launch_sao(const DbkSaoArgs &a, int sample_bytes, hipStream_t stream);

/* ---- deblocking + SAO in one kernel (deblock_sao_fused.inc): 8-bit planes, scalar QP; d.src -> s.dst, d.dst and s.src unused ---- */
/* tile numbering of the fused kernel (filled by its launchers): a 1-D grid, workgroups renumbered so that each XCD gets a
 * contiguous range of tiles (row-major inside a frame: tiles that share cache lines and rim rows meet in one L2) */
struct DbkFusedGrid {
    uint32_t tiles_x, tiles_per_frame, total; /* total = tiles_per_frame * n_frames */
    uint32_t magic_tpf, magic_tx;             /* floor(2^32/d)+1 reciprocals */
    uint32_t per_xcd;                         /* grid size / 8 */
};
struct DbkFusedArgs {
    DbkArgs d;
    DbkSaoArgs s;
    DbkFusedGrid g;
};
struct DbkFusedH265Args {
    DbkH265Args d;
    DbkSaoArgs s;
    DbkFusedGrid g;
};
/* the planes of a 4:2:0 batch in one fused launch: the grid is the planes' grids one after the other */
struct DbkFusedMultiArgs {
    DbkFusedArgs pl[3];
    uint32_t wg_end[3]; /* cumulative grid sizes: blockIdx.x < wg_end[i] belongs to plane <= i */
};
struct DbkFusedMultiH265Args {
    DbkFusedH265Args pl[3];
    uint32_t wg_end[3];
};
/* 8-bit planes and 16-bit containers up to 12 bit, scalar QP */
bool dbk_deblock_sao_supports(const DbkArgs &d, const DbkSaoArgs &s, int sample_bytes, bool chroma);
hipError_t dbk_launch_deblock_sao(const DbkArgs &d, const DbkSaoArgs &s, int sample_bytes, bool chroma, hipStream_t stream);
hipError_t dbk_launch_deblock_sao_h265(const DbkH265Args &h, const DbkSaoArgs &s, int sample_bytes, bool chroma, hipStream_t stream);
/* n = 2 or 3 planes (plane 0 luma, the others chroma), every one accepted by dbk_deblock_sao_supports, one sample width
 * and bit depth, one frame count */
hipError_t dbk_launch_deblock_sao_multi(const DbkArgs *d, const DbkSaoArgs *s, int n, int sample_bytes, hipStream_t stream);
hipError_t dbk_launch_deblock_sao_multi_h265(const DbkH265Args *h, const DbkSaoArgs *s, int n, int sample_bytes, hipStream_t stream);

