/*
 * hevc_deblock.h -- C ABI of the MI355X-native HEVC in-loop deblocking filter.
 *
 * This is the drop-in boundary for the one hot path of RomanKazantsev/gpu_video_codec:
 * the per-frame deblocking filter that main.cu drives through ExecuteCpu / ExecuteGpu.
 * Every entry point below names the reference interface it replaces.  Citations:
 *   cpu.h  = hevc_deblocking_filter/hevc_deblocking_filter_cpu.h
 *   gpu.cu = hevc_deblocking_filter/hevc_deblocking_filter_gpu.cu
 *   main.cu = hevc_deblocking_filter/main.cu
 *
 * The implementation behind this header is hand-written HIP for gfx950 only.  There is no
 * CPU fallback: every compute entry point returns HEVCDBK_ERR_HIP when no HIP device is usable.
 *
 * Plain C: pointers and sizes only, no C++/torch types.  Results are bit-exact with the
 * reference's single-thread CPU path (cpu.h:134-993) including its quirks (SURVEY.md 8a Q-list):
 * frames are UN-padded W x H planes here, the reference's 4-sample zero padding (cpu.h:55-71)
 * is implicit.
 */
#ifndef HEVC_DEBLOCK_H
#define HEVC_DEBLOCK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with hidden visibility: only the entry points below leave it */
#if defined(__GNUC__) || defined(__clang__)
#define HEVCDBK_API __attribute__((visibility("default")))
#else
#define HEVCDBK_API
#endif

/* ---- error codes: the reference throws `const char *` at three sites; the ABI returns ---- */
#define HEVCDBK_OK               0
#define HEVCDBK_ERR_FILE_SIZE   (-1) /* cpu.h:43-45  / gpu.cu:1082-1084 "Incorrect file size" */
#define HEVCDBK_ERR_DIMENSIONS  (-2) /* cpu.h:46-48  / gpu.cu:1085-1087 "Width and height of image must be multiplier of sample block size" */
#define HEVCDBK_ERR_BS_SIZE     (-3) /* cpu.h:122-123 "Incorrect size of input boundary strenght array" */
#define HEVCDBK_ERR_HIP         (-4) /* a HIP call failed or no device (reference prints and continues, gpu.cu:1271-1288) */
#define HEVCDBK_ERR_ARG         (-5)
#define HEVCDBK_ERR_NOMEM       (-6)
#define HEVCDBK_ERR_IO          (-7)
#define HEVCDBK_ERR_UNSUPPORTED (-8) /* alignment / bit depth outside what the kernels handle */

HEVCDBK_API const char *hevcdbk_strerror(int code);

/* ---- context: replaces the file-scope globals of gpu.cu:37-77 (one per device, re-entrant) ---- */
typedef struct hevcdbk_context hevcdbk_context;

HEVCDBK_API int hevcdbk_device_count(void);
/* binds to HIP device `device`, creates the compute stream and the two copy side streams */
HEVCDBK_API int hevcdbk_create(int device, hevcdbk_context **ctx);
HEVCDBK_API void hevcdbk_destroy(hevcdbk_context *ctx);
/* text of the last HIP failure seen by this context ("" if none) */
HEVCDBK_API const char *hevcdbk_last_error(const hevcdbk_context *ctx);
/* GetGpuDeviceInfo() equivalent (main.cu:92-107): fills name, CU count, memory */
typedef struct {
    char name[256];
    char gcn_arch[64];
    int compute_units;
    int wavefront_size;
    int max_threads_per_block;
    size_t total_global_mem;
    size_t shared_mem_per_block;
    size_t total_const_mem;
} hevcdbk_device_info;
HEVCDBK_API int hevcdbk_get_device_info(const hevcdbk_context *ctx, hevcdbk_device_info *info);

/* ---- tables and bS helpers ---- */
/* cpu.h:1021-1033 (copies at gpu.cu:80-85, 92-97) */
HEVCDBK_API const unsigned *hevcdbk_default_tc_table(void);   /* 52 entries */
HEVCDBK_API const unsigned *hevcdbk_default_beta_table(void); /* 52 entries */
/* cpu.h:86-87 / 104-105: element counts of the bS arrays of a plane_w x plane_h plane */
HEVCDBK_API size_t hevcdbk_num_vert_bs(unsigned plane_w, unsigned plane_h);
HEVCDBK_API size_t hevcdbk_num_hor_bs(unsigned plane_w, unsigned plane_h);
/* cpu.h:92-99 / 110-117 (gpu.cu:1136-1180): the default "all intra" pattern, quirks included */
HEVCDBK_API int hevcdbk_default_bs(unsigned plane_w, unsigned plane_h, uint8_t *vert_bs, uint8_t *hor_bs);

/* ---- operands of hevc_deblocking_filter(frame, bS, QP, tc/beta tables) ---- */

/* frame: replaces ReadYuvFrame's planes (cpu.h:1041-1048) / the pinned planes of gpu.cu:37-45.
 * Un-padded planar 4:2:0; U/V may be NULL for a luma-only call.  Caller owns the memory. */
typedef struct {
    unsigned width, height; /* luma samples, multiples of 8 (cpu.h:46-48); with chroma: multiples of 16 */
    unsigned bit_depth;     /* 8..16; the reference is 8 (cpu.h:1202) */
    unsigned sample_bytes;  /* 1 (bit_depth 8 only) or 2 (little-endian containers) */
    void *plane[3];         /* Y, U, V */
    size_t pitch[3];        /* bytes between rows */
} hevcdbk_frame;

/* bS: replaces SetBoundaryStrenght(vert, n, hor, n) (cpu.h:120-132).  Layouts are the reference's:
 * vert[r*(W/8+1)+bx] = vertical edge at x = 8*bx of block row r; hor[by*(W/8)+c] = horizontal edge
 * at y = 8*by of block column c.  NULL pointers => reference default pattern (cpu.h:92-99).
 * The reference can only override luma; chroma overrides are an extension. */
typedef struct {
    const uint8_t *vert; size_t n_vert;
    const uint8_t *hor;  size_t n_hor;
    const uint8_t *chroma_vert; size_t n_chroma_vert;
    const uint8_t *chroma_hor;  size_t n_chroma_hor;
} hevcdbk_bs;

/* QP: replaces the ctor's Qp (cpu.h:35-37, main.cu:117,125,133).  map != NULL selects the per-CTU
 * map extension (BASELINE config 3b): QP of a 4-line edge segment = (QpP + QpQ + 1) >> 1 of the
 * CTUs holding P0 / Q0 of its first line. */
typedef struct {
    unsigned qp;
    const uint8_t *map;  /* map[cy*map_stride + cx], may be NULL */
    unsigned map_stride;
    unsigned ctu_log2;   /* CTU size in luma samples, log2 (6 => 64) */
} hevcdbk_qp;

/* tc / beta tables: replace beta_table / tc_table (cpu.h:1021-1033).  NULL => the reference's. */
typedef struct {
    const unsigned *tc;   /* 52 entries, each <= 255 */
    const unsigned *beta; /* 52 entries, each <= 255 */
} hevcdbk_tables;

/* the reference's three console figures (gpu.cu:1292, 1302-1303), seconds */
typedef struct {
    double exec_s;  /* "Execution Time without copy on GPU"  : kernels + sync   (gpu.cu:1266-1291) */
    double total_s; /* "Execution Time with copy on GPU"     : exec + copy      (gpu.cu:1302)      */
    double copy_s;  /* "Copy Operation Time with GPU buffers": H2D + D2H        (gpu.cu:1246-1258, 1294-1300) */
    double pipelined_s; /* wall time of the overlapped pinned/async pipeline actually run */
    /* Small frames (<= 2 MiB) are not copied at all: the kernel reads and writes page-locked host memory across PCIe
     * itself, so copy_s is 0 and exec_s contains the PCIe traffic.  Large frames: sums over the strips of the frame. */
} hevcdbk_timing;

/*
 * THE operator.  Replaces ReadYuvFrame::DeblockingFilter (cpu.h:134) and the body of ExecuteGpu
 * (gpu.cu:1246-1300: 7 H2D copies, 3 kernel launches, sync, 3 D2H copies) for a frame in HOST
 * memory, in place.  Frames up to 2 MiB: the fused Y+U+V kernel works on page-locked memory across the
 * link itself.  Larger frames are cut into strips of whole block rows: on a large-BAR device the
 * context's crew of host threads writes the strips into HBM through the BAR while the kernels of
 * earlier strips store their results into page-locked memory (the caller's own planes where those
 * are page-locked / registered with tight rows); elsewhere: pinned staging + hipMemcpyAsync on side
 * streams.  See hevcdbk_set_host_threads / hevcdbk_host_register below.  bs / tables / timing may be NULL.
 */
HEVCDBK_API int hevc_deblocking_filter(hevcdbk_context *ctx, hevcdbk_frame *frame, const hevcdbk_bs *bs,
                           const hevcdbk_qp *qp, const hevcdbk_tables *tables,
                           hevcdbk_timing *timing);

/*
 * Host side of hevc_deblocking_filter on a LARGE frame (> 2 MiB) in ordinary pageable memory -- the frame main.cu:112-133
 * gets from ReadYuvFrame, where the reference copies row by row into cudaMallocHost planes on one thread, outside its
 * timers (gpu.cu:1092-1133).  The frame is cut into strips of whole block rows (the first ones short, so the first DMA
 * starts early); a crew of host threads copies strips into the page-locked ring while earlier strips are on the link, in
 * the kernel or on their way back, and copies finished strips out.
 *   hevcdbk_set_host_threads: how many threads copy during a call, the calling thread included (1 = the caller alone;
 *   0 = the default, 4).  Crew threads are created at the first large pageable frame, run on the CPUs next to the GPU's
 *   host bridge (sysfs local_cpulist) where the process is allowed to, sleep between calls and end with the context.
 */
HEVCDBK_API int hevcdbk_set_host_threads(hevcdbk_context *ctx, unsigned n_threads);
HEVCDBK_API unsigned hevcdbk_get_host_threads(const hevcdbk_context *ctx);

/*
 * For callers that hand the same buffers in again and again (a decoder's picture pool): page-locks [ptr, ptr+bytes) in
 * place (hipHostRegister; the reference's equivalent is allocating the planes with cudaMallocHost, gpu.cu:1092-1099).
 * Planes inside a registered range are DMA'd where they lie, with no staging copy, by hevc_deblocking_filter,
 * hevc_deblocking_filter_sequence and the h265 / SAO host entries.  The caller unregisters before freeing the memory.
 */
HEVCDBK_API int hevcdbk_host_register(hevcdbk_context *ctx, void *ptr, size_t bytes);
HEVCDBK_API int hevcdbk_host_unregister(hevcdbk_context *ctx, void *ptr);

/*
 * Where the time of the LAST hevc_deblocking_filter call on a large frame went, strip by strip (measurement aid; the
 * reference prints three sums, gpu.cu:1292-1303).  Host-clock fields are seconds since the call began; a field that does
 * not apply (no staging for page-locked planes; GPU durations when the call had timing == NULL) is 0.
 */
typedef struct {
    int plane;                 /* 0 Y, 1 U, 2 V */
    unsigned row_begin, row_end;
    size_t bytes;
    double stage_begin_s, stage_end_s;     /* the crew's copy into the page-locked ring: first job began / last job ended */
    double enqueue_begin_s, enqueue_end_s; /* H2D + kernel + D2H of the strip handed to the HIP runtime */
    double d2h_seen_s;                     /* the calling thread saw the strip's download complete */
    double unstage_begin_s, unstage_end_s; /* the crew's copy back into the caller's plane */
    double h2d_ms, kernel_ms, d2h_ms;      /* GPU clock, between events on the three streams */
} hevcdbk_strip_trace;
/* copies min(cap, strips) entries to out (may be NULL with cap 0) and stores the strip count in *n_strips */
HEVCDBK_API int hevcdbk_last_frame_trace(const hevcdbk_context *ctx, hevcdbk_strip_trace *out, unsigned cap, unsigned *n_strips);

/*
 * Streaming form for a SEQUENCE of host frames of one geometry (the multi-frame case the reference does
 * not have).  Three frames are in flight: H2D of frame n+1, the kernels of frame n and
 * D2H of frame n-1 overlap on the context's three streams.  Planes that already are page-locked
 * (hevcdbk_host_malloc_pinned, hipHostMalloc, hipHostRegister) are DMA'd in place with no staging copy;
 * pageable planes go through the context's pinned ring.  bs / tables are shared by all frames;
 * timing->pipelined_s is the wall time of the whole sequence.  A per-CTU QP map is not accepted here.  Sequences of small
 * 8-bit 4:2:0 frames (<= 2 MiB per frame, at least 4 of them) travel in groups of up to 64 frames instead: packed back to
 * back into one pinned chunk, one DMA each way and one batched launch per group.
 */
HEVCDBK_API int hevc_deblocking_filter_sequence(hevcdbk_context *ctx, hevcdbk_frame *frames, unsigned n_frames,
                                    const hevcdbk_bs *bs, const hevcdbk_qp *qp, const hevcdbk_tables *tables,
                                    hevcdbk_timing *timing);

/*
 * Device-resident form of the same operator, for callers whose planes already live in HBM
 * (the decoder pipeline case, and what bench.py times).  One launch filters n_frames planes of
 * identical geometry.  src and dst are DEVICE pointers and may be equal (in place); blocks are
 * mutually independent so in-place is race-free.
 */
typedef struct {
    const void *src;       /* frame f plane at src + f*frame_stride */
    void *dst;
    size_t pitch;          /* bytes, multiple of 4 */
    size_t frame_stride;   /* bytes */
    unsigned n_frames;
    unsigned plane_w, plane_h; /* samples of THIS plane (chroma: W/2 x H/2) */
    unsigned bit_depth, sample_bytes;
    int is_chroma;         /* chroma block procedure: bS == 2 only, p0/q0 only (cpu.h:450-992) */
    const uint8_t *vert_bs; /* DEVICE; frame f at vert_bs + f*vert_bs_stride (stride 0 = shared) */
    const uint8_t *hor_bs;
    size_t vert_bs_stride, hor_bs_stride;
    const uint8_t *qp_map; /* DEVICE or NULL; frame f at qp_map + f*qp_map_frame_stride */
    unsigned qp_map_stride, ctu_log2; /* entries per map row (< 2^24, HEVCDBK_ERR_UNSUPPORTED otherwise); log2 of the map unit in luma samples, 3 .. 8 (HEVCDBK_ERR_ARG otherwise) */
    size_t qp_map_frame_stride;
} hevcdbk_device_planes;

/* kernel variant selector for hevc_deblocking_filter_device */
#define HEVCDBK_KERNEL_AUTO    0 /* fastest kernel that supports the operands */
#define HEVCDBK_KERNEL_GENERIC 1 /* one lane per offset block, 32-bit scalar arithmetic (all operand kinds) */
#define HEVCDBK_KERNEL_PACKED  2 /* packed 16-bit arithmetic kernel (8-bit samples, scalar QP) */
/* optional, OR-ed into the selector: how the packed kernels deal offset blocks to lanes.  Both maps produce the same
 * bytes; AUTO picks per geometry from measurements. */
#define HEVCDBK_MAP_AUTO   0x000
#define HEVCDBK_MAP_ROWS   0x100 /* one workgroup per block row */
#define HEVCDBK_MAP_LINEAR 0x200 /* row-major block numbering, workgroups renumbered per XCD */
#define HEVCDBK_MAP_MASK   0x700

HEVCDBK_API int hevc_deblocking_filter_device(hevcdbk_context *ctx, const hevcdbk_device_planes *planes,
                                  unsigned qp, const hevcdbk_tables *tables, int kernel_variant,
                                  void *hip_stream /* NULL => the context's compute stream */);

/* The planes of a batch of frames in ONE call -- normally Y, U, V of a 4:2:0 batch (planes[0] luma; 1 <= n_planes <= 3; all
 * with the same n_frames).  Replaces the reference's three launches per frame (gpu.cu:1266-1285): when the planes share one
 * sample format that the packed kernels take (8 bit -- the reference's case -- or 16-bit containers up to 12 bit) and a scalar
 * QP they go out as ONE fused launch, whatever the frame size; other operands (a QP map, deeper samples) are launched plane by
 * plane, exactly as n_planes calls of hevc_deblocking_filter_device would.  Same bytes either way. */
HEVCDBK_API int hevc_deblocking_filter_device_planes(hevcdbk_context *ctx, const hevcdbk_device_planes *planes,
                                         unsigned n_planes, unsigned qp, const hevcdbk_tables *tables,
                                         int kernel_variant, void *hip_stream);

/* ---- device memory / stream plumbing for hosts without a HIP binding (ctypes, cgo, JNI) ---- */
HEVCDBK_API int hevcdbk_device_malloc(hevcdbk_context *ctx, size_t bytes, void **dptr);
HEVCDBK_API int hevcdbk_device_free(hevcdbk_context *ctx, void *dptr);
HEVCDBK_API int hevcdbk_host_malloc_pinned(hevcdbk_context *ctx, size_t bytes, void **hptr); /* hipHostMalloc, replaces cudaMallocHost gpu.cu:1103-1169 */
HEVCDBK_API int hevcdbk_host_free_pinned(hevcdbk_context *ctx, void *hptr);
HEVCDBK_API int hevcdbk_memcpy_h2d(hevcdbk_context *ctx, void *dptr, const void *hptr, size_t bytes); /* synchronous */
HEVCDBK_API int hevcdbk_memcpy_d2h(hevcdbk_context *ctx, void *hptr, const void *dptr, size_t bytes);
HEVCDBK_API int hevcdbk_memcpy_d2d(hevcdbk_context *ctx, void *dst, const void *src, size_t bytes);
HEVCDBK_API int hevcdbk_memset_d(hevcdbk_context *ctx, void *dptr, int value, size_t bytes);
HEVCDBK_API int hevcdbk_synchronize(hevcdbk_context *ctx); /* all three streams */
HEVCDBK_API void *hevcdbk_compute_stream(hevcdbk_context *ctx); /* hipStream_t */

/*
 * A destination pool chosen among several allocations.  Where the driver puts a buffer's physical pages decides how fast the
 * kernels for 16-bit containers write into it: identical launches run up to 13 % apart from one allocation to the next, the
 * difference follows the DESTINATION buffer, stays with it for its lifetime, and nothing cheaper than the filter itself
 * predicts it (a fill or a DMA copy does not; it is not address translation: profiles/r04/placement.md).  For a pool that lives
 * long -- a decoder's output pictures -- it pays to look once: `probe` describes a launch exactly as
 * hevc_deblocking_filter_device takes it (its dst field is ignored); the function allocates `candidates` (1 .. 16) buffers of the
 * size that launch writes (frame_stride * (n_frames - 1) + pitch * plane_h), runs the launch into each (one warm-up, three
 * timed), keeps the fastest, frees the others and returns it in *dptr (release with hevcdbk_device_free).  best_ms / worst_ms
 * (may be NULL): the fastest and the slowest candidate's kernel time.  The reference allocates its device planes once per run
 * with cudaMalloc (gpu.cu:1110-1133) and has no such choice to make.
 */
HEVCDBK_API int hevcdbk_device_malloc_probed(hevcdbk_context *ctx, const hevcdbk_device_planes *probe, unsigned qp,
                                             const hevcdbk_tables *tables, unsigned candidates, void **dptr, float *best_ms,
                                             float *worst_ms);

/* Timed replay for benchmarks: launches the device operator `steps` times back-to-back on the
 * compute stream with a HIP event pair around EACH launch, synchronises once at the end and
 * writes the per-launch kernel durations (milliseconds) to kernel_ms[steps].  (= hevcdbk_device_replay with no
 * settling and no warm-up.)  The timing window is the reference's "execution time without copy": kernels + one
 * synchronisation, nothing else (gpu.cu:1266-1291).  n_planes is 1 .. 3 (the planes of one 4:2:0 batch) and every plane
 * holds the same n_frames -- anything else is HEVCDBK_ERR_ARG, here and in hevcdbk_device_replay; the compute stream is
 * drained before the first launch, so work an earlier asynchronous call left on it is outside the window. */
HEVCDBK_API int hevcdbk_device_run_timed(hevcdbk_context *ctx, const hevcdbk_device_planes *planes, unsigned n_planes,
                             unsigned qp, const hevcdbk_tables *tables, int kernel_variant,
                             unsigned steps, float *kernel_ms);

/* Steady-state replay: ONE uninterrupted stream of [settle launches][warmup launches][steps timed launches] of the device
 * operator on the compute stream, one synchronisation at the very end (same window semantics as above).  Settling is by
 * time: launches go out until the mean duration of the trailing settle_window launches is within settle_tolerance of the
 * mean of the window before it, for at least settle_min_ms and at most settle_max_ms (settle_max_ms == 0: no settling;
 * settle_min_ms == settle_max_ms: a fixed time).  The host stays ahead of the GPU throughout, so the queue never drains
 * between the phases; the front bracket of the timed window is the host observing the end of the last launch before it. */
typedef struct {
    /* in */
    double settle_min_ms, settle_max_ms;
    double settle_tolerance;   /* relative; 0 => 0.005 */
    unsigned settle_window;    /* launches; 0 => 32 */
    unsigned warmup, steps;
    /* out */
    unsigned settle_launches;  /* launches the settle phase issued */
    int settled;               /* 1: the criterion was met before settle_max_ms */
    double settle_ms;          /* host time the settle phase took to issue */
    double settle_tail_mean_ms;/* mean duration of its last settle_window launches */
    double t_begin, t_end;     /* CLOCK_MONOTONIC seconds: end of the launch before the first timed one as seen by the
                                * host / return of the final synchronisation */
    double wall_ms;            /* (t_end - t_begin) * 1e3: host wall clock of the `steps` timed launches */
    double span_ms;            /* GPU timestamps: begin of the first timed launch -> end of the last one */
} hevcdbk_replay;
HEVCDBK_API int hevcdbk_device_replay(hevcdbk_context *ctx, const hevcdbk_device_planes *planes, unsigned n_planes,
                          unsigned qp, const hevcdbk_tables *tables, int kernel_variant, hevcdbk_replay *replay,
                          float *kernel_ms /* [replay->steps] */);

/* PCI bus id ("0000:0a:00.0") of the context's device, for hosts that read the card's sysfs files (clock / power telemetry);
 * replaces nothing in the reference (GetGpuDeviceInfo, main.cu:92-107, prints properties only).  len >= 13. */
HEVCDBK_API int hevcdbk_device_pci_bus_id(const hevcdbk_context *ctx, char *buf, size_t len);

/* ---- main.cu-shaped harness entry: replaces ExecuteGpu (main.cu:87-90, gpu.cu:1230-1306) ----
 * file in -> filter Y,U,V on the GPU -> file out, printing the reference's three lines.
 * The four launch-dimension arguments are accepted and ignored.  Returns an error code instead of
 * throwing.  `device` selects the HIP device (the reference always uses device 0, main.cu:93). */
HEVCDBK_API int hevcdbk_execute_gpu(const char *input_file_name, const char *output_file_name,
                        unsigned width, unsigned height, unsigned qp,
                        unsigned dimx1, unsigned dimy1, unsigned dimx2, unsigned dimy2, int device);

/*
 * Multi-frame planar 8-bit 4:2:0 file -> file.  The reference's reader takes exactly one frame per file
 * (cpu.h:40-45, gpu.cu:1077-1084: anything else throws "Incorrect file size"); a sequence is what a caller holds in
 * practice, so this entry accepts any whole number of frames (>= 1) and is byte-identical, frame by frame, to running
 * the reference's ReadYuvFrame -> [SetBoundaryStrenght] -> DeblockingFilter -> Save on each.  File reads, GPU work and
 * file writes overlap, chunk by chunk, through three pinned buffers; a chunk (up to 64 frames / 64 MiB, frames back to
 * back as in the file) moves in ONE DMA each way and its planes are filtered as batches of frames whose frame stride is
 * one file frame -- one fused Y+U+V launch per chunk when the geometry allows.  `bs` / `tables` as for hevc_deblocking_filter (NULL = the
 * reference's defaults).  in == out is refused.  timing->pipelined_s = wall time including file I/O.
 */
HEVCDBK_API int hevcdbk_filter_yuv_file(hevcdbk_context *ctx, const char *input_file_name, const char *output_file_name,
                            unsigned width, unsigned height, unsigned qp, const hevcdbk_bs *bs,
                            const hevcdbk_tables *tables, unsigned *n_frames, hevcdbk_timing *timing);

/*
 * The same operator sharded over several GPUs of one node (frames are independent, so there is no exchange
 * step and no collective).  devices[g] is the HIP device of worker g; chunk c of the file (up to 64 frames) is read,
 * filtered and written back at its own offset by worker c mod n_devices, each worker being one host thread with its own
 * context.  A device may be listed more than once (two workers sharing one GPU).  The output is byte-identical to
 * hevcdbk_filter_yuv_file's.  Error codes as there; the first failing worker's code is returned.
 */
HEVCDBK_API int hevcdbk_filter_yuv_file_multi(const int *devices, unsigned n_devices, const char *input_file_name,
                                  const char *output_file_name, unsigned width, unsigned height, unsigned qp,
                                  const hevcdbk_bs *bs, const hevcdbk_tables *tables, unsigned *n_frames,
                                  hevcdbk_timing *timing);

/* ==================================================================================================================
 * Spec-exact mode: ITU-T H.265 (HEVC) clause 8.7.2.
 *
 * The reference's filter is HEVC-shaped but not conformant: it hard-codes bS = 2 (cpu.h:91-99) and one QP per frame
 * (cpu.h:136-137), and its arithmetic differs from the standard in the strong-filter thresholds (cpu.h:1099-1110), the
 * clip of the normal filter's delta (cpu.h:1256), the chroma formula (cpu.h:1453-1458), the hor2 column pairing
 * (cpu.h:383-414) and the filtered frame edges.  The entries below are what a decoder needs instead.  They never change
 * the results of the reference-exact entries above.  Parity: checked against oracle/h265_oracle.c, a restatement of the
 * standard's text written in picture order (no implementation of the standard exists in the reference or in this
 * image: "parity unpinned").
 *
 * bS arrays are 4-sample granular (the standard's unit):
 *   vert: (W/8+1) columns x (H/4) rows, entry (y4, bx) = the edge x = 8*bx over rows 4*y4 .. 4*y4+3
 *   hor:  (H/8+1) rows x (W/4) columns, entry (by, x4) = the edge y = 8*by over columns 4*x4 .. 4*x4+3
 * for a W x H plane; chroma planes (4:2:0) carry their own arrays in the chroma plane's geometry.  An entry holds the
 * bS in bits 1:0 and two flags: HEVCDBK_H265_KEEP_P / _KEEP_Q leave the P / Q samples unmodified (pcm_loop_filter_
 * disabled_flag with a PCM block, cu_transquant_bypass_flag: nDp / nDq = 0 in 8.7.2.5.7).  Edges on the picture boundary
 * are never filtered, whatever the arrays hold.
 * ================================================================================================================== */
#define HEVCDBK_H265_BS_MASK 3u
#define HEVCDBK_H265_KEEP_P  4u
#define HEVCDBK_H265_KEEP_Q  8u

typedef struct hevcdbk_h265_params {
    int tc_offset_div2;   /* slice_tc_offset_div2 (or pps_tc_offset_div2), -6..6 */
    int beta_offset_div2; /* slice_beta_offset_div2, -6..6 */
    int cb_qp_offset;     /* pps_cb_qp_offset (+ slice_cb_qp_offset is NOT added: 8.7.2.5.5 uses cQpPicOffset), -12..12 */
    int cr_qp_offset;     /* pps_cr_qp_offset */
} hevcdbk_h265_params;

HEVCDBK_API size_t hevcdbk_h265_num_vert_bs(unsigned plane_w, unsigned plane_h); /* (plane_w/8+1) * (plane_h/4) */
HEVCDBK_API size_t hevcdbk_h265_num_hor_bs(unsigned plane_w, unsigned plane_h);  /* (plane_h/8+1) * (plane_w/4) */

/* per 4x4 luma unit prediction data the bS derivation (8.7.2.4) reads; (W/4) x (H/4) entries, row-major */
#define HEVCDBK_U_INTRA     0x0001u /* CuPredMode == MODE_INTRA */
#define HEVCDBK_U_CBF       0x0002u /* the luma transform block covering the unit has non-zero coefficient levels */
#define HEVCDBK_U_TU_LEFT   0x0004u /* the unit's left border is a transform block edge */
#define HEVCDBK_U_TU_TOP    0x0008u
#define HEVCDBK_U_PU_LEFT   0x0010u /* the unit's left border is a prediction block edge */
#define HEVCDBK_U_PU_TOP    0x0020u
#define HEVCDBK_U_KEEP      0x0040u /* samples stay unmodified (PCM + pcm_loop_filter_disabled_flag, transquant bypass) */
#define HEVCDBK_U_DBK_OFF   0x0080u /* slice_deblocking_filter_disabled_flag of the slice holding the unit */
#define HEVCDBK_U_PRED_L0   0x0100u /* predFlagL0: mv0 / ref0 valid */
#define HEVCDBK_U_PRED_L1   0x0200u
#define HEVCDBK_U_NOX_LEFT  0x0400u /* left border is a slice / tile boundary in-loop filtering must not cross */
#define HEVCDBK_U_NOX_TOP   0x0800u

typedef struct hevcdbk_h265_units {
    const uint16_t *flags;
    const int16_t *mv0;   /* [unit][2]: x, y in quarter luma samples */
    const int16_t *mv1;
    const int32_t *ref0;  /* identity of the reference PICTURE (e.g. its POC), not its index in the list */
    const int32_t *ref1;
} hevcdbk_h265_units;

/*
 * 8.7.2.4 on the GPU.  `units` and the four outputs are DEVICE pointers; the luma outputs have
 * hevcdbk_h265_num_vert_bs(w, h) / _hor_bs(w, h) entries, the chroma outputs (4:2:0, may both be NULL) those of the
 * (w/2) x (h/2) plane: the luma entry at twice the chroma position (8.7.2.5).  Asynchronous on `hip_stream`
 * (NULL = the context's compute stream).
 */
HEVCDBK_API int hevcdbk_h265_derive_bs_device(hevcdbk_context *ctx, const hevcdbk_h265_units *units, unsigned width, unsigned height,
                                  uint8_t *vert_bs4, uint8_t *hor_bs4, uint8_t *chroma_vert_bs4, uint8_t *chroma_hor_bs4,
                                  void *hip_stream);

/*
 * Spec-exact filter on planes that live in HBM.  `planes` as for hevc_deblocking_filter_device, except that vert_bs /
 * hor_bs are the 4-sample-granular arrays of THIS plane's geometry and qp_map / ctu_log2 describe QpY per unit of
 * (1 << ctu_log2) luma samples (3 = per 8x8, the finest a quantization group can be).  c_idx 0 = luma, 1 = Cb, 2 = Cr
 * (selects cb_qp_offset / cr_qp_offset; planes->is_chroma must agree).  The tables are the standard's (Table 8-12, 54 tc
 * entries); tc index = qPL + 2*(bS-1) + 2*tc_offset_div2, beta index = qPL + 2*beta_offset_div2, chroma through Table 8-10.
 * kernel_variant: HEVCDBK_KERNEL_AUTO, _GENERIC (32-bit arithmetic, every operand kind) or _PACKED (8-bit samples).
 */
HEVCDBK_API int hevc_deblocking_filter_h265_device(hevcdbk_context *ctx, const hevcdbk_device_planes *planes, int c_idx, unsigned qp,
                                       const hevcdbk_h265_params *params, int kernel_variant, void *hip_stream);

/*
 * Host-frame operator of the spec-exact mode: uploads the frame and either `units` (HOST arrays; bS derived on the GPU)
 * or `bs4` (HOST luma arrays in the 4-sample-granular layout, n_vert / n_hor checked against hevcdbk_h265_num_*; the
 * chroma arrays are gathered from them on the GPU; its chroma_* members are ignored), filters Y and, when present, Cb and
 * Cr in place, downloads.  Exactly one of `units` / `bs4` must be non-NULL.  `qp` as for hevc_deblocking_filter.
 */
HEVCDBK_API int hevc_deblocking_filter_h265(hevcdbk_context *ctx, hevcdbk_frame *frame, const hevcdbk_h265_units *units,
                                const hevcdbk_bs *bs4, const hevcdbk_qp *qp, const hevcdbk_h265_params *params,
                                hevcdbk_timing *timing);

/* ==================================================================================================================
 * Sample adaptive offset, ITU-T H.265 clause 8.7.3: the in-loop stage that follows deblocking.  Not
 * present in the reference; parity against oracle/h265_oracle.c only ("parity unpinned").
 * ================================================================================================================== */
typedef struct hevcdbk_sao_ctb {
    uint8_t type;      /* SaoTypeIdx: 0 = not applied, 1 = band offset, 2 = edge offset */
    uint8_t cls;       /* band offset: sao_band_position (0..31); edge offset: SaoEoClass (0 hor, 1 ver, 2 135 deg, 3 45 deg) */
    int8_t offset[4];  /* SaoOffsetVal[1..4]: signed, already scaled by log2OffsetScale */
} hevcdbk_sao_ctb;

/*
 * SAO of planes that live in HBM, src -> dst (they must differ: the edge classifier reads deblocked neighbours).  Of
 * `planes` the members src, dst, pitch, frame_stride, n_frames, plane_w, plane_h (multiples of 8), bit_depth and
 * sample_bytes are used.  `params` (DEVICE memory): one entry per CTB of THIS plane, row stride params_stride entries,
 * params_frame_stride entries between frames (0 = shared); ctb_log2 = CTB size of this plane in samples (luma 4..6, 4:2:0
 * chroma one less: 3..5).  `keep` (DEVICE, may be NULL): one byte per 8x8 samples of this plane, non-zero = left
 * unmodified (PCM with pcm_loop_filter_disabled_flag, cu_transquant_bypass_flag), row stride keep_stride, frame stride
 * keep_frame_stride bytes.  Edge offset leaves a sample alone when one of its two neighbours is outside the picture.
 */
HEVCDBK_API int hevc_sao_filter_device(hevcdbk_context *ctx, const hevcdbk_device_planes *planes, const hevcdbk_sao_ctb *params,
                           unsigned params_stride, size_t params_frame_stride, unsigned ctb_log2, const uint8_t *keep,
                           unsigned keep_stride, size_t keep_frame_stride, void *hip_stream);

/*
 * Deblocking followed by SAO in ONE call, src -> dst (they must differ), of planes that live in HBM -- the in-loop chain of
 * a decoder.  Operands as for hevc_deblocking_filter_device (reference-exact mode: `qp`, `tables`, the
 * planes' bS arrays) resp. hevc_deblocking_filter_h265_device (spec-exact mode: `c_idx`, `qp`, `h265_params`, the planes'
 * 4-sample-granular bS arrays) and for hevc_sao_filter_device (`params` ... `keep_frame_stride`, all DEVICE memory).
 * For 8-bit planes and for 16-bit containers up to 12 bit, with one QP or a QP map, both stages run in ONE kernel: a workgroup
 * deblocks the offset blocks of a tile (8 bit: 192 x 128 samples, 16-bit containers: 128 x 128; plus a one-sample rim) into
 * LDS and applies SAO from there, so every sample is read from HBM once and written once and the deblocked picture never
 * exists in memory.  Other operands (samples deeper than 12 bit, misaligned 16-bit planes) run as two launches through a scratch plane owned
 * by the context.  `fused`: HEVCDBK_FUSED_AUTO picks, _OFF forces the two launches (same bytes; for A/B runs), _ON returns
 * HEVCDBK_ERR_UNSUPPORTED where the fused kernel does not apply.  The scratch plane of the two-launch form belongs to the
 * context and is reused by the next such call; its reuse is fenced by an event, so calls that hand in different streams are
 * ordered on it (like every entry point, a context serves one host thread at a time).  Parity: the deblocking stage as for
 * its own entry points; SAO against oracle/h265_oracle.c only ("parity unpinned").
 */
#define HEVCDBK_FUSED_AUTO 0
#define HEVCDBK_FUSED_OFF  1
#define HEVCDBK_FUSED_ON   2
HEVCDBK_API int hevc_deblock_sao_device(hevcdbk_context *ctx, const hevcdbk_device_planes *planes, unsigned qp,
                                        const hevcdbk_tables *tables, const hevcdbk_sao_ctb *params, unsigned params_stride,
                                        size_t params_frame_stride, unsigned ctb_log2, const uint8_t *keep, unsigned keep_stride,
                                        size_t keep_frame_stride, int fused, void *hip_stream);
HEVCDBK_API int hevc_deblock_sao_h265_device(hevcdbk_context *ctx, const hevcdbk_device_planes *planes, int c_idx, unsigned qp,
                                             const hevcdbk_h265_params *h265_params, const hevcdbk_sao_ctb *params,
                                             unsigned params_stride, size_t params_frame_stride, unsigned ctb_log2,
                                             const uint8_t *keep, unsigned keep_stride, size_t keep_frame_stride, int fused,
                                             void *hip_stream);

/*
 * The same for the planes of a 4:2:0 batch -- planes[0] luma, the others chroma, n_planes <= 3, one frame count -- in ONE
 * call: where the fused kernel takes every plane (one sample width and bit depth) they go out as ONE launch, the planes'
 * tiles one after the other in the grid; otherwise plane by plane exactly as n_planes calls of the entries above would
 * (every plane is checked before the first launch).  sao[i] = the SAO operands of planes[i] (its own CTB grid: 4:2:0 chroma
 * has ctb_log2 one less than luma).  Spec-exact mode: c_idx = the plane's index (0 Y, 1 Cb, 2 Cr).  The reference has
 * neither stage after DeblockingFilter (main.cu:41-43) nor a multi-plane launch (gpu.cu:1269-1285: three launches).
 */
typedef struct {
    const hevcdbk_sao_ctb *params; /* DEVICE memory */
    unsigned params_stride;
    size_t params_frame_stride;
    unsigned ctb_log2;
    const uint8_t *keep;           /* DEVICE memory, may be NULL */
    unsigned keep_stride;
    size_t keep_frame_stride;
} hevcdbk_sao_plane;
HEVCDBK_API int hevc_deblock_sao_device_planes(hevcdbk_context *ctx, const hevcdbk_device_planes *planes, unsigned n_planes,
                                               unsigned qp, const hevcdbk_tables *tables, const hevcdbk_sao_plane *sao, int fused,
                                               void *hip_stream);
HEVCDBK_API int hevc_deblock_sao_h265_device_planes(hevcdbk_context *ctx, const hevcdbk_device_planes *planes, unsigned n_planes,
                                                    unsigned qp, const hevcdbk_h265_params *h265_params,
                                                    const hevcdbk_sao_plane *sao, int fused, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* HEVC_DEBLOCK_H */
