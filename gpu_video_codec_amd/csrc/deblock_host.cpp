/*
 * deblock_host.cpp -- C-ABI host driver (include/hevc_deblock.h) above the HIP kernels.
 *
 * Replaces the host side of the reference's GPU path (gpu.cu:35-77 globals, 1074-1203
 * Initialize/Release, 1230-1306 ExecuteGpu) with a re-entrant per-device context:
 * large frames cut into strips that a crew of host threads (host_crew.h) writes into HBM through
 * the PCIe BAR while the kernels of earlier strips store into page-locked memory (round 4), pinned
 * hipHostMalloc staging + hipMemcpyAsync on two copy side streams where there is no large BAR,
 * kernels on a compute stream.
 *
 * There is deliberately no CPU implementation of the filter in this library.
 *
 * The context and the helpers shared with deblock_host_h265.cpp (spec-exact mode, SAO) are declared in deblock_ctx.h.
 */
#include <hip/hip_runtime_api.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cctype>
#include <chrono>
#include <cmath>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "deblock_ctx.h"
#include "host_crew.h"
#ifdef HEVCDBK_DIAG
#include "hevcdbk_diag.h"
#endif

namespace {

/* cpu.h:1021-1033 (identical copies at gpu.cu:80-85, 92-97) */
const unsigned k_beta_table[52] = {
    0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,
    6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24,
    26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56,
    58, 60, 62, 64};
const unsigned k_tc_table[52] = {
    0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,
    0,  0,  1,  1,  1,  1,  1,  1,  1,  1,  1,  2,  2,  2,  2,  3,
    3,  3,  3,  4,  4,  4,  5,  5,  6,  6,  7,  8,  9,  10, 11, 13,
    14, 16, 18, 20};


} /* namespace */


namespace dbkh {

bool hip_ok(hevcdbk_context *ctx, hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    if (ctx) ctx->last_error = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError(); /* clear sticky error */
    return false;
}

int bind(hevcdbk_context *ctx)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return HEVCDBK_OK;
}

int grow_pinned(hevcdbk_context *ctx, Growable &g, size_t bytes)
{
    if (g.cap >= bytes) return HEVCDBK_OK;
    if (g.p) (void)hipHostFree(g.p);
    g.p = nullptr; g.cap = 0;
    HIP_TRY(ctx, hipHostMalloc(&g.p, bytes, hipHostMallocDefault));
    g.cap = bytes;
    return HEVCDBK_OK;
}

int grow_device(hevcdbk_context *ctx, Growable &g, size_t bytes)
{
    if (g.cap >= bytes) return HEVCDBK_OK;
    if (g.p) (void)hipFree(g.p);
    g.p = nullptr; g.cap = 0;
    HIP_TRY(ctx, hipMalloc(&g.p, bytes));
    g.cap = bytes;
    return HEVCDBK_OK;
}

bool bad_depth(unsigned bit_depth, unsigned sample_bytes)
{
    return bit_depth < 8 || bit_depth > 16 || (sample_bytes != 1 && sample_bytes != 2) ||
           (sample_bytes == 1 && bit_depth != 8);
}

/* true when p is page-locked host memory the GPU can DMA from directly; *dev_ptr (if asked for) = the address a KERNEL uses for
 * it: the same as p for hipHostMalloc memory, possibly another for a hipHostRegister'ed range */
bool is_pinned_host(const void *p, void **dev_ptr)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError(); /* ordinary malloc memory: not an error for us */
        return false;
    }
    if (at.type != hipMemoryTypeHost) return false;
    if (dev_ptr) *dev_ptr = at.devicePointer;
    return true;
}

/* the checks every host-frame entry makes on a frame: depth, the reference's dimension rule (cpu.h:46-48, for the chroma
 * planes as well when they are present), pitches */
int check_frame(const hevcdbk_frame &f, bool &chroma)
{
    if (!f.plane[0] || bad_depth(f.bit_depth, f.sample_bytes)) return HEVCDBK_ERR_ARG;
    const unsigned W = f.width, H = f.height;
    if (W == 0 || H == 0 || W % 8 != 0 || H % 8 != 0) return HEVCDBK_ERR_DIMENSIONS;
    chroma = f.plane[1] && f.plane[2];
    if (chroma && ((W / 2) % 8 != 0 || (H / 2) % 8 != 0)) return HEVCDBK_ERR_DIMENSIONS;
    for (int i = 0; i < (chroma ? 3 : 1); i++)
        if (f.pitch[i] < (size_t)(i ? W / 2 : W) * f.sample_bytes) return HEVCDBK_ERR_ARG;
    return HEVCDBK_OK;
}

/* caller bS arrays of the reference layout: both or neither of a pair, exact sizes (cpu.h:122-123) */
int check_bs(const hevcdbk_bs *bs, unsigned W, unsigned H, bool chroma)
{
    if (!bs) return HEVCDBK_OK;
    if ((bs->vert != nullptr) != (bs->hor != nullptr) || (bs->chroma_vert != nullptr) != (bs->chroma_hor != nullptr)) return HEVCDBK_ERR_ARG;
    if (bs->vert && (bs->n_vert != hevcdbk_num_vert_bs(W, H) || bs->n_hor != hevcdbk_num_hor_bs(W, H))) return HEVCDBK_ERR_BS_SIZE;
    if (bs->chroma_vert && (!chroma || bs->n_chroma_vert != hevcdbk_num_vert_bs(W / 2, H / 2) ||
                            bs->n_chroma_hor != hevcdbk_num_hor_bs(W / 2, H / 2)))
        return HEVCDBK_ERR_BS_SIZE;
    return HEVCDBK_OK;
}

int fill_tables(DbkArgs &a, const hevcdbk_tables *tables, unsigned qp, unsigned bit_depth)
{
    const unsigned *tc = (tables && tables->tc) ? tables->tc : k_tc_table;
    const unsigned *beta = (tables && tables->beta) ? tables->beta : k_beta_table;
    for (int i = 0; i < 52; i++) {
        if (tc[i] > 255 || beta[i] > 255) return HEVCDBK_ERR_ARG;
        a.tc_tab[i] = (uint8_t)tc[i];
        a.beta_tab[i] = (uint8_t)beta[i];
    }
    const unsigned q = qp > 51 ? 51 : qp; /* cpu.h:1064-1072 */
    a.shift = (int)bit_depth - 8;
    a.tc = (int)(tc[q] << a.shift);   /* cpu.h:137; scaled as H.265 8.7.2.5.3 for bit_depth > 8 */
    a.beta = (int)(beta[q] << a.shift); /* cpu.h:136 */
    a.max_v = (1 << bit_depth) - 1;   /* cpu.h:1202 */
    return HEVCDBK_OK;
}

/* geometry of one plane -> DbkArgs (everything except pointers / tables) */
void fill_geometry(DbkArgs &a, unsigned plane_w, unsigned plane_h, int is_chroma)
{
    a.plane_w = (int)plane_w;
    a.plane_h = (int)plane_h;
    a.nbx = (int)(plane_w / 8 + 1); /* cpu.h:141 */
    a.nby = (int)(plane_h / 8 + 1); /* cpu.h:142 */
    a.vstride = (int)(plane_w / 8 + 1);
    a.hstride = (int)(plane_w / 8);
    a.n_vert = (int)hevcdbk_num_vert_bs(plane_w, plane_h);
    a.n_hor = (int)hevcdbk_num_hor_bs(plane_w, plane_h);
    /* cpu.h:224, 369 (luma) vs cpu.h:515, 645 (chroma compares against the LUMA block counts) */
    a.limit_bx = is_chroma ? (int)(2 * plane_w / 8) : a.nbx - 1;
    a.limit_by = is_chroma ? (int)(2 * plane_h / 8) : a.nby - 1;
}

int planes_to_args(const hevcdbk_device_planes *p, unsigned qp, const hevcdbk_tables *tables, DbkArgs &a)
{
    if (!p || !p->src || !p->dst || !p->vert_bs || !p->hor_bs) return HEVCDBK_ERR_ARG;
    if (bad_depth(p->bit_depth, p->sample_bytes)) return HEVCDBK_ERR_ARG;
    if (p->plane_w == 0 || p->plane_h == 0 || p->plane_w % 8 != 0 || p->plane_h % 8 != 0)
        return HEVCDBK_ERR_DIMENSIONS; /* cpu.h:46-48 */
    const size_t align = 4 * p->sample_bytes; /* one 4-sample word */
    if (p->pitch % align != 0 || p->frame_stride % align != 0 ||
        (uintptr_t)p->src % align != 0 || (uintptr_t)p->dst % align != 0)
        return HEVCDBK_ERR_UNSUPPORTED;
    if (p->pitch < (size_t)p->plane_w * p->sample_bytes) return HEVCDBK_ERR_ARG;
    if (p->n_frames > 65535) return HEVCDBK_ERR_UNSUPPORTED;
    std::memset(&a, 0, sizeof(a));
    a.src = (const uint8_t *)p->src;
    a.dst = (uint8_t *)p->dst;
    a.pitch = (long long)p->pitch;
    a.frame_stride = (long long)p->frame_stride;
    a.n_frames = (int)p->n_frames;
    fill_geometry(a, p->plane_w, p->plane_h, p->is_chroma);
    a.vert_bs = p->vert_bs;
    a.hor_bs = p->hor_bs;
    a.vert_bs_stride = (long long)p->vert_bs_stride;
    a.hor_bs_stride = (long long)p->hor_bs_stride;
    if (p->qp_map && (p->ctu_log2 < 3 || p->ctu_log2 > 8)) return HEVCDBK_ERR_ARG; /* as the host-frame operator: units of 8 .. 256 luma samples */
    if (p->qp_map && p->qp_map_stride >= (1u << 24)) return HEVCDBK_ERR_UNSUPPORTED; /* the kernels index the map with 24-bit multiplies */
    a.qp_map = p->qp_map;
    a.map_stride = (int)p->qp_map_stride;
    a.ctu_log2 = (int)p->ctu_log2;
    a.map_frame_stride = (long long)p->qp_map_frame_stride;
    return fill_tables(a, tables, qp, p->bit_depth);
}

int launch(hevcdbk_context *ctx, const DbkArgs &a0, int sample_bytes, bool chroma, int variant, hipStream_t s)
{
    hipError_t e;
    /* bits 8..9 of the selector: block -> lane map of the packed kernels (HEVCDBK_MAP_*); same bytes either way */
    const int map = variant & HEVCDBK_MAP_MASK;
    variant &= ~HEVCDBK_MAP_MASK;
    #ifdef HEVCDBK_DIAG
    if (map != HEVCDBK_MAP_AUTO && map != HEVCDBK_MAP_ROWS && map != HEVCDBK_MAP_LINEAR && map != HEVCDBK_DIAG_MAP_TILES && map != HEVCDBK_DIAG_MAP_STRIPE && map != HEVCDBK_DIAG_MAP_PIPE && map != HEVCDBK_DIAG_MAP_GROUP) return HEVCDBK_ERR_ARG;
#else
    if (map != HEVCDBK_MAP_AUTO && map != HEVCDBK_MAP_ROWS && map != HEVCDBK_MAP_LINEAR) return HEVCDBK_ERR_ARG;
#endif
    DbkArgs a = a0;
    a.n_cus = ctx->n_cus;
    a.map_override = map == HEVCDBK_MAP_ROWS ? 1 : (map == HEVCDBK_MAP_LINEAR ? 2 : (map == 0x300 ? 3 : (map == 0x400 ? 4 : (map == 0x500 ? 5 : (map == 0x600 ? 6 : 0)))));
#ifdef HEVCDBK_DIAG
    if (variant == HEVCDBK_DIAG_KERNEL_COPY) {
        if (!dbk_packed_supports(a, sample_bytes, chroma)) return HEVCDBK_ERR_UNSUPPORTED;
        e = dbk_launch_packed(a, sample_bytes, chroma, 1, s);
    } else
#endif
    if (variant == HEVCDBK_KERNEL_PACKED) {
        if (!dbk_packed_supports(a, sample_bytes, chroma)) return HEVCDBK_ERR_UNSUPPORTED;
        e = dbk_launch_packed(a, sample_bytes, chroma, 0, s);
    } else if (variant == HEVCDBK_KERNEL_GENERIC) {
        e = dbk_launch_generic(a, sample_bytes, chroma, s);
    } else if (variant == HEVCDBK_KERNEL_AUTO) {
        e = dbk_packed_supports(a, sample_bytes, chroma) ? dbk_launch_packed(a, sample_bytes, chroma, 0, s)
                                                 : dbk_launch_generic(a, sample_bytes, chroma, s);
    } else {
        return HEVCDBK_ERR_ARG;
    }
    return hip_ok(ctx, e, "kernel launch") ? HEVCDBK_OK : HEVCDBK_ERR_HIP;
}

#ifdef HEVCDBK_DIAG
/* hevcdbk_diag.h: the knobs of the diagnostic library (never compiled into libhevcdbk.so) */
extern "C" HEVCDBK_API int hevcdbk_diag_set(const char *spec)
{
    DbkDiag d; /* the defaults of the struct */
    const std::string sp = spec ? spec : "";
    size_t i = 0;
    while (i < sp.size()) {
        size_t j = sp.find(',', i);
        if (j == std::string::npos) j = sp.size();
        const std::string tok = sp.substr(i, j - i);
        i = j + 1;
        if (tok.empty()) continue;
        if (tok == "noswz") d.noswz = 1;
        else if (tok == "nofuse") d.nofuse = 1;
        else if (tok == "dmacopy") d.dmacopy = 1;
        else if (tok == "nostrong") d.ablate = 1;
        else if (tok == "nonormal") d.ablate = 2;
        else if (tok == "barriers") d.ablate = 4;
        else if (tok == "queue") d.queue = 1;
        else if (tok == "align") d.align = 1;
        else if (tok == "mode3") d.mode3 = 1;
        else if (tok.compare(0, 5, "prio=") == 0) d.prio = std::atoi(tok.c_str() + 5) & 3;
        else if (tok.compare(0, 6, "dummy=") == 0) d.dummy = std::atoi(tok.c_str() + 6);
        else if (tok.compare(0, 5, "rows=") == 0) d.rows = std::atoi(tok.c_str() + 5);
        else if (tok.compare(0, 4, "lds=") == 0) d.lds = std::atoi(tok.c_str() + 4);
        else if (tok.compare(0, 5, "xpad=") == 0) d.xpad = std::atoi(tok.c_str() + 5);
        else if (tok.compare(0, 3, "wg=") == 0) {
            const int c = std::atoi(tok.c_str() + 3) / 64 * 64;
            if (c < 64 || c > 1024) return HEVCDBK_ERR_ARG;
            d.wg_cap = c;
        } else return HEVCDBK_ERR_ARG;
    }
    g_dbk_diag = d;
    return HEVCDBK_OK;
}
#endif

/*
 * Put the frame's bS arrays (luma vert | luma hor | chroma vert | chroma hor) into ctx->dev_bs on stream `s`.
 * Caller-supplied arrays are uploaded every call; the reference's default pattern (cpu.h:92-99) is built and
 * uploaded once per geometry and stays resident (the reference re-uploads it per call, gpu.cu:1246-1249).
 */
int stage_bs(hevcdbk_context *ctx, unsigned W, unsigned H, bool chroma, const hevcdbk_bs *bs, hipStream_t s, bool *uploaded = nullptr)
{
    if (uploaded) *uploaded = false;
    const size_t nv = hevcdbk_num_vert_bs(W, H), nh = hevcdbk_num_hor_bs(W, H);
    const size_t ncv = chroma ? hevcdbk_num_vert_bs(W / 2, H / 2) : 0, nch = chroma ? hevcdbk_num_hor_bs(W / 2, H / 2) : 0;
    const size_t bs_bytes = nv + nh + ncv + nch;
    if (int rc = grow_device(ctx, ctx->dev_bs, bs_bytes)) return rc;
    const bool user = bs && (bs->vert || bs->chroma_vert);
    if (!user && ctx->bs_default_at == ctx->dev_bs.p && ctx->bs_default_w == W && ctx->bs_default_h == H &&
        ctx->bs_default_chroma == chroma)
        return HEVCDBK_OK;
    /* the pinned copy may still feed an earlier call's upload */
    HIP_TRY(ctx, hipStreamSynchronize(s));
    if (int rc = grow_pinned(ctx, ctx->pin_bs, bs_bytes)) return rc;
    uint8_t *hbs = (uint8_t *)ctx->pin_bs.p;
    if (bs && bs->vert) { std::memcpy(hbs, bs->vert, nv); std::memcpy(hbs + nv, bs->hor, nh); }
    else hevcdbk_default_bs(W, H, hbs, hbs + nv);
    if (chroma) {
        if (bs && bs->chroma_vert) { std::memcpy(hbs + nv + nh, bs->chroma_vert, ncv); std::memcpy(hbs + nv + nh + ncv, bs->chroma_hor, nch); }
        else hevcdbk_default_bs(W / 2, H / 2, hbs + nv + nh, hbs + nv + nh + ncv);
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dev_bs.p, hbs, bs_bytes, hipMemcpyHostToDevice, s));
    if (uploaded) *uploaded = true;
    ctx->bs_default_at = user ? nullptr : ctx->dev_bs.p;
    ctx->bs_default_w = W; ctx->bs_default_h = H; ctx->bs_default_chroma = chroma;
    return HEVCDBK_OK;
}

/* args of plane k of a tightly packed frame whose planes sit at dplane[k] and whose bS sit in ctx->dev_bs */
int frame_plane_args(hevcdbk_context *ctx, void *const dplane[3], int k, unsigned W, unsigned H, unsigned bit_depth,
                     unsigned sb, bool chroma, const hevcdbk_qp *qp, const uint8_t *dmap, const hevcdbk_tables *tables,
                     DbkArgs &a, const size_t *pitch = nullptr /* NULL: tightly packed rows */)
{
    const size_t nv = hevcdbk_num_vert_bs(W, H), nh = hevcdbk_num_hor_bs(W, H);
    const size_t ncv = chroma ? hevcdbk_num_vert_bs(W / 2, H / 2) : 0;
    const uint8_t *dbs = (const uint8_t *)ctx->dev_bs.p;
    const unsigned pw = k ? W / 2 : W, ph = k ? H / 2 : H;
    hevcdbk_device_planes p;
    std::memset(&p, 0, sizeof(p));
    p.src = dplane[k]; p.dst = dplane[k];
    p.pitch = pitch ? pitch[k] : (size_t)pw * sb; p.frame_stride = p.pitch * ph; p.n_frames = 1;
    p.plane_w = pw; p.plane_h = ph; p.bit_depth = bit_depth; p.sample_bytes = sb;
    p.is_chroma = k != 0;
    p.vert_bs = k == 0 ? dbs : dbs + nv + nh;
    p.hor_bs = k == 0 ? dbs + nv : dbs + nv + nh + ncv;
    p.qp_map = dmap; p.qp_map_stride = qp->map_stride; p.ctu_log2 = qp->ctu_log2;
    return planes_to_args(&p, qp->qp, tables, a);
}

/* true (and the launch is done) when the whole frame went out as ONE fused launch */
int launch_frame_fused(hevcdbk_context *ctx, const DbkArgs *args, int npl, unsigned sb, hipStream_t s, bool *done)
{
    const int sbs[3] = {(int)sb, (int)sb, (int)sb};
    *done = false;
    if (!dbk_multi_supports(args, npl, sbs)) return HEVCDBK_OK;
    if (!hip_ok(ctx, dbk_launch_packed_multi(args, npl, (int)sb, s), "kernel launch")) return HEVCDBK_ERR_HIP;
    *done = true;
    return HEVCDBK_OK;
}

/* threads that copy during a large pageable frame, the caller included: what the context was told (hevcdbk_set_host_threads), else 4
 * (tools/ubench/host_stage.hip: the box's staging rate stops growing between 4 and 8 threads).  The product reads no environment;
 * the diagnostic build takes HEVCDBK_HOST_THREADS / _STREAM_STORES / _AFFINITY for A/B runs. */
bool crew_knob(const char *name, bool dflt)
{
#ifdef HEVCDBK_DIAG
    if (const char *e = std::getenv(name)) return std::strtol(e, nullptr, 10) != 0;
#else
    (void)name;
#endif
    return dflt;
}

/* the strip size of the large-frame pipeline (diagnostic build: HEVCDBK_HOST_STRIP_KB overrides, for sweeps) */
size_t crew_strip_bytes(size_t dflt)
{
#ifdef HEVCDBK_DIAG
    if (const char *e = std::getenv("HEVCDBK_HOST_STRIP_KB")) {
        const size_t kb = (size_t)std::strtoul(e, nullptr, 10);
        if (kb >= 64) return kb << 10;
    }
#endif
    return dflt;
}

size_t crew_first_strip_bytes(size_t dflt)
{
#ifdef HEVCDBK_DIAG
    if (const char *e = std::getenv("HEVCDBK_HOST_FIRST_STRIP_KB")) {
        const size_t kb = (size_t)std::strtoul(e, nullptr, 10);
        if (kb >= 32) return kb << 10;
    }
#endif
    return dflt;
}

unsigned crew_size(const hevcdbk_context *ctx)
{
    unsigned n = ctx->host_threads;
#ifdef HEVCDBK_DIAG
    if (n == 0)
        if (const char *e = std::getenv("HEVCDBK_HOST_THREADS")) n = (unsigned)std::strtoul(e, nullptr, 10);
#endif
    if (n == 0) n = 4;
    return n > 64 ? 64 : n;
}

/* The CPUs on the GPU's side of the host (sysfs local_cpulist of its PCI function) that this process may run on: the page-locked
 * ring lives in that socket's memory (hipHostMalloc allocates next to the device), so the crew copies into / out of local DRAM and the
 * DMA does not cross the socket link.  false = unknown: no restriction. */
bool cpus_near_device(int device, cpu_set_t *out)
{
    if (!crew_knob("HEVCDBK_HOST_AFFINITY", true)) return false;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    for (char *c = bus; *c; c++) *c = (char)std::tolower((unsigned char)*c);
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/local_cpulist";
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return false;
    char line[4096] = {0};
    const bool got = std::fgets(line, sizeof line, f) != nullptr;
    std::fclose(f);
    if (!got) return false;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
    CPU_ZERO(out);
    int n = 0;
    for (char *p = line; *p && *p != '\n';) { /* "0-63,128-191" */
        char *end = nullptr;
        long a = std::strtol(p, &end, 10), b = a;
        if (end == p) break;
        if (*end == '-') { p = end + 1; b = std::strtol(p, &end, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (c >= 0 && CPU_ISSET((int)c, &allowed)) { CPU_SET((int)c, out); n++; }
        p = *end == ',' ? end + 1 : end;
        if (*end != ',' ) break;
    }
    return n > 0;
}

int ensure_crew(hevcdbk_context *ctx)
{
    const unsigned n = crew_size(ctx);
    if (ctx->crew && ctx->crew->workers() + 1 == n) return HEVCDBK_OK;
    delete ctx->crew;
    ctx->crew = nullptr;
    const bool stream = crew_knob("HEVCDBK_HOST_STREAM_STORES", true);
    cpu_set_t near;
    const bool have_near = cpus_near_device(ctx->device, &near);
    try {
        ctx->crew = new StageCrew(n - 1, stream, have_near ? &near : nullptr);
    } catch (...) { /* thread creation failed */
        ctx->crew = nullptr;
        return HEVCDBK_ERR_NOMEM;
    }
    return HEVCDBK_OK;
}

uint8_t *push_buffer(hevcdbk_context *ctx, size_t bytes)
{
    if (ctx->large_bar < 0) {
        int v = 0;
        ctx->large_bar = hipDeviceGetAttribute(&v, hipDeviceAttributeIsLargeBar, ctx->device) == hipSuccess && v ? 1 : 0;
        (void)hipGetLastError();
    }
    if (ctx->large_bar != 1) return nullptr;
    if (ctx->dev_push.cap < bytes) {
        if (ctx->dev_push.p) (void)hipFree(ctx->dev_push.p);
        ctx->dev_push.p = nullptr;
        ctx->dev_push.cap = 0;
        if (hipExtMallocWithFlags(&ctx->dev_push.p, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            ctx->dev_push.p = nullptr;
            return nullptr;
        }
        /* "large BAR" says the card's memory CAN be mapped for the host; before the crew stores through the pointer, make sure THIS
         * allocation is mapped into the process (msync fails with ENOMEM on an address range that has no mapping) -- a store to
         * an unmapped address would be a SIGSEGV in the caller's process, the DMA path a few hundred microseconds */
        {
            const long pg = sysconf(_SC_PAGESIZE);
            const uintptr_t a0 = (uintptr_t)ctx->dev_push.p & ~(uintptr_t)(pg - 1);
            const uintptr_t a1 = ((uintptr_t)ctx->dev_push.p + bytes - 1) & ~(uintptr_t)(pg - 1);
            if (msync((void *)a0, (size_t)pg, MS_ASYNC) != 0 || msync((void *)a1, (size_t)pg, MS_ASYNC) != 0) {
                (void)hipFree(ctx->dev_push.p);
                ctx->dev_push.p = nullptr;
                ctx->large_bar = 0; /* ring + DMA from here on */
                return nullptr;
            }
        }
        ctx->dev_push.cap = bytes;
    }
    return (uint8_t *)ctx->dev_push.p;
}

int crew_copy_frame(hevcdbk_context *ctx, const hevcdbk_frame *frame, int npl, const unsigned *pw, const unsigned *ph, unsigned sb,
                    uint8_t *base, const size_t *plane_off, bool to_frame, bool base_is_device)
{
    size_t total = 0;
    for (int i = 0; i < npl; i++) total += (size_t)pw[i] * ph[i] * sb;
    /* below 1 MiB waking the crew costs more than it saves: the calling thread copies alone */
    const bool threads = total >= ((size_t)1 << 20);
    if (threads)
        if (int rc = ensure_crew(ctx)) return rc;
    StageCrew *crew = threads ? ctx->crew : nullptr;
    const unsigned crew_n = crew && crew->workers() ? crew->workers() + 1 : 1;
    CopyGroup g;
    std::vector<CopyJob> jobs;
    for (int i = 0; i < npl; i++) {
        const size_t rb = (size_t)pw[i] * sb;
        unsigned pieces = (unsigned)((rb * ph[i]) / ((size_t)256 << 10));
        pieces = pieces < 1 ? 1 : pieces > 4 * crew_n ? 4 * crew_n : pieces;
        for (unsigned p = 0; p < pieces; p++) {
            const unsigned a = (unsigned)((uint64_t)ph[i] * p / pieces), b = (unsigned)((uint64_t)ph[i] * (p + 1) / pieces);
            if (a == b) continue;
            uint8_t *inner = base + plane_off[i] + (size_t)a * rb, *user = (uint8_t *)frame->plane[i] + (size_t)a * frame->pitch[i];
            if (to_frame) jobs.push_back({user, inner, frame->pitch[i], rb, rb, b - a, &g, false});
            else jobs.push_back({inner, user, rb, frame->pitch[i], rb, b - a, &g, base_is_device});
        }
    }
    if (!crew) {
        for (const CopyJob &j : jobs) copy_rows(j, true);
        return HEVCDBK_OK;
    }
    g.pending.store((int)jobs.size(), std::memory_order_release);
    crew->begin();
    for (const CopyJob &j : jobs) crew->submit(j, to_frame ? StageCrew::LANE_OUT : StageCrew::LANE_IN);
    crew->wait(g); /* this thread copies too */
    crew->end();
    return HEVCDBK_OK;
}

} /* namespace dbkh */

using namespace dbkh;

/* ------------------------------------------------------------------------------------------ */

extern "C" {

const char *hevcdbk_strerror(int code)
{
    switch (code) {
    case HEVCDBK_OK: return "ok";
    case HEVCDBK_ERR_FILE_SIZE: return "Incorrect file size";
    case HEVCDBK_ERR_DIMENSIONS: return "Width and height of image must be multiplier of sample block size";
    case HEVCDBK_ERR_BS_SIZE: return "Incorrect size of input boundary strenght array";
    case HEVCDBK_ERR_HIP: return "HIP runtime failure (see hevcdbk_last_error)";
    case HEVCDBK_ERR_ARG: return "invalid argument";
    case HEVCDBK_ERR_NOMEM: return "out of memory";
    case HEVCDBK_ERR_IO: return "file i/o failure";
    case HEVCDBK_ERR_UNSUPPORTED: return "unsupported operand layout for the HIP kernels";
    default: return "unknown error";
    }
}

int hevcdbk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int hevcdbk_create(int device, hevcdbk_context **out)
{
    if (!out) return HEVCDBK_ERR_ARG;
    *out = nullptr;
    int n = hevcdbk_device_count();
    if (n <= 0 || device < 0 || device >= n) return HEVCDBK_ERR_HIP; /* no CPU fallback, by design */
    hevcdbk_context *ctx = new (std::nothrow) hevcdbk_context;
    if (!ctx) return HEVCDBK_ERR_NOMEM;
    ctx->device = device;
    if (!hip_ok(ctx, hipSetDevice(device), "hipSetDevice") ||
        !hip_ok(ctx, hipStreamCreateWithFlags(&ctx->compute, hipStreamNonBlocking), "hipStreamCreate") ||
        !hip_ok(ctx, hipStreamCreateWithFlags(&ctx->h2d, hipStreamNonBlocking), "hipStreamCreate") ||
        !hip_ok(ctx, hipStreamCreateWithFlags(&ctx->d2h, hipStreamNonBlocking), "hipStreamCreate")) {
        hevcdbk_destroy(ctx);
        return HEVCDBK_ERR_HIP;
    }
    for (auto &e : ctx->ev)
        if (!hip_ok(ctx, hipEventCreate(&e), "hipEventCreate")) {
            hevcdbk_destroy(ctx);
            return HEVCDBK_ERR_HIP;
        }
    if (hipDeviceGetAttribute(&ctx->n_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) ctx->n_cus = 0;
    *out = ctx;
    return HEVCDBK_OK;
}

void hevcdbk_destroy(hevcdbk_context *ctx)
{
    if (!ctx) return;
    delete ctx->crew; /* joins the crew's threads (asleep between calls) */
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (void *p : ctx->registered) (void)hipHostUnregister(p);
    for (auto &g : ctx->pin) if (g.p) (void)hipHostFree(g.p);
    for (auto &g : ctx->dev) if (g.p) (void)hipFree(g.p);
    if (ctx->pin_bs.p) (void)hipHostFree(ctx->pin_bs.p);
    if (ctx->dev_bs.p) (void)hipFree(ctx->dev_bs.p);
    if (ctx->dev_map.p) (void)hipFree(ctx->dev_map.p);
    if (ctx->dev_units.p) (void)hipFree(ctx->dev_units.p);
    if (ctx->dev_tmp.p) (void)hipFree(ctx->dev_tmp.p);
    if (ctx->dev_push.p) (void)hipFree(ctx->dev_push.p);
    if (ctx->tmp_ev) (void)hipEventDestroy(ctx->tmp_ev);
    for (auto &e : ctx->ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : ctx->timed_events) if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < hevcdbk_context::kSeqSlots; k++)
        for (int i = 0; i < 3; i++) {
            if (ctx->seq_pin[k][i].p) (void)hipHostFree(ctx->seq_pin[k][i].p);
            if (ctx->seq_dev[k][i].p) (void)hipFree(ctx->seq_dev[k][i].p);
            if (ctx->seq_ev[k][i]) (void)hipEventDestroy(ctx->seq_ev[k][i]);
        }
    if (ctx->compute) (void)hipStreamDestroy(ctx->compute);
    if (ctx->h2d) (void)hipStreamDestroy(ctx->h2d);
    if (ctx->d2h) (void)hipStreamDestroy(ctx->d2h);
    delete ctx;
}

const char *hevcdbk_last_error(const hevcdbk_context *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int hevcdbk_get_device_info(const hevcdbk_context *ctx, hevcdbk_device_info *info)
{
    if (!ctx || !info) return HEVCDBK_ERR_ARG;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, ctx->device) != hipSuccess) return HEVCDBK_ERR_HIP;
    std::memset(info, 0, sizeof(*info));
    std::snprintf(info->name, sizeof(info->name), "%s", p.name);
    std::snprintf(info->gcn_arch, sizeof(info->gcn_arch), "%s", p.gcnArchName);
    info->compute_units = p.multiProcessorCount;
    info->wavefront_size = p.warpSize;
    info->max_threads_per_block = p.maxThreadsPerBlock;
    info->total_global_mem = p.totalGlobalMem;
    info->shared_mem_per_block = p.sharedMemPerBlock;
    info->total_const_mem = p.totalConstMem;
    return HEVCDBK_OK;
}

const unsigned *hevcdbk_default_tc_table(void) { return k_tc_table; }
const unsigned *hevcdbk_default_beta_table(void) { return k_beta_table; }

/* cpu.h:86 / 104 */
size_t hevcdbk_num_vert_bs(unsigned w, unsigned h) { return (size_t)(w / 8 + 1) * h / 8; }
/* cpu.h:87 / 105 */
size_t hevcdbk_num_hor_bs(unsigned w, unsigned h) { return (size_t)(h / 8 + 1) * w / 8; }

int hevcdbk_default_bs(unsigned w, unsigned h, uint8_t *vert_bs, uint8_t *hor_bs)
{
    if (!vert_bs || !hor_bs) return HEVCDBK_ERR_ARG;
    if (w == 0 || h == 0 || w % 8 != 0 || h % 8 != 0) return HEVCDBK_ERR_DIMENSIONS;
    const size_t nv = hevcdbk_num_vert_bs(w, h), nh = hevcdbk_num_hor_bs(w, h);
    for (size_t i = 0; i < nv; i++) vert_bs[i] = (i % (w / 8 + 1) == 0) ? 0 : 2; /* cpu.h:92-95 */
    for (size_t i = 0; i < nh; i++) hor_bs[i] = (i % (h / 8 + 1) == 0) ? 0 : 2;  /* cpu.h:96-99, stride quirk Q3 */
    return HEVCDBK_OK;
}

/* ---- plumbing ---- */

int hevcdbk_device_malloc(hevcdbk_context *ctx, size_t bytes, void **dptr)
{
    if (!ctx || !dptr) return HEVCDBK_ERR_ARG;
    if (int rc = bind(ctx)) return rc;
    HIP_TRY(ctx, hipMalloc(dptr, bytes ? bytes : 1));
    return HEVCDBK_OK;
}
int hevcdbk_device_free(hevcdbk_context *ctx, void *dptr)
{
    if (!ctx) return HEVCDBK_ERR_ARG;
    if (int rc = bind(ctx)) return rc;
    HIP_TRY(ctx, hipFree(dptr));
    return HEVCDBK_OK;
}
int hevcdbk_host_malloc_pinned(hevcdbk_context *ctx, size_t bytes, void **hptr)
{
    if (!ctx || !hptr) return HEVCDBK_ERR_ARG;
    if (int rc = bind(ctx)) return rc;
    HIP_TRY(ctx, hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault));
    return HEVCDBK_OK;
}
int hevcdbk_host_free_pinned(hevcdbk_context *ctx, void *hptr)
{
    if (!ctx) return HEVCDBK_ERR_ARG;
    if (int rc = bind(ctx)) return rc;
    HIP_TRY(ctx, hipHostFree(hptr));
    return HEVCDBK_OK;
}
int hevcdbk_host_register(hevcdbk_context *ctx, void *ptr, size_t bytes)
{
    if (!ctx || !ptr || bytes == 0) return HEVCDBK_ERR_ARG;
    if (int rc = bind(ctx)) return rc;
    HIP_TRY(ctx, hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    ctx->registered.push_back(ptr);
    return HEVCDBK_OK;
}
int hevcdbk_host_unregister(hevcdbk_context *ctx, void *ptr)
{
    if (!ctx || !ptr) return HEVCDBK_ERR_ARG;
    if (int rc = bind(ctx)) return rc;
    for (size_t i = 0; i < ctx->registered.size(); i++)
        if (ctx->registered[i] == ptr) {
            /* transfers of the host-frame operators have completed when they return; the streaming operator synchronises too */
            HIP_TRY(ctx, hipHostUnregister(ptr));
            ctx->registered.erase(ctx->registered.begin() + (long)i);
            return HEVCDBK_OK;
        }
    return HEVCDBK_ERR_ARG; /* not registered through this context */
}
int hevcdbk_set_host_threads(hevcdbk_context *ctx, unsigned n_threads)
{
    if (!ctx || n_threads > 64) return HEVCDBK_ERR_ARG;
    ctx->host_threads = n_threads; /* the crew is (re)built by the next large pageable frame */
    return HEVCDBK_OK;
}
unsigned hevcdbk_get_host_threads(const hevcdbk_context *ctx) { return ctx ? crew_size(ctx) : 0; }
int hevcdbk_last_frame_trace(const hevcdbk_context *ctx, hevcdbk_strip_trace *out, unsigned cap, unsigned *n_strips)
{
    if (!ctx || (!out && cap)) return HEVCDBK_ERR_ARG;
    const size_t n = ctx->trace.size();
    for (size_t i = 0; i < n && i < cap; i++) out[i] = ctx->trace[i];
    if (n_strips) *n_strips = (unsigned)n;
    return HEVCDBK_OK;
}
int hevcdbk_memcpy_h2d(hevcdbk_context *ctx, void *dptr, const void *hptr, size_t bytes)
{
    if (!ctx) return HEVCDBK_ERR_ARG;
    if (int rc = bind(ctx)) return rc;
    HIP_TRY(ctx, hipMemcpy(dptr, hptr, bytes, hipMemcpyHostToDevice));
    return HEVCDBK_OK;
}
int hevcdbk_memcpy_d2h(hevcdbk_context *ctx, void *hptr, const void *dptr, size_t bytes)
{
    if (!ctx) return HEVCDBK_ERR_ARG;
    if (int rc = bind(ctx)) return rc;
    HIP_TRY(ctx, hipMemcpy(hptr, dptr, bytes, hipMemcpyDeviceToHost));
    return HEVCDBK_OK;
}
int hevcdbk_memcpy_d2d(hevcdbk_context *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx) return HEVCDBK_ERR_ARG;
    if (int rc = bind(ctx)) return rc;
    HIP_TRY(ctx, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice));
    return HEVCDBK_OK;
}
int hevcdbk_memset_d(hevcdbk_context *ctx, void *dptr, int value, size_t bytes)
{
    if (!ctx) return HEVCDBK_ERR_ARG;
    if (int rc = bind(ctx)) return rc;
    HIP_TRY(ctx, hipMemset(dptr, value, bytes));
    return HEVCDBK_OK;
}
int hevcdbk_synchronize(hevcdbk_context *ctx)
{
    if (!ctx) return HEVCDBK_ERR_ARG;
    if (int rc = bind(ctx)) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->h2d));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->compute));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->d2h));
    return HEVCDBK_OK;
}
void *hevcdbk_compute_stream(hevcdbk_context *ctx) { return ctx ? (void *)ctx->compute : nullptr; }

/* ---- device-resident operator ---- */

int hevc_deblocking_filter_device(hevcdbk_context *ctx, const hevcdbk_device_planes *planes, unsigned qp,
                                  const hevcdbk_tables *tables, int kernel_variant, void *hip_stream)
{
    if (!ctx) return HEVCDBK_ERR_ARG;
    DbkArgs a;
    if (int rc = planes_to_args(planes, qp, tables, a)) return rc;
    if (int rc = bind(ctx)) return rc;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->compute;
    return launch(ctx, a, (int)planes->sample_bytes, planes->is_chroma != 0, kernel_variant, s);
}

/* would launch() accept this plane?  (the checks launch() makes before it enqueues anything) */
static int check_launch(const DbkArgs &a, int sample_bytes, bool chroma, int variant)
{
    const int map = variant & HEVCDBK_MAP_MASK, fam = variant & ~HEVCDBK_MAP_MASK;
#ifdef HEVCDBK_DIAG
    if (map != HEVCDBK_MAP_AUTO && map != HEVCDBK_MAP_ROWS && map != HEVCDBK_MAP_LINEAR && map != HEVCDBK_DIAG_MAP_TILES && map != HEVCDBK_DIAG_MAP_STRIPE && map != HEVCDBK_DIAG_MAP_PIPE && map != HEVCDBK_DIAG_MAP_GROUP) return HEVCDBK_ERR_ARG;
    if (fam == HEVCDBK_DIAG_KERNEL_COPY) return dbk_packed_supports(a, sample_bytes, chroma) ? HEVCDBK_OK : HEVCDBK_ERR_UNSUPPORTED;
#else
    if (map != HEVCDBK_MAP_AUTO && map != HEVCDBK_MAP_ROWS && map != HEVCDBK_MAP_LINEAR) return HEVCDBK_ERR_ARG;
#endif
    if (fam == HEVCDBK_KERNEL_PACKED) return dbk_packed_supports(a, sample_bytes, chroma) ? HEVCDBK_OK : HEVCDBK_ERR_UNSUPPORTED;
    if (fam == HEVCDBK_KERNEL_GENERIC || fam == HEVCDBK_KERNEL_AUTO) return HEVCDBK_OK;
    return HEVCDBK_ERR_ARG;
}

/* true when the planes of a batch go out as ONE fused launch (dbk_packed_multi_kernel / dbk_packed16_multi_kernel) */
static bool planes_fuse(const hevcdbk_device_planes *planes, const DbkArgs *args, unsigned n_planes, int kernel_variant, int *sb0)
{
    const int fam = kernel_variant & ~HEVCDBK_MAP_MASK, map = kernel_variant & HEVCDBK_MAP_MASK;
    if (n_planes < 2 || n_planes > 3 || map != HEVCDBK_MAP_AUTO || (fam != HEVCDBK_KERNEL_AUTO && fam != HEVCDBK_KERNEL_PACKED) ||
        planes[0].is_chroma)
        return false;
    int sbs[3] = {0, 0, 0};
    for (unsigned i = 0; i < n_planes; i++) {
        sbs[i] = (int)planes[i].sample_bytes;
        if (!((i == 0 || planes[i].is_chroma) && dbk_packed_supports(args[i], sbs[i], planes[i].is_chroma != 0))) return false;
    }
    *sb0 = sbs[0];
    return dbk_multi_supports(args, (int)n_planes, sbs);
}

/* the planes of a batch: one fused launch where the fused kernel applies, else plane by plane.  Every plane is checked
 * BEFORE the first launch, so an unsupported operand returns its error with nothing enqueued (no partly written dst).
 * ev_start / ev_stop (may be NULL): the step's first launch stamps its own begin into ev_start and its last launch its own
 * end into ev_stop -- kernel time as a profiler's kernel trace sees it, no barrier packets between the launches. */
static int launch_planes(hevcdbk_context *ctx, const hevcdbk_device_planes *planes, const DbkArgs *args, unsigned n_planes,
                         int kernel_variant, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr)
{
    int sb0 = 0;
    if (planes_fuse(planes, args, n_planes, kernel_variant, &sb0)) {
        dbk_set_next_launch_events(ev_start, ev_stop);
        const hipError_t e = dbk_launch_packed_multi(args, (int)n_planes, sb0, s);
        dbk_set_next_launch_events(nullptr, nullptr);
        return hip_ok(ctx, e, "kernel launch") ? HEVCDBK_OK : HEVCDBK_ERR_HIP;
    }
    for (unsigned i = 0; i < n_planes; i++)
        if (int rc = check_launch(args[i], (int)planes[i].sample_bytes, planes[i].is_chroma != 0, kernel_variant)) return rc;
    for (unsigned i = 0; i < n_planes; i++) {
        dbk_set_next_launch_events(i == 0 ? ev_start : nullptr, i + 1 == n_planes ? ev_stop : nullptr);
        const int rc = launch(ctx, args[i], (int)planes[i].sample_bytes, planes[i].is_chroma != 0, kernel_variant, s);
        dbk_set_next_launch_events(nullptr, nullptr);
        if (rc) return rc;
    }
    return HEVCDBK_OK;
}

int hevc_deblocking_filter_device_planes(hevcdbk_context *ctx, const hevcdbk_device_planes *planes, unsigned n_planes,
                                         unsigned qp, const hevcdbk_tables *tables, int kernel_variant, void *hip_stream)
{
    if (!ctx || !planes || n_planes == 0 || n_planes > 3) return HEVCDBK_ERR_ARG;
    DbkArgs args[3];
    for (unsigned i = 0; i < n_planes; i++) {
        if (int rc = planes_to_args(&planes[i], qp, tables, args[i])) return rc;
        if (planes[i].n_frames != planes[0].n_frames) return HEVCDBK_ERR_ARG;
    }
    if (int rc = bind(ctx)) return rc;
    return launch_planes(ctx, planes, args, n_planes, kernel_variant, hip_stream ? (hipStream_t)hip_stream : ctx->compute);
}

static int ensure_timed_events(hevcdbk_context *ctx, size_t need)
{
    while (ctx->timed_events.size() < need) {
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreate(&e));
        ctx->timed_events.push_back(e);
    }
    return HEVCDBK_OK;
}

static double monotonic_s()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int hevcdbk_device_run_timed(hevcdbk_context *ctx, const hevcdbk_device_planes *planes, unsigned n_planes,
                             unsigned qp, const hevcdbk_tables *tables, int kernel_variant, unsigned steps,
                             float *kernel_ms)
{
    if (!ctx || !planes || n_planes == 0 || n_planes > 3 || !kernel_ms) return HEVCDBK_ERR_ARG;
    hevcdbk_replay r;
    std::memset(&r, 0, sizeof(r));
    r.steps = steps;
    return hevcdbk_device_replay(ctx, planes, n_planes, qp, tables, kernel_variant, &r, kernel_ms);
}

/*
 * One uninterrupted stream of launches: [settle ...][warm-up x W][timed x K], one synchronisation at the very end.
 *
 * Settling is by TIME, not by count: the clock governor works in milliseconds (the card idles at a few hundred MHz and needs
 * tens of milliseconds under load to reach the state it then holds; under this VALU-heavy kernel it ends at the socket power
 * cap).  Launches are issued until the mean duration of the trailing `settle_window` launches lies within
 * `settle_tolerance` of the mean of the window before it, but for no less than settle_min_ms and no more than settle_max_ms.
 * The host reads the per-launch durations from events of launches that have already completed while it stays up to
 * kAhead launches ahead of the GPU, so the queue never drains between the phases.
 */
int hevcdbk_device_replay(hevcdbk_context *ctx, const hevcdbk_device_planes *planes, unsigned n_planes, unsigned qp,
                          const hevcdbk_tables *tables, int kernel_variant, hevcdbk_replay *r, float *kernel_ms)
{
    if (!ctx || !planes || n_planes == 0 || n_planes > 3 || !r || (r->steps && !kernel_ms)) return HEVCDBK_ERR_ARG;
    if (r->settle_min_ms < 0 || r->settle_max_ms < r->settle_min_ms || r->settle_tolerance < 0) return HEVCDBK_ERR_ARG;
    DbkArgs args[3];
    for (unsigned i = 0; i < n_planes; i++) {
        if (int rc = planes_to_args(&planes[i], qp, tables, args[i])) return rc;
        if (planes[i].n_frames != planes[0].n_frames) return HEVCDBK_ERR_ARG;
    }
    if (int rc = bind(ctx)) return rc;
    constexpr unsigned kRing = 256, kAhead = 48;
    const unsigned steps = r->steps, warm = r->warmup;
    /* events: [0, 2*kRing) the settle ring, then one pair per warm-up and timed launch */
    if (int rc = ensure_timed_events(ctx, 2 * (size_t)kRing + 2 * ((size_t)warm + steps))) return rc;
    hipEvent_t *ev = ctx->timed_events.data();
    const unsigned W = r->settle_window ? r->settle_window : 32;
    const double tol = r->settle_tolerance > 0 ? r->settle_tolerance : 0.005;

    r->settle_launches = 0; r->settle_ms = 0; r->settled = 0; r->settle_tail_mean_ms = 0;
    std::vector<float> hist; /* durations of completed settle launches, in launch order */
    unsigned launched = 0, read = 0;
    bool stop = !(r->settle_max_ms > 0);
    /* with neither settling nor warm-up nothing precedes the timed launches inside this call, and the front bracket is the
     * moment they are handed over: work an earlier asynchronous call left on the compute stream must not be inside the
     * bracket, so the stream is drained first -- its queue is empty at t_first */
    if (stop && warm == 0) HIP_TRY(ctx, hipStreamSynchronize(ctx->compute));
    const double t_first = monotonic_s();
    auto harvest = [&](bool block) -> int {
        while (read < launched) {
            hipEvent_t e0 = ev[2 * (read % kRing)], e1 = ev[2 * (read % kRing) + 1];
            if (block) HIP_TRY(ctx, hipEventSynchronize(e1));
            else {
                const hipError_t q = hipEventQuery(e1);
                if (q == hipErrorNotReady) { (void)hipGetLastError(); break; }
                if (!hip_ok(ctx, q, "hipEventQuery")) return HEVCDBK_ERR_HIP;
            }
            float ms = 0;
            HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
            hist.push_back(ms);
            read++;
            block = false;
        }
        return HEVCDBK_OK;
    };
    while (!stop) {
        if (launched - read >= kAhead) { if (int rc = harvest(true)) return rc; }
        if (int rc = launch_planes(ctx, planes, args, n_planes, kernel_variant, ctx->compute, ev[2 * (launched % kRing)],
                                   ev[2 * (launched % kRing) + 1]))
            return rc;
        launched++;
        if (int rc = harvest(false)) return rc;
        const double el_ms = (monotonic_s() - t_first) * 1e3;
        if (el_ms >= r->settle_max_ms) stop = true;
        else if (el_ms >= r->settle_min_ms && hist.size() >= 2 * (size_t)W) {
            double a = 0, b = 0;
            for (unsigned i = 0; i < W; i++) { a += hist[hist.size() - 1 - i]; b += hist[hist.size() - 1 - W - i]; }
            if (std::fabs(a - b) <= tol * b) { stop = true; r->settled = 1; }
        }
    }
    r->settle_launches = launched;
    r->settle_ms = launched ? (monotonic_s() - t_first) * 1e3 : 0.0;

    /* warm-up and timed launches go out behind the settle launches still in flight */
    hipEvent_t *tev = ev + 2 * (size_t)kRing;
    for (unsigned s = 0; s < warm + steps; s++)
        if (int rc = launch_planes(ctx, planes, args, n_planes, kernel_variant, ctx->compute, tev[2 * s], tev[2 * s + 1])) return rc;
    /* front bracket of the timed window: the host observes the end of the launch BEFORE the first timed one while the timed
     * launches are already queued behind it (no idle gap on the GPU); with nothing in front of them, a stream
     * synchronisation before the first launch would be the bracket -- the queue is empty then anyway */
    if (warm > 0) HIP_TRY(ctx, hipEventSynchronize(tev[2 * (warm - 1) + 1]));
    else if (launched > 0) HIP_TRY(ctx, hipEventSynchronize(ev[2 * ((launched - 1) % kRing) + 1]));
    r->t_begin = (warm > 0 || launched > 0) ? monotonic_s() : t_first;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->compute));
    r->t_end = monotonic_s();
    r->wall_ms = (r->t_end - r->t_begin) * 1e3;
    for (unsigned s = 0; s < steps; s++)
        HIP_TRY(ctx, hipEventElapsedTime(&kernel_ms[s], tev[2 * (warm + s)], tev[2 * (warm + s) + 1]));
    r->span_ms = 0;
    if (steps) {
        float sp = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&sp, tev[2 * warm], tev[2 * (warm + steps - 1) + 1]));
        r->span_ms = sp;
    }
    if (int rc = harvest(true)) return rc; /* everything has completed: the tail of the settle phase */
    if (!hist.empty()) {
        const size_t n = hist.size() < W ? hist.size() : W;
        double a = 0;
        for (size_t i = 0; i < n; i++) a += hist[hist.size() - 1 - i];
        r->settle_tail_mean_ms = a / (double)n;
    }
    return HEVCDBK_OK;
}

int hevcdbk_device_malloc_probed(hevcdbk_context *ctx, const hevcdbk_device_planes *probe, unsigned qp, const hevcdbk_tables *tables,
                                 unsigned candidates, void **dptr, float *best_ms, float *worst_ms)
{
    if (!ctx || !probe || !dptr || candidates == 0 || candidates > 16 || !probe->src || probe->n_frames == 0) return HEVCDBK_ERR_ARG;
    *dptr = nullptr;
    const size_t bytes = probe->frame_stride * (size_t)(probe->n_frames - 1) + probe->pitch * (size_t)probe->plane_h;
    if (bytes == 0) return HEVCDBK_ERR_ARG;
    if (int rc = bind(ctx)) return rc;
    void *cand[16] = {};
    float t[16];
    int rc = HEVCDBK_OK;
    unsigned n = 0;
    for (; n < candidates; n++) {
        if (hipMalloc(&cand[n], bytes) != hipSuccess) { /* fewer candidates than asked for is not an error as long as there is one */
            (void)hipGetLastError();
            cand[n] = nullptr;
            break;
        }
    }
    if (n == 0) return HEVCDBK_ERR_NOMEM;
    /* the card idles at a few hundred MHz: launches go out for 100-200 ms first (into candidate 0), so that the first candidates
     * are not timed at a lower clock than the last; then two passes over the candidates, the best of three launches each */
    {
        hevcdbk_device_planes p = *probe;
        p.dst = cand[0];
        hevcdbk_replay r;
        std::memset(&r, 0, sizeof(r));
        r.settle_min_ms = 100.0;
        r.settle_max_ms = 200.0;
        rc = hevcdbk_device_replay(ctx, &p, 1, qp, tables, HEVCDBK_KERNEL_AUTO, &r, nullptr);
    }
    for (unsigned k = 0; k < n; k++) t[k] = 1e30f;
    for (int pass = 0; pass < 2 && rc == HEVCDBK_OK; pass++)
        for (unsigned k = 0; k < n && rc == HEVCDBK_OK; k++) {
            hevcdbk_device_planes p = *probe;
            p.dst = cand[k];
            hevcdbk_replay r;
            std::memset(&r, 0, sizeof(r));
            r.warmup = 1;
            r.steps = 3;
            float ms[3] = {0.f, 0.f, 0.f};
            rc = hevcdbk_device_replay(ctx, &p, 1, qp, tables, HEVCDBK_KERNEL_AUTO, &r, ms);
            for (int i = 0; i < 3; i++) t[k] = ms[i] < t[k] ? ms[i] : t[k];
        }
    unsigned best = 0, worst = 0;
    if (rc == HEVCDBK_OK)
        for (unsigned k = 1; k < n; k++) {
            if (t[k] < t[best]) best = k;
            if (t[k] > t[worst]) worst = k;
        }
    for (unsigned k = 0; k < n; k++)
        if (rc != HEVCDBK_OK || k != best) (void)hipFree(cand[k]);
    if (rc != HEVCDBK_OK) return rc;
    *dptr = cand[best];
    if (best_ms) *best_ms = t[best];
    if (worst_ms) *worst_ms = t[worst];
    return HEVCDBK_OK;
}

int hevcdbk_device_pci_bus_id(const hevcdbk_context *ctx, char *buf, size_t len)
{
    if (!ctx || !buf || len < 13) return HEVCDBK_ERR_ARG;
    if (hipDeviceGetPCIBusId(buf, (int)len, ctx->device) != hipSuccess) {
        (void)hipGetLastError();
        return HEVCDBK_ERR_HIP;
    }
    return HEVCDBK_OK;
}

/* ---- host-frame operator ---- */

int hevc_deblocking_filter(hevcdbk_context *ctx, hevcdbk_frame *frame, const hevcdbk_bs *bs,
                           const hevcdbk_qp *qp, const hevcdbk_tables *tables, hevcdbk_timing *timing)
{
    if (!ctx || !frame || !qp) return HEVCDBK_ERR_ARG;
    bool chroma = false;
    if (int rc = check_frame(*frame, chroma)) return rc;
    const unsigned W = frame->width, H = frame->height, sb = frame->sample_bytes;
    const int npl = chroma ? 3 : 1;
    const unsigned pw[3] = {W, W / 2, W / 2}, ph[3] = {H, H / 2, H / 2};
    if (int rc = check_bs(bs, W, H, chroma)) return rc;
    if (int rc = bind(ctx)) return rc;

    /* staging (the reference allocates per call, gpu.cu:1103-1169 + 1236-1244; here it persists): the planes sit
     * one after the other, 256-byte aligned, in ONE pinned and ONE device buffer, so a small frame moves in one DMA */
    size_t plane_bytes[3] = {0, 0, 0}, plane_off[3] = {0, 0, 0}, frame_bytes = 0;
    for (int i = 0; i < npl; i++) {
        plane_bytes[i] = (size_t)pw[i] * ph[i] * sb;
        plane_off[i] = frame_bytes;
        frame_bytes = (frame_bytes + plane_bytes[i] + 255) & ~(size_t)255;
    }
    ctx->trace.clear();
    if (int rc = grow_pinned(ctx, ctx->pin[0], frame_bytes)) return rc;
    if (int rc = grow_device(ctx, ctx->dev[0], frame_bytes)) return rc;
    uint8_t *hplane[3], *dplane_b[3];
    for (int i = 0; i < 3; i++) {
        hplane[i] = (uint8_t *)ctx->pin[0].p + plane_off[i];
        dplane_b[i] = (uint8_t *)ctx->dev[0].p + plane_off[i];
    }
    const uint8_t *dmap = nullptr;
    size_t map_rows = 0;
    if (qp->map) {
        if (qp->ctu_log2 < 3 || qp->ctu_log2 > 8 || qp->map_stride < ((W + (1u << qp->ctu_log2) - 1) >> qp->ctu_log2))
            return HEVCDBK_ERR_ARG;
        map_rows = (H + (1u << qp->ctu_log2) - 1) >> qp->ctu_log2;
        if (int rc = grow_device(ctx, ctx->dev_map, map_rows * qp->map_stride)) return rc;
        dmap = (const uint8_t *)ctx->dev_map.p;
    }

    const auto wall0 = std::chrono::steady_clock::now();
    DbkArgs args[3];
    void *dplane[3] = {dplane_b[0], dplane_b[1], dplane_b[2]};

    /* events of the small-frame path: 12 bS / QP map uploaded ; 1..4 h2d ; 4..5 kernel ; 5..9 d2h */
    hipEvent_t *ev = ctx->ev;
    bool bs_uploaded = false;
    if (int rc = stage_bs(ctx, W, H, chroma, bs, ctx->h2d, &bs_uploaded)) return rc;
    if (dmap) HIP_TRY(ctx, hipMemcpyAsync((void *)dmap, qp->map, map_rows * qp->map_stride, hipMemcpyHostToDevice, ctx->h2d));
    const bool operands_uploaded = bs_uploaded || dmap != nullptr; /* else nothing is in flight on h2d that a kernel has to wait for */
    if (operands_uploaded) HIP_TRY(ctx, hipEventRecord(ev[12], ctx->h2d)); /* every kernel of this call waits for it */
    for (int i = 0; i < npl; i++)
        if (int rc = frame_plane_args(ctx, dplane, i, W, H, frame->bit_depth, sb, chroma, qp, dmap, tables, args[i])) return rc;

    /* Small 4:2:0 frames (8-bit or 16-bit containers), where a DMA or a launch costs more than the work it carries: one H2D, one fused
     * launch, one D2H, all on the compute stream (no cross-stream events to pay for). */
    const int sbs[3] = {(int)sb, (int)sb, (int)sb};
    const bool small = chroma && frame_bytes <= ((size_t)2 << 20) && dbk_multi_supports(args, npl, sbs);
    /* planes in page-locked caller memory (hevcdbk_host_malloc_pinned) are DMA'd where they lie -- not for small frames,
     * where one DMA of the packed staging buffer beats three DMAs of the caller's planes */
    bool zc[3] = {false, false, false};
    if (!small)
        for (int i = 0; i < npl; i++) zc[i] = is_pinned_host(frame->plane[i]);
#ifdef HEVCDBK_DIAG /* "dmacopy": small frames through DMA copies around the kernel instead (for A/B runs) */
    const bool host_direct = !g_dbk_diag.dmacopy;
#else
    constexpr bool host_direct = true;
#endif
    if (small && host_direct) {
        /*
         * Small frame: no DMA at all.  The staging buffer is page-locked, fine-grained host memory the GPU can address, so
         * the fused kernel reads and writes it across PCIe itself -- both directions at once, no copy set-up latency
         * (352x288: 50 us per call against 82 us with a DMA each way; 1280x720: 182 against 230 us).  The reference's
         * "copy" figure is therefore 0 for such frames and its "exec" figure contains the PCIe traffic.
         */
        /* planes that are page-locked caller memory themselves (and word aligned) need no staging either */
        bool direct = true;
        void *kplane[3] = {nullptr, nullptr, nullptr}; /* the planes as a kernel addresses them */
        for (int i = 0; i < npl && direct; i++)
            direct = frame->pitch[i] % (4 * sb) == 0 && (uintptr_t)frame->plane[i] % (4 * sb) == 0 && is_pinned_host(frame->plane[i], &kplane[i]) &&
                     kplane[i] != nullptr;
        if (!direct)
            for (int i = 0; i < npl; i++) {
                const size_t rb = (size_t)pw[i] * sb;
                for (unsigned r = 0; r < ph[i]; r++)
                    std::memcpy(hplane[i] + r * rb, (const uint8_t *)frame->plane[i] + r * frame->pitch[i], rb);
            }
        void *hp[3] = {direct ? kplane[0] : (void *)hplane[0], direct ? kplane[1] : (void *)hplane[1],
                       direct ? kplane[2] : (void *)hplane[2]};
        DbkArgs ha[3];
        for (int i = 0; i < npl; i++)
            if (int rc = frame_plane_args(ctx, hp, i, W, H, frame->bit_depth, sb, chroma, qp, dmap, tables, ha[i],
                                          direct ? frame->pitch : nullptr))
                return rc;
        /* The reference's own case (352x288, README.md:19-24) is all fixed costs, so they are counted: no cross-stream wait when the
         * default bS is resident and there is no QP map (nothing was uploaded); the fused launch stamps its own begin and end into
         * ev[4] / ev[5] (no marker packets, two HIP calls fewer); and this thread polls the end event instead of sleeping in a
         * stream synchronisation (round 4: 52 -> about 40 us per call). */
        if (operands_uploaded) HIP_TRY(ctx, hipStreamWaitEvent(ctx->compute, ev[12], 0));
        bool fused = false;
        dbk_set_next_launch_events(ev[4], ev[5]);
        const int frc = launch_frame_fused(ctx, ha, npl, sb, ctx->compute, &fused);
        dbk_set_next_launch_events(nullptr, nullptr);
        if (frc) return frc;
        if (!fused) { /* operands the fused launch does not take after all (alignment of the caller's planes): plane by plane */
            for (int k = 0; k < npl; k++) {
                dbk_set_next_launch_events(k == 0 ? ev[4] : nullptr, k == npl - 1 ? ev[5] : nullptr);
                const int rc = launch(ctx, ha[k], (int)sb, k != 0, HEVCDBK_KERNEL_AUTO, ctx->compute);
                dbk_set_next_launch_events(nullptr, nullptr);
                if (rc) return rc;
            }
        }
        for (;;) {
            const hipError_t q = hipEventQuery(ev[5]);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) HIP_TRY(ctx, q);
            (void)hipGetLastError();
            _mm_pause();
        }
        if (!direct)
            for (int i = 0; i < npl; i++) {
                const size_t rb = (size_t)pw[i] * sb;
                for (unsigned r = 0; r < ph[i]; r++)
                    std::memcpy((uint8_t *)frame->plane[i] + r * frame->pitch[i], hplane[i] + r * rb, rb);
            }
        if (timing) {
            float ms = 0.f;
            HIP_TRY(ctx, hipEventElapsedTime(&ms, ev[4], ev[5]));
            timing->exec_s = ms * 1e-3;
            timing->copy_s = 0.0;
            timing->total_s = timing->exec_s;
            timing->pipelined_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
        }
        return HEVCDBK_OK;
    }
    if (small) {
        /* pack the caller's pitched planes into the pinned staging buffer */
        for (int i = 0; i < npl; i++) {
            const size_t rb = (size_t)pw[i] * sb;
            for (unsigned r = 0; r < ph[i]; r++)
                std::memcpy(hplane[i] + r * rb, (const uint8_t *)frame->plane[i] + r * frame->pitch[i], rb);
        }
        if (operands_uploaded) HIP_TRY(ctx, hipStreamWaitEvent(ctx->compute, ev[12], 0)); /* the bS upload, if there was one */
        HIP_TRY(ctx, hipEventRecord(ev[1], ctx->compute));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->dev[0].p, ctx->pin[0].p, frame_bytes, hipMemcpyHostToDevice, ctx->compute));
        HIP_TRY(ctx, hipEventRecord(ev[4], ctx->compute));
        bool fused = false;
        if (int rc = launch_frame_fused(ctx, args, npl, sb, ctx->compute, &fused)) return rc;
        if (!fused)
            for (int k = 0; k < npl; k++)
                if (int rc = launch(ctx, args[k], (int)sb, k != 0, HEVCDBK_KERNEL_AUTO, ctx->compute)) return rc;
        HIP_TRY(ctx, hipEventRecord(ev[5], ctx->compute));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->pin[0].p, ctx->dev[0].p, frame_bytes, hipMemcpyDeviceToHost, ctx->compute));
        HIP_TRY(ctx, hipEventRecord(ev[9], ctx->compute));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->compute));
    } else {
        /*
         * Large frame: strip pipeline.  Image rows 8b-4 .. 8b+3 belong to block row b and to no other, so a plane splits
         * into strips of whole block rows with no halo; strip s is uploaded, filtered (a launch over its block rows only)
         * and downloaded while strip s+1 is being packed / uploaded and strip s-1 downloaded: the two DMA directions, the
         * kernels and the host-side staging copies all overlap inside ONE frame.  The first strips are short (the link is
         * idle until the first one has been staged), later ones long enough for the DMA engines to reach their rate
         * (tools/ubench/host_stage.hip: one 512 KiB DMA 19 us, 2 MiB 46 us, 8.3 MB 155 us).  Pageable planes are staged
         * by the context's crew of host threads (host_crew.h) while this thread feeds the three streams.
         */
        struct Strip { int plane, b0, b1; unsigned r0, r1; };
        std::vector<Strip> strips;
        bool any_staged = false;
        /* The kernel writes its results straight into page-locked host memory -- the ring, or the caller's own page-locked plane when its
         * rows are tight -- instead of into HBM with a D2H DMA behind it: posted writes across the link run at the DMA engines' rate
         * (tools/ubench/host_stage.hip: 8.3 MB in 0.159 ms against 0.156 ms), and a DMA's set-up, two events and a cross-stream wait
         * per strip disappear from the chain.  Pitched page-locked planes keep the 2-D DMA. */
        const bool direct_out = crew_knob("HEVCDBK_HOST_DIRECT_OUT", true);
        /* experiments of the diagnostic build (profiles/r04/experiments.md): all off in the product */
        const bool x_direct_in = crew_knob("HEVCDBK_HOST_DIRECT_IN", false);
        const int x_h2d_streams = crew_knob("HEVCDBK_HOST_H2D_STREAMS2", false) ? 2 : 1;
        const int x_k_streams = crew_knob("HEVCDBK_HOST_K_STREAMS2", false) ? 2 : 1;
        /* The bS / QP map upload (issued on h2d above, ev[12] behind it) comes first on every stream a strip's kernel may run on.
         * With H2D DMAs the strips' own events on h2d implied it; a strip the crew writes through the BAR has no such event, and
         * its kernel would otherwise be free to start before a caller's bS array or QP map has arrived. */
        if (operands_uploaded) {
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->compute, ev[12], 0));
            if (x_direct_in || x_h2d_streams == 2 || x_k_streams == 2) HIP_TRY(ctx, hipStreamWaitEvent(ctx->d2h, ev[12], 0));
        }
        /* Pageable planes on a large-BAR device: the crew writes the caller's rows straight into HBM through the BAR (posted writes,
         * tools/ubench/bar_write.hip: 37-43 GB/s from one or two cores, as fast as filling the ring) -- no ring on the way in and no
         * H2D DMA with its set-up, event and cross-stream wait per strip.  The buffer is fine-grained device memory, so no cache of the
         * GPU holds a line of it across launches. */
        bool push_in = direct_out && crew_knob("HEVCDBK_HOST_PUSH", true);
        uint8_t *kout[3] = {nullptr, nullptr, nullptr}; /* where the kernel of plane i writes (device-side address), NULL = HBM + DMA */
        for (int i = 0; i < npl; i++) {
            any_staged |= !zc[i];
            if (!direct_out) continue;
            const size_t rb = (size_t)pw[i] * sb;
            void *dp = nullptr;
            if (!zc[i]) kout[i] = hplane[i]; /* hipHostMalloc memory: one address for host and device */
            else if (frame->pitch[i] == rb && (uintptr_t)frame->plane[i] % 4 == 0 && is_pinned_host(frame->plane[i], &dp) && dp)
                kout[i] = (uint8_t *)dp;
        }
        {
            /* strips: 256 KiB, then doubling up to `mid` (1.5 MiB when a copy-out rides along -- the crew, the link and the
             * un-staging then work on three different strips at any time -- 2 MiB when the link works alone), the last `mid` of a
             * frame cut in two so that little is left to do when the last transfer ends */
            const size_t mid_staged = crew_strip_bytes((size_t)3 << 19), mid_locked = crew_strip_bytes((size_t)2 << 20);
            size_t target = crew_first_strip_bytes((size_t)256 << 10);
            for (int i = 0; i < npl; i++) {
                const size_t rb = (size_t)pw[i] * sb, mid = zc[i] ? mid_locked : mid_staged;
                const int nby = (int)(ph[i] / 8 + 1);
                bool cut_tail = i == npl - 1; /* once, at the end of the frame */
                for (int b0 = 0; b0 < nby;) {
                    int per = (int)(((target < mid ? target : mid) / rb) / 8);
                    per = per < 1 ? 1 : per;
                    const int left = nby - b0;
                    int take = per;
                    if (left <= per + per / 2) { /* no sliver at the end of a plane */
                        take = left;
                        if (cut_tail && left > 2 && target >= mid) take = (2 * left + 2) / 3;
                        cut_tail = false;
                    }
                    const int b1 = b0 + take;
                    const unsigned r0 = b0 == 0 ? 0u : (unsigned)(8 * b0 - 4), r1 = b1 == nby ? ph[i] : (unsigned)(8 * b1 - 4);
                    strips.push_back({i, b0, b1, r0, r1});
                    b0 = b1;
                    if (target < mid) target *= 2;
                }
            }
        }
        const size_t ns = strips.size();
        const size_t need = 6 * ns;
        while (ctx->timed_events.size() < need) {
            hipEvent_t e;
            HIP_TRY(ctx, hipEventCreate(&e));
            ctx->timed_events.push_back(e);
        }
        hipEvent_t *te = ctx->timed_events.data(); /* per strip: h2d start/end, kernel start/end, d2h start/end */
        StageCrew *crew = nullptr;
        uint8_t *dpush = nullptr;
        if (any_staged || push_in) { /* page-locked caller planes are pushed too: faster than a chain of DMAs for ONE frame */
            if (int rc = ensure_crew(ctx)) return rc;
            crew = ctx->crew;
            if (push_in) dpush = push_buffer(ctx, frame_bytes); /* NULL: no such memory here -> ring + DMA */
        }
                std::unique_ptr<CopyGroup[]> gin(new CopyGroup[ns]), gout(new CopyGroup[ns]);
        /* every path out of this block waits for the jobs it handed out (they point into gin / gout) and puts the crew to sleep */
        struct CrewScope {
            StageCrew *c; CopyGroup *a, *b; size_t n;
            ~CrewScope() { if (!c) return; for (size_t k = 0; k < n; k++) { c->wait(a[k]); c->wait(b[k]); } c->end(); }
        } scope{crew, gin.get(), gout.get(), ns};
        const int64_t ns0 = std::chrono::duration_cast<std::chrono::nanoseconds>(wall0.time_since_epoch()).count();
        auto since = [ns0](int64_t t) { return t ? (double)(t - ns0) * 1e-9 : 0.0; };
        ctx->trace.assign(ns, hevcdbk_strip_trace{});
        /* a strip as row-range jobs for the crew: in = caller plane -> HBM (BAR) or ring, out = ring -> caller plane */
        auto hand_out = [&](size_t k, bool in) {
            const Strip &st = strips[k];
            const int i = st.plane;
            const size_t rb = (size_t)pw[i] * sb, bytes = (size_t)(st.r1 - st.r0) * rb;
            /* pieces of at least 128 KiB: as many as there are crew threads -- two for writes through the BAR, which two cores saturate */
            const unsigned crew_n = crew->workers() ? crew->workers() : 1, most = in && dpush && crew_n > 2 ? 2 : crew_n;
            unsigned pieces = (unsigned)(bytes / ((size_t)128 << 10));
            pieces = pieces < 1 ? 1 : pieces > most ? most : pieces;
            CopyGroup &g = in ? gin[k] : gout[k];
            g.pending.store((int)pieces, std::memory_order_release);
            const unsigned rows = st.r1 - st.r0;
            for (unsigned p = 0; p < pieces; p++) {
                const unsigned a = st.r0 + (unsigned)((uint64_t)rows * p / pieces), b = st.r0 + (unsigned)((uint64_t)rows * (p + 1) / pieces);
                uint8_t *ring = hplane[i] + (size_t)a * rb, *user = (uint8_t *)frame->plane[i] + (size_t)a * frame->pitch[i];
                if (in && dpush) crew->submit({dpush + plane_off[i] + (size_t)a * rb, user, rb, frame->pitch[i], rb, b - a, &g, true}, StageCrew::LANE_IN);
                else if (in) crew->submit({ring, user, rb, frame->pitch[i], rb, b - a, &g}, StageCrew::LANE_IN);
                else crew->submit({user, ring, frame->pitch[i], rb, rb, b - a, &g}, StageCrew::LANE_OUT);
            }
        };
        auto enqueue = [&](size_t k) -> int {
            const Strip &st = strips[k];
            const int i = st.plane;
            const size_t rb = (size_t)pw[i] * sb, off = (size_t)st.r0 * rb, bytes = (size_t)(st.r1 - st.r0) * rb;
            hevcdbk_strip_trace &tr = ctx->trace[k];
            tr.plane = i; tr.row_begin = st.r0; tr.row_end = st.r1; tr.bytes = bytes;
            if (dpush || !zc[i]) {
                tr.stage_begin_s = since(gin[k].first_ns.load());
                tr.stage_end_s = since(gin[k].done_ns.load());
            }
            tr.enqueue_begin_s = since(crew_now_ns());
            const bool tight = frame->pitch[i] == rb;
            const bool pushed = dpush != nullptr; /* the strip is in HBM already: the crew wrote it there */
            const bool din = pushed || (x_direct_in && kout[i]); /* (experiment: the kernel reads the page-locked strip itself) */
            hipStream_t hs = x_h2d_streams == 2 && (k & 1) ? ctx->d2h : ctx->h2d;
            hipStream_t ks = x_k_streams == 2 && (k & 1) ? ctx->d2h : ctx->compute;
            if (!din) {
                if (timing) HIP_TRY(ctx, hipEventRecord(te[6 * k + 0], hs));
                if (zc[i] && !tight)
                    HIP_TRY(ctx, hipMemcpy2DAsync((uint8_t *)dplane[i] + off, rb, (const uint8_t *)frame->plane[i] + (size_t)st.r0 * frame->pitch[i],
                                                  frame->pitch[i], rb, st.r1 - st.r0, hipMemcpyHostToDevice, hs));
                else
                    HIP_TRY(ctx, hipMemcpyAsync((uint8_t *)dplane[i] + off, (zc[i] ? (const uint8_t *)frame->plane[i] : hplane[i]) + off, bytes,
                                                hipMemcpyHostToDevice, hs));
                HIP_TRY(ctx, hipEventRecord(te[6 * k + 1], hs));
                HIP_TRY(ctx, hipStreamWaitEvent(ks, te[6 * k + 1], 0));
            }
            DbkArgs sa = args[i];
            sa.by_begin = st.b0;
            sa.by_count = st.b1 - st.b0;
            if (kout[i]) sa.dst = kout[i]; /* src: the strip in HBM; dst: page-locked host memory, same pitch */
            if (pushed) sa.src = dpush + plane_off[i];
            else if (din) sa.src = kout[i];
            /* the launch stamps its own begin and end into the strip's events (one kernel per launch() in this library): no marker
             * packets around the kernels, whose chain is what the frame waits for */
            dbk_set_next_launch_events(timing ? te[6 * k + 2] : nullptr, te[6 * k + 3]);
            const int lrc = launch(ctx, sa, (int)sb, i != 0, HEVCDBK_KERNEL_AUTO, ks);
            dbk_set_next_launch_events(nullptr, nullptr);
            if (lrc) return lrc;
            if (!kout[i]) {
                HIP_TRY(ctx, hipStreamWaitEvent(ctx->d2h, te[6 * k + 3], 0));
                if (timing) HIP_TRY(ctx, hipEventRecord(te[6 * k + 4], ctx->d2h));
                if (zc[i] && !tight)
                    HIP_TRY(ctx, hipMemcpy2DAsync((uint8_t *)frame->plane[i] + (size_t)st.r0 * frame->pitch[i], frame->pitch[i],
                                                  (uint8_t *)dplane[i] + off, rb, rb, st.r1 - st.r0, hipMemcpyDeviceToHost, ctx->d2h));
                else
                    HIP_TRY(ctx, hipMemcpyAsync((zc[i] ? (uint8_t *)frame->plane[i] : hplane[i]) + off, (uint8_t *)dplane[i] + off, bytes,
                                                hipMemcpyDeviceToHost, ctx->d2h));
                HIP_TRY(ctx, hipEventRecord(te[6 * k + 5], ctx->d2h));
            }
            tr.enqueue_end_s = since(crew_now_ns());
            return HEVCDBK_OK;
        };
        /*
         * One loop feeds everything, in strip order.  The copies-in of the whole frame go to the crew's first lane at once, so they run
         * without a pause from the first microsecond; a strip whose copy-in is complete is handed to the runtime; a strip whose results
         * have landed (event query, no blocking) goes to the crew's second lane for its copy-out.  This thread copies too until it has
         * something to launch (the crew is still waking up then) and whenever there are no crew threads (hevcdbk_set_host_threads 1).
         */
        if (crew) crew->begin();
        const bool alone = !crew || crew->workers() == 0;
        size_t n_launched = 0, n_out = 0;
        bool helped_once = false;
        if (crew)
            for (size_t k = 0; k < ns; k++)
                if (dpush || !zc[strips[k].plane]) hand_out(k, true);
        while (n_out < ns) {
            bool progress = false;
            if (n_launched < ns && ((!dpush && zc[strips[n_launched].plane]) || gin[n_launched].pending.load(std::memory_order_acquire) == 0)) {
                if (int rc = enqueue(n_launched)) return rc;
                n_launched++;
                progress = true;
            }
            if (n_out < n_launched) {
                const hipError_t q = hipEventQuery(te[6 * n_out + (kout[strips[n_out].plane] ? 3 : 5)]);
                if (q == hipSuccess) {
                    ctx->trace[n_out].d2h_seen_s = since(crew_now_ns());
                    if (!zc[strips[n_out].plane]) hand_out(n_out, false);
                    n_out++;
                    progress = true;
                } else if (q != hipErrorNotReady) {
                    HIP_TRY(ctx, q);
                } else {
                    (void)hipGetLastError(); /* "not ready" is not an error */
                }
            }
            if (progress) continue;
            if (crew && (alone || !helped_once)) { /* the one job: the first piece of the first strip, while the crew wakes up */
                helped_once = true;
                if (crew->help()) continue;
            }
            _mm_pause();
        }
        for (size_t k = 0; k < ns; k++)
            if (!zc[strips[k].plane]) {
                crew->wait(gout[k]);
                ctx->trace[k].unstage_begin_s = since(gout[k].first_ns.load());
                ctx->trace[k].unstage_end_s = since(gout[k].done_ns.load());
            }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->compute));
        if (x_k_streams == 2 || x_h2d_streams == 2) HIP_TRY(ctx, hipStreamSynchronize(ctx->d2h));
        const auto wall1s = std::chrono::steady_clock::now();
        if (timing) {
            double copy = 0.0, exec = 0.0, push_covered = 0.0;
            for (size_t k = 0; k < ns; k++) {
                float ms = 0.f;
                if (dpush) { /* the upload was the crew's stores through the BAR: host clock, the time during
                                                        which at least one strip was being written (strips overlap) */
                    const double b = ctx->trace[k].stage_begin_s, e = ctx->trace[k].stage_end_s;
                    if (e > push_covered) copy += (e - (b > push_covered ? b : push_covered)) * 1e3;
                    if (e > push_covered) push_covered = e;
                } else if (!(x_direct_in && kout[strips[k].plane])) {
                    HIP_TRY(ctx, hipEventElapsedTime(&ms, te[6 * k + 0], te[6 * k + 1])); copy += ms; ctx->trace[k].h2d_ms = ms;
                }
                HIP_TRY(ctx, hipEventElapsedTime(&ms, te[6 * k + 2], te[6 * k + 3])); exec += ms; ctx->trace[k].kernel_ms = ms;
                if (kout[strips[k].plane]) continue; /* the download is the kernel's own stores: part of exec_s */
                HIP_TRY(ctx, hipEventElapsedTime(&ms, te[6 * k + 4], te[6 * k + 5])); copy += ms; ctx->trace[k].d2h_ms = ms;
            }
            timing->exec_s = exec * 1e-3;   /* the reference's figures are sums of its serial phases (gpu.cu:1292-1303) */
            timing->copy_s = copy * 1e-3;
            timing->total_s = timing->exec_s + timing->copy_s;
            timing->pipelined_s = std::chrono::duration<double>(wall1s - wall0).count();
        }
        return HEVCDBK_OK;
    }

    /* small frame: un-stage, and the reference's three figures from the events around its three steps */
    for (int i = 0; i < npl; i++) {
        const size_t rb = (size_t)pw[i] * sb;
        for (unsigned r = 0; r < ph[i]; r++)
            std::memcpy((uint8_t *)frame->plane[i] + r * frame->pitch[i], hplane[i] + r * rb, rb);
    }
    const auto wall1 = std::chrono::steady_clock::now();
    if (timing) {
        float ms = 0.f;
        double copy = 0.0;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ev[1], ev[4])); copy += ms;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ev[5], ev[9])); copy += ms;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ev[4], ev[5]));
        timing->exec_s = ms * 1e-3;
        timing->copy_s = copy * 1e-3;
        timing->total_s = timing->exec_s + timing->copy_s; /* gpu.cu:1302 */
        timing->pipelined_s = std::chrono::duration<double>(wall1 - wall0).count();
    }
    return HEVCDBK_OK;
}

/* ---- streaming host operator: H2D(n+1) || kernels(n) || D2H(n-1) ------------------------------------ */

namespace {
int filter_chunk(hevcdbk_context *ctx, uint8_t *host, size_t k, unsigned W, unsigned H, unsigned qp, const hevcdbk_bs *bs,
                 const hevcdbk_tables *tables, int slot = 0, hipEvent_t done = nullptr);
}


extern "C" int hevc_deblocking_filter_sequence(hevcdbk_context *ctx, hevcdbk_frame *frames, unsigned n_frames,
                                               const hevcdbk_bs *bs, const hevcdbk_qp *qp, const hevcdbk_tables *tables,
                                               hevcdbk_timing *timing)
{
    if (!ctx || !frames || !qp) return HEVCDBK_ERR_ARG;
    if (n_frames == 0) return HEVCDBK_OK;
    if (qp->map) return HEVCDBK_ERR_UNSUPPORTED; /* one QP map per frame is not part of the streaming form */
    const hevcdbk_frame &f0 = frames[0];
    bool chroma = false;
    if (int rc = check_frame(f0, chroma)) return rc;
    const unsigned W = f0.width, H = f0.height, sb = f0.sample_bytes;
    const int npl = chroma ? 3 : 1;
    const unsigned pw[3] = {W, W / 2, W / 2}, ph[3] = {H, H / 2, H / 2};
    for (unsigned i = 1; i < n_frames; i++) {
        const hevcdbk_frame &fr = frames[i];
        bool c2 = false;
        if (fr.width != W || fr.height != H || fr.bit_depth != f0.bit_depth || fr.sample_bytes != sb) return HEVCDBK_ERR_ARG;
        if (int rc = check_frame(fr, c2)) return rc;
        if (c2 != chroma) return HEVCDBK_ERR_ARG;
    }
    if (int rc = check_bs(bs, W, H, chroma)) return rc;
    if (int rc = bind(ctx)) return rc;
    /* Small 8-bit 4:2:0 frames: per-frame DMA and launch costs outweigh the data, so the frames travel in groups -- packed
     * back to back into one pinned chunk, ONE DMA each way and ONE batched fused launch per group (the file operator's path) */
    {
        const size_t fb = (size_t)W * H * 3 / 2;
        if (chroma && sb == 1 && f0.bit_depth == 8 && fb <= ((size_t)2 << 20) && n_frames >= 4) {
            size_t G = ((size_t)64 << 20) / fb;
            G = G > 64 ? 64 : G;
            /* two pinned chunks, two device chunks: the GPU works on group g while the host un-stages group g-1 and stages g+1 */
            uint8_t *chunk[2];
            for (int c = 0; c < 2; c++) {
                if (int rc = grow_pinned(ctx, ctx->seq_pin[0][c], G * fb)) return rc;
                chunk[c] = (uint8_t *)ctx->seq_pin[0][c].p;
                if (!ctx->seq_ev[0][c]) HIP_TRY(ctx, hipEventCreate(&ctx->seq_ev[0][c]));
            }
            const size_t poff[3] = {0, (size_t)W * H, (size_t)W * H + (size_t)(W / 2) * (H / 2)};
            const auto wall0 = std::chrono::steady_clock::now();
            auto stage = [&](unsigned g0, size_t k, uint8_t *c, bool in) {
                for (size_t i = 0; i < k; i++)
                    for (int p = 0; p < 3; p++) {
                        uint8_t *cp = c + i * fb + poff[p];
                        uint8_t *fp = (uint8_t *)frames[g0 + i].plane[p];
                        const size_t pitch = frames[g0 + i].pitch[p];
                        if (pitch == pw[p]) { /* tightly packed plane: one copy */
                            if (in) std::memcpy(cp, fp, (size_t)pw[p] * ph[p]);
                            else std::memcpy(fp, cp, (size_t)pw[p] * ph[p]);
                            continue;
                        }
                        for (unsigned r = 0; r < ph[p]; r++) {
                            if (in) std::memcpy(cp + (size_t)r * pw[p], fp + r * pitch, pw[p]);
                            else std::memcpy(fp + r * pitch, cp + (size_t)r * pw[p], pw[p]);
                        }
                    }
            };
            unsigned prev_g0 = 0;
            size_t prev_k = 0;
            int n = 0;
            for (unsigned g0 = 0; g0 < n_frames; g0 += (unsigned)G, n++) {
                const size_t k = n_frames - g0 < G ? n_frames - g0 : G;
                stage(g0, k, chunk[n & 1], true);
                if (int rc = filter_chunk(ctx, chunk[n & 1], k, W, H, qp->qp, bs, tables, n & 1, ctx->seq_ev[0][n & 1])) return rc;
                if (prev_k) {
                    HIP_TRY(ctx, hipEventSynchronize(ctx->seq_ev[0][(n - 1) & 1]));
                    stage(prev_g0, prev_k, chunk[(n - 1) & 1], false);
                }
                prev_g0 = g0;
                prev_k = k;
            }
            HIP_TRY(ctx, hipEventSynchronize(ctx->seq_ev[0][(n - 1) & 1]));
            stage(prev_g0, prev_k, chunk[(n - 1) & 1], false);
            if (timing) {
                std::memset(timing, 0, sizeof(*timing));
                timing->pipelined_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
            }
            return HEVCDBK_OK;
        }
    }
    constexpr int K = hevcdbk_context::kSeqSlots;
    size_t row_bytes[3], plane_bytes[3];
    for (int k = 0; k < npl; k++) {
        row_bytes[k] = (size_t)pw[k] * sb;
        plane_bytes[k] = row_bytes[k] * ph[k];
    }
    /*
     * Large frames that ALL lie in ordinary pageable memory, on a large-BAR device (round 4): the single-frame operator's way in
     * and out, with a frame where it has a strip.  The staging crew writes frame i straight into slot i % K of fine-grained HBM
     * through the BAR, the kernels of frame i read that slot and store their results into slot i % K of a page-locked ring, the
     * crew copies them out into the caller's planes -- three frames at different stages at any time, no DMA, no staging copy on
     * the way in, and this thread only launches and polls.  A slot of HBM is written again once the kernels that read it have
     * ended, a slot of the ring once its copy-out has ended.
     */
    {
        size_t poff[3] = {0, 0, 0}, fbytes = 0;
        for (int k = 0; k < npl; k++) {
            poff[k] = fbytes;
            fbytes = (fbytes + plane_bytes[k] + 255) & ~(size_t)255;
        }
        bool pageable = fbytes > ((size_t)2 << 20) && n_frames >= 2;
        for (unsigned i = 0; i < n_frames && pageable; i++)
            for (int k = 0; k < npl && pageable; k++) pageable = !is_pinned_host(frames[i].plane[k]);
        uint8_t *dpush = pageable ? push_buffer(ctx, (size_t)K * fbytes) : nullptr;
        if (dpush) {
            if (int rc = grow_pinned(ctx, ctx->pin[1], (size_t)K * fbytes)) return rc;
            uint8_t *const ring = (uint8_t *)ctx->pin[1].p;
            if (int rc = ensure_crew(ctx)) return rc;
            StageCrew *crew = ctx->crew;
            if (int rc = ensure_timed_events(ctx, (size_t)K)) return rc;
            hipEvent_t *done = ctx->timed_events.data(); /* per slot: the end of the kernels that read / wrote it last */
            const auto wall0 = std::chrono::steady_clock::now();
            if (int rc = stage_bs(ctx, W, H, chroma, bs, ctx->h2d)) return rc;
            HIP_TRY(ctx, hipEventRecord(ctx->ev[12], ctx->h2d));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->compute, ctx->ev[12], 0));
            std::unique_ptr<CopyGroup[]> gin(new CopyGroup[n_frames]), gout(new CopyGroup[n_frames]);
            struct Scope { /* every path out waits for the jobs handed out and puts the crew to sleep */
                StageCrew *c; CopyGroup *a, *b; unsigned n;
                ~Scope() { for (unsigned i = 0; i < n; i++) { c->wait(a[i]); c->wait(b[i]); } c->end(); }
            } scope{crew, gin.get(), gout.get(), n_frames};
            const unsigned crew_n = crew->workers() ? crew->workers() : 1;
            auto hand = [&](unsigned i, bool in) {
                const hevcdbk_frame &fr = frames[i];
                const size_t slot = (size_t)(i % K) * fbytes;
                CopyGroup &g = in ? gin[i] : gout[i];
                struct Piece { int k; unsigned a, b; };
                Piece pc[3 * 16];
                int np = 0;
                for (int k = 0; k < npl; k++) {
                    unsigned pieces = (unsigned)(plane_bytes[k] / ((size_t)512 << 10));
                    const unsigned most = in ? (crew_n > 2 ? 4u : 2u) : 2 * crew_n; /* the BAR is saturated by two cores: few, long pieces */
                    pieces = pieces < 1 ? 1 : pieces > most ? most : pieces;
                    pieces = pieces > 16 ? 16 : pieces;
                    for (unsigned q = 0; q < pieces; q++) pc[np++] = {k, (unsigned)((uint64_t)ph[k] * q / pieces), (unsigned)((uint64_t)ph[k] * (q + 1) / pieces)};
                }
                g.pending.store(np, std::memory_order_release);
                for (int q = 0; q < np; q++) {
                    const int k = pc[q].k;
                    uint8_t *user = (uint8_t *)fr.plane[k] + (size_t)pc[q].a * fr.pitch[k];
                    const size_t inner = slot + poff[k] + (size_t)pc[q].a * row_bytes[k];
                    if (in) crew->submit({dpush + inner, user, row_bytes[k], fr.pitch[k], row_bytes[k], pc[q].b - pc[q].a, &g, true}, StageCrew::LANE_IN);
                    else crew->submit({user, ring + inner, fr.pitch[k], row_bytes[k], row_bytes[k], pc[q].b - pc[q].a, &g, false}, StageCrew::LANE_OUT);
                }
            };
            crew->begin();
            const bool alone = crew->workers() == 0;
            unsigned n_in = 0, n_launched = 0, n_out = 0, n_fin = 0;
            while (n_fin < n_frames) {
                bool progress = false;
                /* copies in: slot i % K of HBM is free once the kernels of frame i - K have been seen to end */
                if (n_in < n_frames && n_in < n_out + (unsigned)K) {
                    hand(n_in++, true);
                    progress = true;
                }
                /* launch: the frame is in HBM, and its ring slot has been copied out */
                if (n_launched < n_in && gin[n_launched].pending.load(std::memory_order_acquire) == 0 && n_launched < n_fin + (unsigned)K) {
                    const unsigned i = n_launched;
                    const size_t slot = (size_t)(i % K) * fbytes;
                    DbkArgs args[3];
                    void *dplane[3] = {dpush + slot + poff[0], dpush + slot + poff[1], dpush + slot + poff[2]};
                    for (int k = 0; k < npl; k++) {
                        if (int rc = frame_plane_args(ctx, dplane, k, W, H, f0.bit_depth, sb, chroma, qp, nullptr, tables, args[k])) return rc;
                        args[k].dst = ring + slot + poff[k]; /* the kernels store into page-locked host memory */
                    }
                    const int sbs[3] = {(int)sb, (int)sb, (int)sb};
                    if (chroma && dbk_multi_supports(args, npl, sbs)) {
                        dbk_set_next_launch_events(nullptr, done[i % K]);
                        const hipError_t e = dbk_launch_packed_multi(args, npl, (int)sb, ctx->compute);
                        dbk_set_next_launch_events(nullptr, nullptr);
                        if (!hip_ok(ctx, e, "kernel launch")) return HEVCDBK_ERR_HIP;
                    } else {
                        for (int k = 0; k < npl; k++) {
                            if (k == npl - 1) dbk_set_next_launch_events(nullptr, done[i % K]); /* one stream: the last launch ends last */
                            const int rc = launch(ctx, args[k], (int)sb, k != 0, HEVCDBK_KERNEL_AUTO, ctx->compute);
                            dbk_set_next_launch_events(nullptr, nullptr);
                            if (rc) return rc;
                        }
                    }
                    n_launched++;
                    progress = true;
                }
                /* results landed: copies out */
                if (n_out < n_launched) {
                    const hipError_t q = hipEventQuery(done[n_out % K]);
                    if (q == hipSuccess) {
                        hand(n_out++, false);
                        progress = true;
                    } else if (q != hipErrorNotReady) {
                        HIP_TRY(ctx, q);
                    } else {
                        (void)hipGetLastError();
                    }
                }
                while (n_fin < n_out && gout[n_fin].pending.load(std::memory_order_acquire) == 0) {
                    n_fin++;
                    progress = true;
                }
                if (progress) continue;
                if (alone && crew->help()) continue;
                _mm_pause();
            }
            HIP_TRY(ctx, hipStreamSynchronize(ctx->compute));
            if (timing) {
                std::memset(timing, 0, sizeof(*timing));
                timing->pipelined_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
            }
            return HEVCDBK_OK;
        }
    }
    for (int s = 0; s < K; s++) {
        for (int k = 0; k < npl; k++)
            if (int rc = grow_device(ctx, ctx->seq_dev[s][k], plane_bytes[k])) return rc;
        for (int e = 0; e < 3; e++)
            if (!ctx->seq_ev[s][e]) HIP_TRY(ctx, hipEventCreate(&ctx->seq_ev[s][e]));
    }
    /* bS: shared by all frames of the sequence, uploaded once (the default pattern: once per geometry) */
    const auto wall0 = std::chrono::steady_clock::now();
    if (int rc = stage_bs(ctx, W, H, chroma, bs, ctx->h2d)) return rc;

    std::vector<uint8_t> pinned((size_t)n_frames * 3, 0); /* plane k of frame i is page-locked caller memory */
    /* un-stage frame `i` (its D2H has been issued into slot i % K) */
    auto finish = [&](unsigned i) -> int {
        const int s = (int)(i % K);
        HIP_TRY(ctx, hipEventSynchronize(ctx->seq_ev[s][2]));
        hevcdbk_frame &fr = frames[i];
        for (int k = 0; k < npl; k++) {
            if (pinned[(size_t)i * 3 + k]) continue; /* the DMA wrote the caller's plane directly */
            for (unsigned r = 0; r < ph[k]; r++)
                std::memcpy((uint8_t *)fr.plane[k] + r * fr.pitch[k], (const uint8_t *)ctx->seq_pin[s][k].p + r * row_bytes[k], row_bytes[k]);
        }
        return HEVCDBK_OK;
    };

    for (unsigned i = 0; i < n_frames; i++) {
        const int s = (int)(i % K);
        if (i >= (unsigned)K)
            if (int rc = finish(i - K)) return rc; /* slot s is free again once frame i-K has left it */
        hevcdbk_frame &fr = frames[i];
        /* H2D: straight from pinned caller memory, or via the slot's pinned staging */
        for (int k = 0; k < npl; k++) {
            const void *hsrc = fr.plane[k];
            size_t hpitch = fr.pitch[k];
            pinned[(size_t)i * 3 + k] = is_pinned_host(fr.plane[k]); /* asked once per plane: the query is not free */
            if (!pinned[(size_t)i * 3 + k]) {
                if (int rc = grow_pinned(ctx, ctx->seq_pin[s][k], plane_bytes[k])) return rc;
                for (unsigned r = 0; r < ph[k]; r++)
                    std::memcpy((uint8_t *)ctx->seq_pin[s][k].p + r * row_bytes[k], (const uint8_t *)fr.plane[k] + r * fr.pitch[k], row_bytes[k]);
                hsrc = ctx->seq_pin[s][k].p;
                hpitch = row_bytes[k];
            }
            HIP_TRY(ctx, hipMemcpy2DAsync(ctx->seq_dev[s][k].p, row_bytes[k], hsrc, hpitch, row_bytes[k], ph[k],
                                          hipMemcpyHostToDevice, ctx->h2d));
        }
        HIP_TRY(ctx, hipEventRecord(ctx->seq_ev[s][0], ctx->h2d));
        /* kernels */
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->compute, ctx->seq_ev[s][0], 0));
        DbkArgs args[3];
        void *dplane[3] = {ctx->seq_dev[s][0].p, ctx->seq_dev[s][1].p, ctx->seq_dev[s][2].p};
        for (int k = 0; k < npl; k++)
            if (int rc = frame_plane_args(ctx, dplane, k, W, H, f0.bit_depth, sb, chroma, qp, nullptr, tables, args[k])) return rc;
        bool fused = false;
        if (chroma)
            if (int rc = launch_frame_fused(ctx, args, npl, sb, ctx->compute, &fused)) return rc;
        if (!fused)
            for (int k = 0; k < npl; k++)
                if (int rc = launch(ctx, args[k], (int)sb, k != 0, HEVCDBK_KERNEL_AUTO, ctx->compute)) return rc;
        HIP_TRY(ctx, hipEventRecord(ctx->seq_ev[s][1], ctx->compute));
        /* D2H */
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->d2h, ctx->seq_ev[s][1], 0));
        for (int k = 0; k < npl; k++) {
            void *hdst = fr.plane[k];
            size_t hpitch = fr.pitch[k];
            if (!pinned[(size_t)i * 3 + k]) { hdst = ctx->seq_pin[s][k].p; hpitch = row_bytes[k]; }
            HIP_TRY(ctx, hipMemcpy2DAsync(hdst, hpitch, ctx->seq_dev[s][k].p, row_bytes[k], row_bytes[k], ph[k],
                                          hipMemcpyDeviceToHost, ctx->d2h));
        }
        HIP_TRY(ctx, hipEventRecord(ctx->seq_ev[s][2], ctx->d2h));
        /* slot s is written again by the H2D of frame i+K, which is issued only after finish(i) has seen this
         * frame's D2H (and so its kernels) complete */
    }
    for (unsigned i = n_frames > (unsigned)K ? n_frames - K : 0; i < n_frames; i++)
        if (int rc = finish(i)) return rc;
    const auto wall1 = std::chrono::steady_clock::now();
    if (timing) {
        std::memset(timing, 0, sizeof(*timing));
        timing->pipelined_s = std::chrono::duration<double>(wall1 - wall0).count(); /* whole sequence */
    }
    return HEVCDBK_OK;
}

namespace {

/* k planar 8-bit 4:2:0 frames back to back in pinned host memory: upload, filter all planes as batches, download */
/* slot: which of the two device chunk buffers to use; done == NULL: return when the chunk is back in `host`, else return at
 * once and record `done` behind the download */
int filter_chunk(hevcdbk_context *ctx, uint8_t *host, size_t k, unsigned W, unsigned H, unsigned qp, const hevcdbk_bs *bs,
                 const hevcdbk_tables *tables, int slot, hipEvent_t done)
{
    const size_t ysz = (size_t)W * H, csz = ysz / 4, fb = ysz + 2 * csz;
    const size_t nv = hevcdbk_num_vert_bs(W, H), nh = hevcdbk_num_hor_bs(W, H);
    const size_t ncv = hevcdbk_num_vert_bs(W / 2, H / 2);
    if (int rc = check_bs(bs, W, H, true)) return rc;
    Growable &dbuf = ctx->dev[1 + (slot & 1)];
    if (int rc = grow_device(ctx, dbuf, k * fb)) return rc;
    uint8_t *d = (uint8_t *)dbuf.p;
    hipStream_t s = ctx->compute;
    if (int rc = stage_bs(ctx, W, H, true, bs, s)) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(d, host, k * fb, hipMemcpyHostToDevice, s));
    const uint8_t *dbs = (const uint8_t *)ctx->dev_bs.p;
    DbkArgs args[3];
    for (int i = 0; i < 3; i++) {
        hevcdbk_device_planes p;
        std::memset(&p, 0, sizeof(p));
        p.src = p.dst = d + (i == 0 ? 0 : (i == 1 ? ysz : ysz + csz));
        p.plane_w = i ? W / 2 : W; p.plane_h = i ? H / 2 : H;
        p.pitch = p.plane_w; p.frame_stride = fb; p.n_frames = (unsigned)k;
        p.bit_depth = 8; p.sample_bytes = 1; p.is_chroma = i != 0;
        p.vert_bs = i == 0 ? dbs : dbs + nv + nh;
        p.hor_bs = i == 0 ? dbs + nv : dbs + nv + nh + ncv;
        if (int rc = planes_to_args(&p, qp, tables, args[i])) return rc;
    }
    bool fused = false;
    if (int rc = launch_frame_fused(ctx, args, 3, 1, s, &fused)) return rc;
    if (!fused)
        for (int i = 0; i < 3; i++)
            if (int rc = launch(ctx, args[i], 1, i != 0, HEVCDBK_KERNEL_AUTO, s)) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(host, d, k * fb, hipMemcpyDeviceToHost, s));
    if (done) HIP_TRY(ctx, hipEventRecord(done, s));
    else HIP_TRY(ctx, hipStreamSynchronize(s));
    return HEVCDBK_OK;
}

} /* namespace */

/* ---- multi-frame .yuv file -> file: read || filter || write ---------------------------- */

int hevcdbk_filter_yuv_file(hevcdbk_context *ctx, const char *in_name, const char *out_name, unsigned width,
                            unsigned height, unsigned qp, const hevcdbk_bs *bs, const hevcdbk_tables *tables,
                            unsigned *n_frames_out, hevcdbk_timing *timing)
{
    if (!ctx || !in_name || !out_name || std::strcmp(in_name, out_name) == 0) return HEVCDBK_ERR_ARG;
    FILE *fi = std::fopen(in_name, "rb");
    if (!fi) return HEVCDBK_ERR_IO;
    std::fseek(fi, 0, SEEK_END);
    const long long length = std::ftell(fi);
    std::fseek(fi, 0, SEEK_SET);
    const size_t ysz = (size_t)width * height, csz = ysz / 4, fb = ysz + 2 * csz;
    /* cpu.h:43-48 order: size first (here: a whole number of frames, at least one), then divisibility */
    if (fb == 0 || length <= 0 || (unsigned long long)length % fb != 0) { std::fclose(fi); return HEVCDBK_ERR_FILE_SIZE; }
    if (width % 8 != 0 || height % 8 != 0 || (width / 2) % 8 != 0 || (height / 2) % 8 != 0) {
        std::fclose(fi);
        return HEVCDBK_ERR_DIMENSIONS;
    }
    const unsigned long long n = (unsigned long long)length / fb;
    if (n > 0xffffffffull) { std::fclose(fi); return HEVCDBK_ERR_FILE_SIZE; }
    if (int rc = bind(ctx)) { std::fclose(fi); return rc; }
    FILE *fo = std::fopen(out_name, "wb");
    if (!fo) { std::fclose(fi); return HEVCDBK_ERR_IO; }

    /* three pinned chunk buffers: one being read into, one on the GPU, one being written out */
    size_t chunk = ((size_t)64 << 20) / fb;
    chunk = chunk < 1 ? 1 : (chunk > 64 ? 64 : chunk);
    if (chunk > n) chunk = (size_t)n;
    const size_t nchunks = (size_t)((n + chunk - 1) / chunk);
    uint8_t *buf[3] = {nullptr, nullptr, nullptr};
    int rc = HEVCDBK_OK;
    for (int i = 0; i < 3 && rc == HEVCDBK_OK; i++)
        if (!hip_ok(ctx, hipHostMalloc((void **)&buf[i], chunk * fb, hipHostMallocDefault), "hipHostMalloc")) rc = HEVCDBK_ERR_HIP;
    auto frames_in = [&](size_t c) { return (size_t)((c + 1) * chunk <= n ? chunk : n - c * chunk); };
    const auto wall0 = std::chrono::steady_clock::now();
    bool io_ok = true;
    if (rc == HEVCDBK_OK) io_ok = std::fread(buf[0], fb, frames_in(0), fi) == frames_in(0);
    for (size_t c = 0; c < nchunks && rc == HEVCDBK_OK && io_ok; c++) {
        bool rd_ok = true, wr_ok = true;
        std::thread rd, wr;
        if (c + 1 < nchunks)
            rd = std::thread([&, c] { rd_ok = std::fread(buf[(c + 1) % 3], fb, frames_in(c + 1), fi) == frames_in(c + 1); });
        if (c >= 1)
            wr = std::thread([&, c] { wr_ok = std::fwrite(buf[(c - 1) % 3], fb, frames_in(c - 1), fo) == frames_in(c - 1); });
        /* the k frames of the chunk lie back to back (Y, U, V, Y, U, V, ...): ONE DMA in, the three planes as batches of
         * k frames with frame stride = one file frame (one fused launch when the geometry allows), ONE DMA out */
        const size_t k = frames_in(c);
        rc = filter_chunk(ctx, buf[c % 3], k, width, height, qp, bs, tables);
        if (rd.joinable()) rd.join();
        if (wr.joinable()) wr.join();
        io_ok = rd_ok && wr_ok;
    }
    if (rc == HEVCDBK_OK && io_ok)
        io_ok = std::fwrite(buf[(nchunks - 1) % 3], fb, frames_in(nchunks - 1), fo) == frames_in(nchunks - 1);
    const auto wall1 = std::chrono::steady_clock::now();
    std::fclose(fi);
    if (std::fclose(fo) != 0) io_ok = false;
    for (auto *b : buf) if (b) (void)hipHostFree(b);
    if (rc != HEVCDBK_OK) return rc;
    if (!io_ok) return HEVCDBK_ERR_IO;
    if (n_frames_out) *n_frames_out = (unsigned)n;
    if (timing) {
        std::memset(timing, 0, sizeof(*timing));
        timing->pipelined_s = std::chrono::duration<double>(wall1 - wall0).count();
    }
    return HEVCDBK_OK;
}

/* ---- the same file operator sharded over several GPUs of one node -------------------------------- */
/*
 * Frames are independent, so the path shards with no exchange step: chunk c of the file (up to 64 frames) belongs to
 * worker c mod G; every worker is one host thread with its own context (= its own device, streams and staging), reads
 * its chunks with pread, filters them, and writes them back at the same offsets with pwrite.  No collective, no
 * xGMI traffic; the shared resources are the file system and the PCIe root complexes.
 */
int hevcdbk_filter_yuv_file_multi(const int *devices, unsigned n_devices, const char *in_name, const char *out_name,
                                  unsigned width, unsigned height, unsigned qp, const hevcdbk_bs *bs,
                                  const hevcdbk_tables *tables, unsigned *n_frames_out, hevcdbk_timing *timing)
{
    if (!devices || n_devices == 0 || n_devices > 64 || !in_name || !out_name || std::strcmp(in_name, out_name) == 0)
        return HEVCDBK_ERR_ARG;
    const int fdi = ::open(in_name, O_RDONLY);
    if (fdi < 0) return HEVCDBK_ERR_IO;
    struct stat st;
    if (::fstat(fdi, &st) != 0) { ::close(fdi); return HEVCDBK_ERR_IO; }
    const long long length = (long long)st.st_size;
    const size_t ysz = (size_t)width * height, csz = ysz / 4, fb = ysz + 2 * csz;
    if (fb == 0 || length <= 0 || (unsigned long long)length % fb != 0) { ::close(fdi); return HEVCDBK_ERR_FILE_SIZE; }
    if (width % 8 != 0 || height % 8 != 0 || (width / 2) % 8 != 0 || (height / 2) % 8 != 0) { ::close(fdi); return HEVCDBK_ERR_DIMENSIONS; }
    const unsigned long long n = (unsigned long long)length / fb;
    if (n > 0xffffffffull) { ::close(fdi); return HEVCDBK_ERR_FILE_SIZE; }
    const int fdo = ::open(out_name, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fdo < 0) { ::close(fdi); return HEVCDBK_ERR_IO; }
    if (::ftruncate(fdo, (off_t)length) != 0) { ::close(fdi); ::close(fdo); return HEVCDBK_ERR_IO; }

    size_t chunk = ((size_t)64 << 20) / fb;
    chunk = chunk < 1 ? 1 : (chunk > 64 ? 64 : chunk);
    /* at least one chunk per device when the file is long enough */
    while (chunk > 1 && (n + chunk - 1) / chunk < n_devices) chunk = (chunk + 1) / 2;
    const size_t nchunks = (size_t)((n + chunk - 1) / chunk);
    std::vector<int> rcs(n_devices, HEVCDBK_OK);
    const auto wall0 = std::chrono::steady_clock::now();
    auto worker = [&](unsigned g) {
        hevcdbk_context *ctx = nullptr;
        int rc = hevcdbk_create(devices[g], &ctx);
        uint8_t *buf = nullptr;
        if (rc == HEVCDBK_OK && hipHostMalloc((void **)&buf, chunk * fb, hipHostMallocDefault) != hipSuccess) rc = HEVCDBK_ERR_HIP;
        for (size_t c = g; c < nchunks && rc == HEVCDBK_OK; c += n_devices) {
            const size_t k = (size_t)((c + 1) * chunk <= n ? chunk : n - c * chunk);
            const off_t off = (off_t)(c * chunk * fb);
            size_t done = 0;
            while (done < k * fb) {
                const ssize_t got = ::pread(fdi, buf + done, k * fb - done, off + (off_t)done);
                if (got <= 0) { rc = HEVCDBK_ERR_IO; break; }
                done += (size_t)got;
            }
            if (rc == HEVCDBK_OK) rc = filter_chunk(ctx, buf, k, width, height, qp, bs, tables);
            done = 0;
            while (rc == HEVCDBK_OK && done < k * fb) {
                const ssize_t put = ::pwrite(fdo, buf + done, k * fb - done, off + (off_t)done);
                if (put <= 0) { rc = HEVCDBK_ERR_IO; break; }
                done += (size_t)put;
            }
        }
        if (buf) (void)hipHostFree(buf);
        if (ctx) hevcdbk_destroy(ctx);
        rcs[g] = rc;
    };
    std::vector<std::thread> th;
    for (unsigned g = 0; g < n_devices; g++) th.emplace_back(worker, g);
    for (auto &t : th) t.join();
    const auto wall1 = std::chrono::steady_clock::now();
    ::close(fdi);
    const bool closed = ::close(fdo) == 0;
    for (int rc : rcs)
        if (rc != HEVCDBK_OK) return rc;
    if (!closed) return HEVCDBK_ERR_IO;
    if (n_frames_out) *n_frames_out = (unsigned)n;
    if (timing) {
        std::memset(timing, 0, sizeof(*timing));
        timing->pipelined_s = std::chrono::duration<double>(wall1 - wall0).count();
    }
    return HEVCDBK_OK;
}

/* ---- ExecuteGpu equivalent (gpu.cu:1230-1306) ---- */

int hevcdbk_execute_gpu(const char *in_name, const char *out_name, unsigned width, unsigned height, unsigned qp,
                        unsigned, unsigned, unsigned, unsigned, int device)
{
    if (!in_name || !out_name) return HEVCDBK_ERR_ARG;
    FILE *fp = std::fopen(in_name, "rb");
    if (!fp) return HEVCDBK_ERR_IO;
    std::fseek(fp, 0, SEEK_END);
    const long length = std::ftell(fp);
    std::fseek(fp, 0, SEEK_SET);
    /* same order as gpu.cu:1082-1087: size first, then divisibility */
    if ((unsigned long)length != 3ul * width * height / 2) { std::fclose(fp); return HEVCDBK_ERR_FILE_SIZE; }
    if (width % 8 != 0 || height % 8 != 0) { std::fclose(fp); return HEVCDBK_ERR_DIMENSIONS; }
    std::vector<uint8_t> buf((size_t)length);
    const size_t got = std::fread(buf.data(), 1, buf.size(), fp);
    std::fclose(fp);
    if (got != buf.size()) return HEVCDBK_ERR_IO;

    hevcdbk_context *ctx = nullptr;
    if (int rc = hevcdbk_create(device, &ctx)) return rc;
    const size_t ysz = (size_t)width * height, csz = ysz / 4;
    hevcdbk_frame fr;
    std::memset(&fr, 0, sizeof(fr));
    fr.width = width; fr.height = height; fr.bit_depth = 8; fr.sample_bytes = 1;
    fr.plane[0] = buf.data(); fr.plane[1] = buf.data() + ysz; fr.plane[2] = buf.data() + ysz + csz;
    fr.pitch[0] = width; fr.pitch[1] = width / 2; fr.pitch[2] = width / 2;
    hevcdbk_qp q = {qp, nullptr, 0, 6};
    hevcdbk_timing t;
    /* one untimed call so the figures exclude allocation, like the reference's windows (gpu.cu:1236-1246) */
    std::vector<uint8_t> warm(buf);
    hevcdbk_frame fw = fr;
    fw.plane[0] = warm.data(); fw.plane[1] = warm.data() + ysz; fw.plane[2] = warm.data() + ysz + csz;
    int rc = hevc_deblocking_filter(ctx, &fw, nullptr, &q, nullptr, nullptr);
    if (rc == HEVCDBK_OK) rc = hevc_deblocking_filter(ctx, &fr, nullptr, &q, nullptr, &t);
    if (rc != HEVCDBK_OK) {
        std::fprintf(stderr, "hevc_deblocking_filter failed: %s (%s)\n", hevcdbk_strerror(rc), hevcdbk_last_error(ctx));
        hevcdbk_destroy(ctx);
        return rc;
    }
    /* the reference's three lines, gpu.cu:1292, 1302, 1303 */
    std::printf("Execution Time without copy on GPU: %gs\n", t.exec_s);
    std::printf("Execution Time with copy on GPU: %gs\n", t.total_s);
    std::printf("Copy Operation Time with GPU buffers: %gs\n", t.copy_s);
    std::fflush(stdout);
    hevcdbk_destroy(ctx);

    FILE *fo = std::fopen(out_name, "wb"); /* Save(): Y, U, V order (gpu.cu:1205-1228) */
    if (!fo) return HEVCDBK_ERR_IO;
    const size_t put = std::fwrite(buf.data(), 1, buf.size(), fo);
    std::fclose(fo);
    return put == buf.size() ? HEVCDBK_OK : HEVCDBK_ERR_IO;
}

} /* extern "C" */
