/*
 * deblock_ctx.h -- PRIVATE to the library: the context behind `hevcdbk_context *` and the host-side helpers the C-ABI
 * translation units share (deblock_host.cpp: context, plumbing, reference-exact operators, file operators;
 * deblock_host_h265.cpp: spec-exact mode and SAO).  Nothing here is exported (the library is built with hidden visibility).
 */
#pragma once
#include <hip/hip_runtime_api.h>

#include <string>
#include <vector>

#include "../../include/hevc_deblock.h"
#include "deblock_kernels.h"

namespace dbkh { class StageCrew; }

struct Growable {
    void *p = nullptr;
    size_t cap = 0;
};

struct hevcdbk_context {
    int device = 0;
    int n_cus = 0; /* compute units (hipDeviceAttributeMultiprocessorCount): sizes the persistent stripe launches */
    hipStream_t compute = nullptr, h2d = nullptr, d2h = nullptr;
    hipEvent_t ev[16] = {};
    std::string last_error;
    /* staging for the host-frame operator: pinned host + device, grown on demand */
    Growable pin[3], dev[3];
    Growable pin_bs, dev_bs, dev_map, dev_units;
    Growable dev_tmp; /* the deblocked planes between the two launches of hevc_deblock_sao_*_device where the fused kernel does not apply */
    hipEvent_t tmp_ev = nullptr; /* end of the last launch that read dev_tmp: the next user of the scratch plane waits for it */
    bool tmp_used = false;
    std::vector<hipEvent_t> timed_events;
    /* streaming operator: ring of kSeqSlots frames in flight */
    static constexpr int kSeqSlots = 3;
    Growable seq_pin[kSeqSlots][3], seq_dev[kSeqSlots][3];
    hipEvent_t seq_ev[kSeqSlots][3] = {}; /* [slot][0 = h2d done, 1 = kernels done, 2 = d2h done] */
    /* dev_bs holds the DEFAULT bS (cpu.h:92-99) of this geometry when bs_default_at == dev_bs.p: no re-upload */
    const void *bs_default_at = nullptr;
    unsigned bs_default_w = 0, bs_default_h = 0;
    bool bs_default_chroma = false;
    /* large pageable frames of the host-frame operator: the staging crew (host_crew.h), created at the first such call */
    dbkh::StageCrew *crew = nullptr;
    Growable dev_push; /* fine-grained HBM the crew writes the caller's rows into through the PCIe BAR (large-BAR devices) */
    int large_bar = -1; /* hipDeviceAttributeIsLargeBar, asked once */
    unsigned host_threads = 0; /* threads copying during a call, the caller included; 0 = default (4) */
    std::vector<hevcdbk_strip_trace> trace; /* the strips of the last large-frame call (hevcdbk_last_frame_trace) */
    std::vector<void *> registered; /* hevcdbk_host_register'ed ranges still registered: released with the context */
};

namespace dbkh {

bool hip_ok(hevcdbk_context *ctx, hipError_t e, const char *what);
#define HIP_TRY(ctx, call)                                  \
    do {                                                    \
        if (!hip_ok((ctx), (call), #call)) return HEVCDBK_ERR_HIP; \
    } while (0)

int bind(hevcdbk_context *ctx);                                             /* hipSetDevice(ctx->device) */
int grow_pinned(hevcdbk_context *ctx, Growable &g, size_t bytes);
int grow_device(hevcdbk_context *ctx, Growable &g, size_t bytes);
bool bad_depth(unsigned bit_depth, unsigned sample_bytes);
bool is_pinned_host(const void *p, void **dev_ptr = nullptr);
int check_frame(const hevcdbk_frame &f, bool &chroma);
int check_bs(const hevcdbk_bs *bs, unsigned W, unsigned H, bool chroma);
int planes_to_args(const hevcdbk_device_planes *p, unsigned qp, const hevcdbk_tables *tables, DbkArgs &a);
int launch(hevcdbk_context *ctx, const DbkArgs &a0, int sample_bytes, bool chroma, int variant, hipStream_t s);
/* host side of the frame operators (deblock_host.cpp): the staging crew, and HBM the crew writes through the PCIe BAR */
int ensure_crew(hevcdbk_context *ctx);
/* fine-grained device memory of at least `bytes` that the host's cores can write (large-BAR devices), or NULL: no such memory here */
uint8_t *push_buffer(hevcdbk_context *ctx, size_t bytes);
/* every plane of `frame` (tight rows of pw[i] * sb bytes at base + plane_off[i]) copied by the crew and the calling thread, which
 * returns when all of it is done.  to_frame: base -> the caller's planes, else the caller's planes -> base; base_is_device: base is
 * HBM seen through the BAR (streaming stores, flushed by a read) */
int crew_copy_frame(hevcdbk_context *ctx, const hevcdbk_frame *frame, int npl, const unsigned *pw, const unsigned *ph, unsigned sb,
                    uint8_t *base, const size_t *plane_off, bool to_frame, bool base_is_device);

} /* namespace dbkh */
