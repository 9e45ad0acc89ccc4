#!/usr/bin/env python3
"""Deblocking + SAO of F x 3840x2160 8-bit luma planes in HBM, src -> dst, in one call: the fused kernel
(hevc_deblock_sao_device, one workgroup = one 192x128 tile through LDS) against the two launches it replaces
(HEVCDBK_FUSED_OFF: deblocking into the context's scratch plane, then the SAO pass).  Seeded per-CTB SAO parameters (one
third off / band / edge, or --types), wall clock over back-to-back calls after a settling period.  Diagnostic."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_video_codec_amd import deblock, synth, _lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--qp", type=int, default=32)
    ap.add_argument("--mode", choices=["ref", "h265"], default="ref")
    ap.add_argument("--types", default="mix")
    ap.add_argument("--diag", action="append", default=None,
                    help="diagnostic library with these knobs (csrc/hevcdbk_diag.h), e.g. noswz; may be given several times: one result per knob set")
    a = ap.parse_args()
    if a.diag is not None:
        _lib.use_diagnostic_library(a.diag[0] or None)
    w, h, n = a.width, a.height, a.frames
    ctx = deblock.Context(0)
    b = deblock.DeviceBatch(ctx, w, h, n, per_frame_bs=False)
    src = np.stack([synth.blocky_plane(w, h, seed=7, frame=i) for i in range(4)])
    b.upload_all(np.concatenate([src] * (n // 4 + 1))[:n])
    rng = np.random.RandomState(5)
    rows, cols = (h + 63) // 64, (w + 63) // 64
    prm = np.zeros((rows, cols), np.dtype(_lib.SAO_CTB_DTYPE))
    prm["type"] = rng.randint(0, 3, (rows, cols))
    if a.types != "mix":
        prm["type"] = {"off": 0, "band": 1, "edge": 2}[a.types]
    prm["cls"] = np.where(prm["type"] == 1, rng.randint(0, 32, (rows, cols)), rng.randint(0, 4, (rows, cols)))
    prm["offset"] = rng.randint(-7, 8, (rows, cols, 4))
    dp = ctx.alloc(prm.nbytes)
    dp.upload(prm.view(np.uint8).ravel())
    p = b.planes()
    if a.mode == "h265":
        L = _lib.lib()
        nv, nh = L.hevcdbk_h265_num_vert_bs(w, h), L.hevcdbk_h265_num_hor_bs(w, h)
        dv, dh = ctx.alloc(nv), ctx.alloc(nh)
        dv.upload(np.full(nv, 2, np.uint8))
        dh.upload(np.full(nh, 2, np.uint8))
        p.vert_bs, p.hor_bs, p.vert_bs_stride, p.hor_bs_stride = dv.ptr, dh.ptr, 0, 0

    def call(fused):
        if a.mode == "ref":
            ctx.deblock_sao_device(p, a.qp, dp.ptr, cols, 6, fused=fused)
        else:
            ctx.deblock_sao_h265_device(p, a.qp, dp.ptr, cols, 6, fused=fused)

    nbytes = 2 * n * w * h
    out = {}
    runs = [("fused", _lib.FUSED_ON, None), ("two_launches", _lib.FUSED_OFF, None), ("fused_again", _lib.FUSED_ON, None)]
    if a.diag is not None:
        runs = [("fused[%s]" % k, _lib.FUSED_ON, k) for k in a.diag] * 2
    for name, fused, knobs in runs:
        if knobs is not None:
            _lib.use_diagnostic_library(knobs or None)
            if name in out:
                name += " again"
        for _ in range(max(a.steps, 100)):
            call(fused)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            call(fused)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        out[name] = {"ms_per_step": dt * 1e3, "frames_per_s": n / dt, "frac_of_8TBps_read_once_write_once": nbytes / dt / 8e12}
    print(json.dumps({"stage": "deblock+sao", "mode": a.mode, "workload": "%dx%d 8-bit luma x %d, QP %d, CTB types: %s" % (w, h, n, a.qp, a.types),
                      **out}))


if __name__ == "__main__":
    main()
