#!/usr/bin/env python3
"""SAO (H.265 8.7.3) kernel rate: 64 x 3840x2160 8-bit luma in HBM, src -> dst, seeded per-CTB parameters (one third off,
band, edge each), wall clock over back-to-back launches.  Diagnostic."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_video_codec_amd import deblock, synth, _lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--bit-depth", type=int, default=8)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--types", default="mix", help="mix (one third off / band / edge), off, band, edge: the SAO type of every CTB")
    ap.add_argument("--ctb-log2", type=int, default=6, help="CTB size (6 = 64 samples; 5 = 32: the chroma planes of 4:2:0, or a stream with 32-sample CTBs)")
    ap.add_argument("--merge", action="store_true", help="merge the parameters from the left (p 0.5) / upper (p 0.25) CTB, as a stream's SAO merge flags do")
    ap.add_argument("--diag", default=None, help="run on libhevcdbk_diag.so with these knobs (noswz = the plain 3-D strip numbering)")
    a = ap.parse_args()
    if a.diag is not None:
        _lib.use_diagnostic_library(a.diag)
    w, h, n, bd = a.width, a.height, a.frames, a.bit_depth
    sb = 1 if bd == 8 else 2
    ctx = deblock.Context(0)
    b = deblock.DeviceBatch(ctx, w, h, n, bit_depth=bd, per_frame_bs=False)
    src = np.stack([synth.blocky_plane(w, h, seed=7, frame=i, bit_depth=bd) for i in range(4)])
    b.upload_all(np.concatenate([src] * (n // 4 + 1))[:n])
    rng = np.random.RandomState(5)
    cs = 1 << a.ctb_log2
    rows, cols = (h + cs - 1) // cs, (w + cs - 1) // cs
    prm = np.zeros((rows, cols), np.dtype(_lib.SAO_CTB_DTYPE))
    prm["type"] = rng.randint(0, 3, (rows, cols))
    if a.types != "mix":
        prm["type"] = {"off": 0, "band": 1, "edge": 2}[a.types]
    prm["cls"] = np.where(prm["type"] == 1, rng.randint(0, 32, (rows, cols)), rng.randint(0, 4, (rows, cols)))
    prm["offset"] = rng.randint(-7, 8, (rows, cols, 4))
    if a.merge:
        from oracle import h265
        prm = h265.merge_sao_params(prm, seed=18)
    dp = ctx.alloc(prm.nbytes)
    dp.upload(prm.view(np.uint8).ravel())
    p = b.planes()
    for _ in range(100):
        ctx.sao_device(p, dp.ptr, cols, a.ctb_log2)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ctx.sao_device(p, dp.ptr, cols, a.ctb_log2)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    nbytes = 2 * n * w * h * sb
    print(json.dumps({"stage": "sao", "ms_per_launch": dt * 1e3, "frames_per_s": n / dt, "GBps": nbytes / dt * 1e-9,
                      "frac_of_8TBps": nbytes / dt / 8e12, "workload": "%dx%d %d-bit luma x %d, %d-sample CTBs, types: %s" % (w, h, bd, n, cs, a.types), "diag": a.diag}))


if __name__ == "__main__":
    main()
