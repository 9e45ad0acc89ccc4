#!/usr/bin/env python3
"""Device-resident rate of whole 4:2:0 frames (Y + U + V, BASELINE config 4), reference-exact mode: one fused launch per step (three with
--diag nofuse) over batches of F frames, per-step HIP-event time.  Diagnostic beside bench.py (whose metric is luma frames/s)."""
import argparse, json, os, sys
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
    ap.add_argument("--qp", type=int, default=32)
    ap.add_argument("--diag", default=None, help="load libhevcdbk_diag.so with these knobs, e.g. nofuse (three launches per step)")
    a = ap.parse_args()
    if a.diag is not None:
        _lib.use_diagnostic_library(a.diag)
    w, h, F, bd = a.width, a.height, a.frames, a.bit_depth
    sb = 1 if bd == 8 else 2
    ctx = deblock.Context(0)
    batches = []
    for i, (pw, ph, ch) in enumerate(((w, h, False), (w // 2, h // 2, True), (w // 2, h // 2, True))):
        b = deblock.DeviceBatch(ctx, pw, ph, F, bit_depth=bd, is_chroma=ch, per_frame_bs=False)
        distinct = min(F, 4)
        src = np.stack([synth.blocky_plane(pw, ph, seed=3 + i, frame=f, bit_depth=bd, dc_range=4 if ch else 6) for f in range(distinct)])
        b.upload_all(np.concatenate([src] * (F // distinct + 1))[:F])
        batches.append(b)
    planes = [b.planes() for b in batches]
    for variant, name in ((_lib.KERNEL_AUTO, "auto"), (_lib.KERNEL_GENERIC, "generic")):
        ctx.run_timed(planes, a.qp, 100, variant=variant)
        ms = ctx.run_timed(planes, a.qp, a.steps, variant=variant)
        nbytes = F * sb * 2 * (w * h + 2 * (w // 2) * (h // 2))
        t = float(np.mean(ms)) * 1e-3
        print(json.dumps({"workload": "%dx%d %d-bit 4:2:0 x %d frames, QP %d, default bS" % (w, h, bd, F, a.qp), "kernels": name, "diag": a.diag,
                          "ms_per_step": t * 1e3, "yuv420_frames_per_s": F / t, "sample_GBps": nbytes / t * 1e-9,
                          "frac_of_8TBps": nbytes / t / 8e12}))


if __name__ == "__main__":
    main()
