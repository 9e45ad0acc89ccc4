#!/usr/bin/env python3
"""BASELINE config 3's operands at benchmark size: reference-exact deblocking of 64 x 3840x2160 8-bit luma with bS drawn from
{0, 1, 2} per edge (the reference's LCG) and / or a QP per 64 x 64 CTU (+-6 around --qp), packed kernel, wall clock per launch.
Diagnostic; parity of these operands is the business of tests/test_gpu_parity.py (the QP map is unpinned)."""
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
    ap.add_argument("--qp-map", type=int, default=6, metavar="LOG2", help="QP per unit of (1 << LOG2) luma samples (0 = one QP)")
    ap.add_argument("--bs", choices=["default", "lcg"], default="lcg")
    a = ap.parse_args()
    from oracle import oracle
    w, h, n = a.width, a.height, a.frames
    ctx = deblock.Context(0)
    b = deblock.DeviceBatch(ctx, w, h, n, per_frame_bs=False)
    src = np.stack([synth.blocky_plane(w, h, seed=7, frame=i) for i in range(8)])
    b.upload_all(np.concatenate([src] * (n // 8 + 1))[:n])
    if a.bs == "lcg":
        b.set_bs(0, *oracle.lcg_bs(w, h, 9))
    nbytes = n * (2 * w * h + (w // 8 + 1) * (h // 8) + (h // 8 + 1) * (w // 8))
    if a.qp_map:
        qmap = synth.ctu_qp_map(w, h, seed=29, lo=max(a.qp - 6, 0), hi=min(a.qp + 6, 51), ctu_log2=a.qp_map)
        b.set_qp_map(qmap, a.qp_map)
        nbytes += n * qmap.size
    p = b.planes()
    qp = 0 if a.qp_map else a.qp
    for _ in range(150):
        ctx.filter_device(p, qp, variant=_lib.KERNEL_PACKED)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ctx.filter_device(p, qp, variant=_lib.KERNEL_PACKED)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    want = oracle.filter_plane(src[(n - 1) % 8], qp, vert_bs=oracle.lcg_bs(w, h, 9)[0] if a.bs == "lcg" else None,
                               hor_bs=oracle.lcg_bs(w, h, 9)[1] if a.bs == "lcg" else None, qp_map=qmap if a.qp_map else None,
                               ctu_log2=a.qp_map or 6, threads=8)
    ok = bool(np.array_equal(b.download_frame(n - 1), want))
    print(json.dumps({"mode": "reference-exact", "ms_per_launch": dt * 1e3, "frames_per_s": n / dt, "frac_of_8TBps": nbytes / dt / 8e12,
                      "bit_exact_vs_oracle": ok,
                      "workload": "%dx%d 8-bit luma x %d, %s, bS %s" % (w, h, n, "QP %d +-6 per %d x %d unit" % (a.qp, 1 << a.qp_map, 1 << a.qp_map)
                                                                        if a.qp_map else "QP %d" % a.qp, a.bs)}))


if __name__ == "__main__":
    main()
