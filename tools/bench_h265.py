#!/usr/bin/env python3
"""Spec-exact mode (H.265 8.7.2) kernel rate on the bench workload (64 x 3840x2160 8-bit luma in HBM, QP 32, bS 2 on every
interior edge), generic vs packed kernel.  Wall clock around back-to-back launches (no per-launch events in this entry);
diagnostic, not bench.py's metric.  Parity of this mode is the business of tests/test_gpu_h265.py."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_video_codec_amd import deblock, synth, _lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--qp", type=int, default=32)
    ap.add_argument("--qp-map", type=int, default=0, metavar="LOG2",
                    help="QP per unit of (1 << LOG2) luma samples, qp-6 .. qp+6 (3..6; 0 = one QP): what a stream with cu_qp_delta has")
    ap.add_argument("--bs", choices=["2", "mixed"], default="2", help="bS 2 on every interior edge, or a seeded mix of 0 / 1 / 2 per 4-sample segment")
    ap.add_argument("--only", choices=["generic", "packed"], default=None)
    a = ap.parse_args()
    w, h, n = a.width, a.height, a.frames
    ctx = deblock.Context(0)
    b = deblock.DeviceBatch(ctx, w, h, n, per_frame_bs=False)
    distinct = min(n, 8)
    src = np.stack([synth.blocky_plane(w, h, seed=7, frame=i) for i in range(distinct)])
    b.upload_all(np.concatenate([src] * (n // distinct + 1))[:n])
    vb = np.zeros((h // 4, w // 8 + 1), np.uint8)
    vb[:, 1:w // 8] = 2
    hb = np.zeros((h // 8 + 1, w // 4), np.uint8)
    hb[1:h // 8, :] = 2
    if a.bs == "mixed":
        rng = np.random.RandomState(3)
        vb[:, 1:w // 8] = rng.randint(0, 3, (h // 4, w // 8 - 1))
        hb[1:h // 8, :] = rng.randint(0, 3, (h // 8 - 1, w // 4))
    dv, dh = ctx.alloc(vb.size), ctx.alloc(hb.size)
    dv.upload(vb)
    dh.upload(hb)
    p = b.planes()
    p.vert_bs, p.hor_bs, p.vert_bs_stride, p.hor_bs_stride = dv.ptr, dh.ptr, 0, 0
    bytes_per_launch = n * (2 * w * h + vb.size + hb.size)
    if a.qp_map:
        qmap = synth.ctu_qp_map(w, h, seed=29, lo=max(a.qp - 6, 0), hi=min(a.qp + 6, 51), ctu_log2=a.qp_map)
        dm = ctx.alloc(qmap.nbytes)
        dm.upload(qmap)
        p.qp_map, p.qp_map_stride, p.ctu_log2, p.qp_map_frame_stride = dm.ptr, qmap.shape[1], a.qp_map, 0
        bytes_per_launch += n * qmap.size
    for name, variant in (("generic", _lib.KERNEL_GENERIC), ("packed", _lib.KERNEL_PACKED)):
        if a.only and a.only != name:
            continue
        for _ in range(100):
            ctx.filter_device_h265(p, a.qp, variant=variant)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            ctx.filter_device_h265(p, a.qp, variant=variant)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print(json.dumps({"mode": "h265", "kernel": name, "ms_per_launch": dt * 1e3, "frames_per_s": n / dt,
                          "GBps": bytes_per_launch / dt * 1e-9, "frac_of_8TBps": bytes_per_launch / dt / 8e12,
                          "workload": "%dx%d 8-bit luma x %d, QP %d%s, bS %s" % (w, h, n, a.qp, " +-6 per %d x %d unit" % (1 << a.qp_map, 1 << a.qp_map) if a.qp_map else "", a.bs)}))


if __name__ == "__main__":
    main()
