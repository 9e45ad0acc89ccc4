#!/usr/bin/env python3
"""Host-frame operator on a small 4:2:0 frame (the reference's own CPU-runnable case, configs[0]): median of the
reference's timing triple over repeated calls.  --diag nofuse (libhevcdbk_diag.so) gives the three-launch form for comparison."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_video_codec_amd import _lib, deblock, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=352)
    ap.add_argument("--height", type=int, default=288)
    ap.add_argument("--qp", type=int, default=35)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--diag", default=None, help="run on libhevcdbk_diag.so with these knobs (nofuse, dmacopy, ...): A/B runs only")
    ap.add_argument("--file-frames", type=int, default=0, help="also time hevcdbk_filter_yuv_file on a file of N frames")
    ap.add_argument("--sequence-frames", type=int, default=0, help="also time hevc_deblocking_filter_sequence on N frames")
    a = ap.parse_args()
    if a.diag is not None:
        _lib.use_diagnostic_library(a.diag)
    ctx = deblock.Context(0)
    if a.sequence_frames:
        base = [synth.blocky_yuv420(a.width, a.height, seed=11, frame=i) for i in range(8)]
        for kind in ("pageable", "pinned"):
            frames = []
            for i in range(a.sequence_frames):
                src = base[i % 8]
                if kind == "pinned":
                    pl = tuple(ctx.pinned_array(p.shape, p.dtype) for p in src)
                    for d, s_ in zip(pl, src):
                        d[:] = s_
                else:
                    pl = tuple(p.copy() for p in src)
                frames.append(pl)
            best = None
            for _ in range(3):
                t = ctx.filter_sequence(frames, qp=a.qp)
                best = t if best is None else min(best, t)
            print(json.dumps({"sequence_operator": {"memory": kind, "frames": a.sequence_frames, "wall_s": best,
                                                    "frames_per_s": a.sequence_frames / best}}))
            if kind == "pinned":
                for pl in frames:
                    for d in pl:
                        ctx.free_pinned(d)
    if a.file_frames:
        import tempfile
        d = "/dev/shm" if os.path.isdir("/dev/shm") else None
        with tempfile.TemporaryDirectory(dir=d) as td:
            src, dst = os.path.join(td, "in.yuv"), os.path.join(td, "out.yuv")
            with open(src, "wb") as fh:
                frames = [b"".join(p.tobytes() for p in synth.blocky_yuv420(a.width, a.height, seed=9, frame=i)) for i in range(8)]
                for i in range(a.file_frames):
                    fh.write(frames[i % 8])
            best = None
            for _ in range(3):
                n, wall = ctx.filter_yuv_file(src, dst, a.width, a.height, a.qp)
                best = wall if best is None else min(best, wall)
            print(json.dumps({"file_operator": {"frames": n, "wall_s": best, "frames_per_s": n / best,
                                                "MBps_each_way": n * a.width * a.height * 1.5 / best * 1e-6,
                                                "where": td}}))
    y0, u0, v0 = synth.blocky_yuv420(a.width, a.height, seed=5)
    if True:  # the same call on page-locked planes (large frames: DMA'd in place; small frames: the kernel works on them directly)
        pl = [ctx.pinned_array(p.shape, p.dtype) for p in (y0, u0, v0)]
        walls = []
        for _ in range(a.reps):
            for d, s_ in zip(pl, (y0, u0, v0)):
                d[:] = s_
            t0 = time.perf_counter()
            ctx.filter_frame(*pl, qp=a.qp, want_timing=False)
            walls.append(time.perf_counter() - t0)
        print(json.dumps({"single_frame_page_locked_planes": {"wall_s": float(np.median(walls[len(walls) // 4:])),
                                                              "frames_per_s": 1.0 / float(np.median(walls[len(walls) // 4:]))}}))
        for d in pl:
            ctx.free_pinned(d)
    rows = []
    for _ in range(a.reps):
        y, u, v = y0.copy(), u0.copy(), v0.copy()
        t0 = time.perf_counter()
        tm = ctx.filter_frame(y, u, v, qp=a.qp)
        tm["wall_s"] = time.perf_counter() - t0
        rows.append(tm)
    rows = rows[len(rows) // 4:]
    med = {k: float(np.median([r[k] for r in rows])) for k in rows[0]}
    med.update(width=a.width, height=a.height, diag=a.diag or "")
    print(json.dumps(med))


if __name__ == "__main__":
    main()
