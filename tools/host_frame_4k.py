"""tools/host_frame_4k.py -- the reference-shaped host call (hevc_deblocking_filter, frame in ordinary pageable memory) on one
3840x2160 8-bit luma frame: wall clock per call, the operator's own timing figures, and (library of round 4 on) the per-strip
trace of the last call (stage / enqueue / DMA / kernel / un-stage, host and GPU clocks).  Run it bare, or under
`rocprofv3 --memory-copy-trace --kernel-trace` for the DMA engine's view of the same calls.  Output: one JSON object.

    python3 tools/host_frame_4k.py [--calls 20] [--threads N] [--memory pageable|registered|pinned] [--width W --height H]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpu_video_codec_amd import deblock, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=20)
    ap.add_argument("--threads", type=int, default=None, help="host staging threads of the context (default: the library's)")
    ap.add_argument("--memory", default="pageable", choices=["pageable", "registered", "pinned"])
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--qp", type=int, default=32)
    ap.add_argument("--chroma", action="store_true")
    ap.add_argument("--fresh", action="store_true", help="a newly allocated pageable frame for every call (as bench.py's e2e leg does)")
    ap.add_argument("--affinity", default="none", choices=["none", "near", "far"],
                    help="pin this process to the CPUs next to GPU 0 (sysfs local_cpulist), or to all the others, before the first HIP call")
    ap.add_argument("--h265", action="store_true", help="the spec-exact entry (hevc_deblocking_filter_h265, bS 2 on every interior edge) instead")
    ap.add_argument("--diag", action="store_true", help="load libhevcdbk_diag.so: it reads HEVCDBK_HOST_STREAM_STORES / _AFFINITY / _THREADS (A/B runs)")
    ap.add_argument("--check", action="store_true", help="compare the last call's output with the oracle")
    args = ap.parse_args()
    w, h = args.width, args.height
    if args.affinity != "none":
        from gpu_video_codec_amd import shard
        near = shard.cpus_near_gpu(0)
        if near:
            allowed = os.sched_getaffinity(0)
            pick = (allowed & near) if args.affinity == "near" else (allowed - near)
            if pick:
                os.sched_setaffinity(0, pick)
    if args.diag:
        from gpu_video_codec_amd import _lib
        _lib.use_diagnostic_library(None)
    if args.chroma:
        y0, u0, v0 = synth.blocky_yuv420(w, h, seed=5)
        src = [y0, u0, v0]
    else:
        src = [synth.blocky_plane(w, h, seed=5) if hasattr(synth, "blocky_plane") else synth.blocky_yuv420(w, h, seed=5)[0]]
    out = {"width": w, "height": h, "memory": args.memory, "fresh_buffers": bool(args.fresh), "entry": "hevc_deblocking_filter_h265" if args.h265 else "hevc_deblocking_filter", "affinity": args.affinity, "env": {k: v for k, v in os.environ.items() if k.startswith("HEVCDBK_HOST")}, "planes": len(src), "bytes_each_way": int(sum(p.nbytes for p in src))}
    with deblock.Context(0) as ctx:
        if args.threads is not None and hasattr(ctx, "set_host_threads"):
            ctx.set_host_threads(args.threads)
        if hasattr(ctx, "host_threads"):
            out["host_threads"] = ctx.host_threads()
        if args.memory == "pinned":
            bufs = [ctx.pinned_array(p.shape, p.dtype) for p in src]
        else:
            bufs = [np.empty_like(p) for p in src]
            if args.memory == "registered":
                t0 = time.perf_counter()
                for b in bufs:
                    ctx.host_register(b)
                out["register_s"] = time.perf_counter() - t0
        walls, tms = [], []
        if args.h265:
            vb4 = np.zeros((h // 4, w // 8 + 1), np.uint8)
            vb4[:, 1:w // 8] = 2
            hb4 = np.zeros((h // 8 + 1, w // 4), np.uint8)
            hb4[1:h // 8, :] = 2
        for _ in range(args.calls + 2):
            if args.fresh and args.memory == "pageable":
                bufs = [p.copy() for p in src]
            else:
                for b, p in zip(bufs, src):
                    b[:] = p
            t0 = time.perf_counter()
            if args.h265:
                tm = ctx.filter_frame_h265(*bufs, qp=args.qp, vert_bs4=vb4, hor_bs4=hb4)
            else:
                tm = ctx.filter_frame(*bufs, qp=args.qp)
            walls.append(time.perf_counter() - t0)
            tms.append(tm)
        walls, tms = walls[2:], tms[2:]
        out["wall_s_median"] = float(np.median(walls))
        out["wall_s_min"] = float(np.min(walls))
        out["frames_per_s"] = 1.0 / out["wall_s_median"]
        for k in ("exec_s", "copy_s", "total_s", "pipelined_s"):
            out[k + "_median"] = float(np.median([t[k] for t in tms]))
        if hasattr(ctx, "last_frame_trace"):
            out["last_call_strips"] = ctx.last_frame_trace()
        if args.check and args.h265:
            from oracle import h265
            out["luma_bit_exact_vs_oracle"] = bool(np.array_equal(bufs[0], h265.filter_plane(src[0], args.qp, vb4, hb4)))
        elif args.check:
            from oracle import oracle
            ok = True
            for i, (b, p) in enumerate(zip(bufs, src)):
                want = oracle.filter_plane(p, args.qp, threads=8) if i == 0 else None
                if want is not None:
                    ok &= bool(np.array_equal(b, want))
            out["luma_bit_exact_vs_oracle"] = ok
        if args.memory == "registered":
            for b in bufs:
                ctx.host_unregister(b)
        if args.memory == "pinned":
            for b in bufs:
                ctx.free_pinned(b)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
