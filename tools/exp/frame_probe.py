#!/usr/bin/env python3
"""Config 5's placement sensitivity at a finer grain: inside a slow and a fast destination pool, where is the time?  Several pools
in one process; each pool timed whole (32 frames per launch), then in groups of G frames (one launch per group, same source
frames), event-timed.  If the groups of a slow pool are all slow the property is the pool's; if they are bimodal it belongs to
pieces of memory and a pool could be put together from fast pieces.   python3 tools/exp/frame_probe.py [--pools 5] [--group 2]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpu_video_codec_amd import deblock, synth, _lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=7680)
    ap.add_argument("--height", type=int, default=4320)
    ap.add_argument("--bit-depth", type=int, default=10)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--pools", type=int, default=5)
    ap.add_argument("--group", type=int, default=2)
    ap.add_argument("--only-slowest-and-fastest", action="store_true")
    ap.add_argument("--reps", type=int, default=16)
    a = ap.parse_args()
    w, h, bd, F, G = a.width, a.height, a.bit_depth, a.frames, a.group
    sb = 1 if bd == 8 else 2
    ctx = deblock.Context(0)
    raw = np.ascontiguousarray(synth.blocky_plane(w, h, seed=3, frame=0, bit_depth=bd)).view(np.uint8).ravel()
    pools = []
    for _ in range(a.pools):
        b = deblock.DeviceBatch(ctx, w, h, F, bit_depth=bd, per_frame_bs=False)
        for f in range(F):
            b.src.upload(raw, f * w * h * sb)
        pools.append(b)
    fb = w * h * sb
    out = {"workload": "%dx%d %d-bit luma; source = pool 0" % (w, h, bd), "pools": []}
    def whole_of(b):
        p = pools[0].planes()
        p.dst = b.dst.ptr
        ms, _ = ctx.replay([p], 32, a.reps, warmup=2, settle_min_ms=0, settle_max_ms=80.0, variant=_lib.KERNEL_AUTO)
        return round(float(np.mean(ms)), 4)

    for rnd in range(2):  # every pool whole, twice, before anything else: the card is at its working clock from the first round on
        print(json.dumps({"whole_ms_round_%d" % rnd: [whole_of(b) for b in pools]}), flush=True)
    for k, b in enumerate(pools):
        p = pools[0].planes()
        p.dst = b.dst.ptr
        ms, _ = ctx.replay([p], 32, a.reps, warmup=2, settle_min_ms=0, settle_max_ms=80.0, variant=_lib.KERNEL_AUTO)
        whole = float(np.mean(ms))
        groups = []
        for g0 in range(0, F, G):
            q = pools[0].planes()
            q.n_frames = G
            q.src = pools[0].src.ptr + g0 * fb
            q.dst = b.dst.ptr + g0 * fb
            ms, _ = ctx.replay([q], 32, a.reps, warmup=2, settle_min_ms=0, settle_max_ms=0.0, variant=_lib.KERNEL_AUTO)
            groups.append(round(float(np.median(ms)) * 1e3, 1))
        e = {"pool": k, "dst": b.dst.ptr, "whole_ms": round(whole, 4), "group_us": groups, "groups_sum_ms": round(sum(groups) * 1e-3, 4)}
        out["pools"].append(e)
        print(json.dumps(e), flush=True)


if __name__ == "__main__":
    main()
