#!/usr/bin/env python3
"""Config 5's placement sensitivity (profiles/r04/placement.md) against the per-XCD renumbering of the row-major map: with 32 frames
per launch every XCD owns exactly 4 frames, so the eight XCDs walk through the batch at offsets that are equal modulo 128 KiB all
the time.  Diagnostic knob xpad=N puts N padding workgroups behind every XCD's range.  Several destination pools in ONE process
(one source), every knob value on every pool, event-timed.   python3 tools/exp/xcd_pad.py [--pools 6] [--pads 0,1,3,8,33]"""
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
    ap.add_argument("--pools", type=int, default=6)
    ap.add_argument("--pads", default="0,1,3,8,33")
    ap.add_argument("--reps", type=int, default=12)
    a = ap.parse_args()
    _lib.use_diagnostic_library("")
    w, h, bd, F = a.width, a.height, a.bit_depth, a.frames
    sb = 1 if bd == 8 else 2
    ctx = deblock.Context(0)
    raw = np.ascontiguousarray(synth.blocky_plane(w, h, seed=3, frame=0, bit_depth=bd)).view(np.uint8).ravel()
    pools = []
    for _ in range(a.pools):
        b = deblock.DeviceBatch(ctx, w, h, F, bit_depth=bd, per_frame_bs=False)
        for f in range(F):
            b.src.upload(raw, f * w * h * sb)
        pools.append(b)
    out = {"workload": "%dx%d %d-bit luma, %d frames per launch, source = pool 0" % (w, h, bd, F), "ms": {}}
    for rnd in range(2):
        for pad in [int(x) for x in a.pads.split(",")]:
            _lib.use_diagnostic_library("xpad=%d" % pad if pad else "")
            row = []
            for b in pools:
                p = pools[0].planes()
                p.dst = b.dst.ptr
                ms, _info = ctx.replay([p], 32, a.reps, warmup=2, settle_min_ms=0, settle_max_ms=60.0, variant=_lib.KERNEL_AUTO)
                row.append(round(float(np.mean(ms)), 4))
            out["ms"]["xpad=%d round %d" % (pad, rnd)] = row
            print("xpad=%-3d round %d  %s" % (pad, rnd, row), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
