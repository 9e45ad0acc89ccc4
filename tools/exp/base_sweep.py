#!/usr/bin/env python3
"""Does the placement of a frame pool inside its allocation matter?  7680x4320 10-bit luma, 32 frames per launch, un-padded rows;
ONE source and ONE destination allocation, the pool placed at different byte offsets inside them (source and destination
shifted alike, or the destination alone).  Run ON the GPU box."""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gpu_video_codec_amd import _lib, deblock, synth

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=7680)
ap.add_argument("--height", type=int, default=4320)
ap.add_argument("--bit-depth", type=int, default=10)
ap.add_argument("--frames", type=int, default=32)
a = ap.parse_args()
w, h, bd, F = a.width, a.height, a.bit_depth, a.frames
sb = 1 if bd == 8 else 2
ctx = deblock.Context(0)
b = deblock.DeviceBatch(ctx, w, h, 1, bit_depth=bd, per_frame_bs=False)   # for its bS arrays only
frame = synth.blocky_plane(w, h, seed=3, frame=0, bit_depth=bd)
fb = w * h * sb
slack = 4 << 20
S, D = ctx.alloc(F * fb + slack), ctx.alloc(F * fb + slack)
alg = F * (2 * w * h * sb + (w // 8 + 1) * (h // 8) + (h // 8 + 1) * (w // 8))
raw = np.ascontiguousarray(frame).view(np.uint8).ravel()
for f in range(F):   # content once; a shifted pool reads the same bytes at another phase (timing only)
    S.upload(raw, f * fb)
offs = [0, 128, 256, 512, 1024, 2048, 4096, 8192, 65536, 1 << 20, (1 << 21) + 4096]
for rep in range(2):
    for mode in ("both", "dst_only"):
        for off in offs:
            so = off if mode == "both" else 0
            p = b.planes()
            p.src, p.dst = S.ptr + so, D.ptr + off
            p.n_frames, p.frame_stride = F, fb
            ms, info = ctx.replay([p], 32, 120, settle_min_ms=120, settle_max_ms=400)
            t = float(np.mean(ms))
            print(json.dumps({"rep": rep, "mode": mode, "offset": off, "src_mod_2M": (S.ptr + so) % (1 << 21), "ms": round(t, 4),
                              "frac": round(alg / (t * 1e-3) / 8e12, 4)}), flush=True)
