#!/usr/bin/env python3
"""Does the row pitch of a frame pool matter?  7680x4320 10-bit luma (row = 15360 bytes), 32 frames per launch, the packed
kernel on pools allocated with different pitches, each in a fresh process-independent allocation; per pitch: mean of 150
timed launches after settling.  Run ON the GPU box: python3 tools/exp/pitch_sweep.py [--width 7680 --height 4320 --bit-depth 10]"""
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
ap.add_argument("--pads", default="0,128,256,512,1024,1152,2176")
a = ap.parse_args()
w, h, bd, F = a.width, a.height, a.bit_depth, a.frames
sb = 1 if bd == 8 else 2
ctx = deblock.Context(0)
base = [synth.blocky_plane(w, h, seed=3, frame=i, bit_depth=bd) for i in range(2)]
frames = np.stack([base[i % 2] for i in range(F)])
alg = F * (2 * w * h * sb + (w // 8 + 1) * (h // 8) + (h // 8 + 1) * (w // 8))
for rep in range(2):
    for pad in [int(x) for x in a.pads.split(",")]:
        pitch = w * sb + pad
        b = deblock.DeviceBatch(ctx, w, h, F, bit_depth=bd, pitch=pitch, per_frame_bs=False)
        b.upload_all(frames)
        p = b.planes()
        ms, info = ctx.replay([p], 32, 150, settle_min_ms=150, settle_max_ms=600)
        t = float(np.mean(ms))
        print(json.dumps({"rep": rep, "pitch": pitch, "pad": pad, "ms": round(t, 4), "frac": round(alg / (t * 1e-3) / 8e12, 4)}), flush=True)
        b.free()
