#!/usr/bin/env python3
"""Does the placement of dst relative to src change the kernel time?  One big allocation, src at its start, dst at
src_end + delta for several deltas (and dst == src for the copy variant).  Timing only (diagnostic library)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from gpu_video_codec_amd import _lib, deblock, synth
_lib.use_diagnostic_library("")
sys.path.insert(0, ROOT)
import bench
w, h, F, qp = 3840, 2160, 256, 32
ctx = deblock.Context(0)
frames = bench.make_frames(w, h, F, 8, seed=1)
fb = w * h
slack = 64 << 20
big = ctx.alloc(2 * fb * F + slack)
batch = deblock.DeviceBatch(ctx, w, h, F, storage=(big, big))
batch.upload_all(frames)
res = []
def run(name, variant, dst_off, steps=100):
    p = batch.planes()
    p.dst = big.ptr + dst_off
    ctx.run_timed([p], qp, 60, variant=variant)
    ms = ctx.run_timed([p], qp, steps, variant=variant)
    r = {"name": name, "dst_off": dst_off, "ms": float(np.mean(ms)), "min": float(np.min(ms))}
    res.append(r)
    print(json.dumps(r), flush=True)
base = fb * F
for rep in range(2):
    for d in (0, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 8 << 20, (16 << 20) + 65536 * 3, 32 << 20):
        run("copy", _lib.DIAG_KERNEL_COPY, base + d)
    run("copy_inplace", _lib.DIAG_KERNEL_COPY, 0)
for d in (0, 65536, 1 << 20, (2 << 20) + 4096, (16 << 20) + 65536 * 3):
    run("filter", _lib.KERNEL_PACKED, base + d)
run("filter_inplace_evolving", _lib.KERNEL_PACKED, 0)
