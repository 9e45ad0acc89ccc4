#!/usr/bin/env python3
"""How much does WHERE the driver places a frame pool matter, inside one process?  Six source / destination pools allocated
one after the other (all kept alive), the same frames in each; copy variant and filter timed on every pool, twice."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from gpu_video_codec_amd import _lib, deblock
_lib.use_diagnostic_library("")
import bench
w, h, F, qp = 3840, 2160, 256, 32
ctx = deblock.Context(0)
frames = bench.make_frames(w, h, F, 8, seed=1)
pools = []
for i in range(6):
    b = deblock.DeviceBatch(ctx, w, h, F)
    b.upload_all(frames)
    pools.append(b)
    if i % 2 == 1:
        pad = ctx.alloc((3 << 20) + 4096 * i)   # perturb where the next pool lands
for rep in range(2):
    for i, b in enumerate(pools):
        p = b.planes()
        ctx.run_timed([p], qp, 60, variant=_lib.DIAG_KERNEL_COPY)
        c = float(np.mean(ctx.run_timed([p], qp, 100, variant=_lib.DIAG_KERNEL_COPY)))
        ctx.run_timed([p], qp, 100, variant=_lib.KERNEL_PACKED)
        f = float(np.mean(ctx.run_timed([p], qp, 150, variant=_lib.KERNEL_PACKED)))
        print(json.dumps({"rep": rep, "pool": i, "src": hex(b.src.ptr), "dst": hex(b.dst.ptr), "copy_ms": round(c, 4), "filter_ms": round(f, 4)}), flush=True)
