#!/usr/bin/env python3
"""Per-launch kernel time over a long back-to-back run (shows how the clock settles under the VALU load)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpu_video_codec_amd import _lib, deblock
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
which = sys.argv[2] if len(sys.argv) > 2 else "packed"
if which == "copy":
    _lib.use_diagnostic_library()
variant = {"packed": _lib.KERNEL_PACKED, "copy": _lib.DIAG_KERNEL_COPY}[which]
ctx = deblock.Context(0)
frames = bench.make_frames(3840, 2160, 64, 8)
b = deblock.DeviceBatch(ctx, 3840, 2160, 64)
b.upload_all(frames)
ms = ctx.run_timed([b.planes()], 32, steps, variant=variant)
print("first 10:", np.round(ms[:10], 4))
for i in range(0, steps, max(steps // 10, 1)):
    print("steps %3d-%3d mean %.4f ms" % (i, min(i + steps // 10, steps) - 1, ms[i:i + steps // 10].mean()))
print("overall mean %.4f  min %.4f  max %.4f" % (ms.mean(), ms.min(), ms.max()))
