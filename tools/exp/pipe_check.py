#!/usr/bin/env python3
"""Parity of the diagnostic PIPE map against the oracle on a handful of geometries (run on a GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gpu_video_codec_amd import _lib, deblock, synth
from oracle import oracle
spec = sys.argv[1] if len(sys.argv) > 1 else "rows=4"
_lib.use_diagnostic_library(spec)
from test_gpu_parity import run_batch
rng = np.random.default_rng(4242)
variant = _lib.KERNEL_PACKED | (_lib.DIAG_MAP_GROUP if os.environ.get("DBK_MAP") == "group" else _lib.DIAG_MAP_PIPE)
bad = 0
with deblock.Context(0) as ctx:
    for (w, h, n) in [(3840, 72, 2), (3840, 136, 3), (1920, 264, 2), (1280, 72, 5), (512, 40, 3), (128, 8, 2), (128, 24, 2),
                      (256, 136, 40), (4096, 48, 1), (7680, 40, 1), (1024, 1032, 3), (384, 16, 3), (3840, 2160, 2), (64, 16, 1), (520, 72, 1)]:
        fr = np.stack([synth.blocky_plane(w, h, seed=int(rng.integers(1, 1 << 30))) for _ in range(min(n, 4))])
        fr = np.concatenate([fr] * (n // len(fr) + 1))[:n].copy()
        fr[0, : h // 2, : w // 3] = rng.integers(0, 256, (h // 2, w // 3), dtype=np.uint8)
        if n > 1:
            fr[1] = fr[1][::-1]
        bss = [oracle.lcg_bs(w, h, 5 + f) if f % 2 == 0 else oracle.default_bs(w, h) for f in range(n)]
        for qp, in_place in ((37, False), (32, True), (17, False)):
            got = run_batch(ctx, fr, qp, variant=variant, bs=bss, in_place=in_place)
            for f in sorted({0, 1 % n, n // 2, n - 1}):
                ok = np.array_equal(got[f], oracle.filter_plane(fr[f], qp, vert_bs=bss[f][0], hor_bs=bss[f][1], threads=8))
                if not ok:
                    bad += 1
                    print("MISMATCH", w, h, n, qp, in_place, f, flush=True)
print("PIPE-OK" if bad == 0 else "PIPE-BAD %d" % bad, spec)
sys.exit(1 if bad else 0)
