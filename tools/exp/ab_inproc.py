#!/usr/bin/env python3
"""Same-process, same-memory A/B of several builds of the library (ON the GPU box).

Round 2's A/B runs were one process per build; identical runs then differ by up to 3 % with where the driver places the
frame pool, which hides effects of 1 %.  Here every build is loaded into ONE process (ctypes, RTLD_LOCAL: each library
binds its own symbols and registers its own kernels), all of them work on the SAME source / destination / bS buffers, and
the builds take turns -- A B C A B C ... -- each turn one uninterrupted settle + timed stream (hevcdbk_device_replay; a
library without that entry, i.e. round 2's, gets settle launches + hevcdbk_device_run_timed).  Per build: mean / median of
the per-turn means, and the ratio to the first build turn by turn.

    python3 tools/exp/ab_inproc.py [--rounds 6] [--steps 120] [--frames 256] name=path[:variant[:diag knobs]] ...
variant: auto (default) | copy (diagnostic library only); diag knobs: hevcdbk_diag_set() string of the diagnostic library, e.g. wg=256
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gpu_video_codec_amd import _lib as L0  # struct definitions only; its own library is not loaded here  # noqa: E402
from gpu_video_codec_amd import synth  # noqa: E402


class Lib:
    def __init__(self, name, path, variant, diag=None):
        self.name, self.path = name, path
        fam, _, mp = variant.partition("+")   # e.g. "auto+linear", "copy+group" (the maps beyond rows / linear: diagnostic library)
        self.variant = {"auto": 0, "packed": 2, "copy": 100}[fam or "auto"]
        self.map_bits = {"": 0, "rows": 0x100, "linear": 0x200, "stripe": 0x300, "tiles": 0x400, "pipe": 0x500, "group": 0x600}[mp]
        self.L = C.CDLL(os.path.abspath(path), mode=os.RTLD_LOCAL | os.RTLD_NOW)
        self.L.hevcdbk_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        self.L.hevcdbk_destroy.argtypes = [C.c_void_p]
        self.L.hevcdbk_device_malloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        self.L.hevcdbk_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        self.L.hevcdbk_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        self.L.hevcdbk_default_bs.argtypes = [C.c_uint, C.c_uint, C.c_void_p, C.c_void_p]
        self.L.hevcdbk_device_run_timed.argtypes = [C.c_void_p, C.POINTER(L0.DevicePlanes), C.c_uint, C.c_uint, C.c_void_p, C.c_int,
                                                    C.c_uint, C.POINTER(C.c_float)]
        self.has_replay = hasattr(self.L, "hevcdbk_device_replay")
        if self.has_replay:
            self.L.hevcdbk_device_replay.argtypes = [C.c_void_p, C.POINTER(L0.DevicePlanes), C.c_uint, C.c_uint, C.c_void_p, C.c_int,
                                                     C.POINTER(L0.Replay), C.POINTER(C.c_float)]
        if self.variant == 100 or "diag" in os.path.basename(path):
            self.L.hevcdbk_diag_set.argtypes = [C.c_char_p]
            if self.L.hevcdbk_diag_set(diag.encode() if diag else None) != 0:
                raise SystemExit("%s: hevcdbk_diag_set(%r) refused" % (name, diag))
        h = C.c_void_p()
        rc = self.L.hevcdbk_create(0, C.byref(h))
        if rc:
            raise SystemExit("%s: hevcdbk_create -> %d" % (name, rc))
        self.h = h

    def malloc(self, n):
        p = C.c_void_p()
        assert self.L.hevcdbk_device_malloc(self.h, n, C.byref(p)) == 0
        return p.value

    def turn(self, planes, qp, steps, settle_ms):
        arr = (L0.DevicePlanes * 1)(planes)
        ms = (C.c_float * steps)()
        if self.has_replay:
            r = L0.Replay(settle_min_ms=settle_ms, settle_max_ms=settle_ms, warmup=3, steps=steps)
            rc = self.L.hevcdbk_device_replay(self.h, arr, 1, qp, None, self.variant | self.map_bits, C.byref(r), ms)
        else:
            n = max(int(settle_ms / 0.8), 1)
            tmp = (C.c_float * n)()
            rc = self.L.hevcdbk_device_run_timed(self.h, arr, 1, qp, None, self.variant, n, tmp)
            rc = rc or self.L.hevcdbk_device_run_timed(self.h, arr, 1, qp, None, self.variant, steps, ms)
        if rc:
            raise SystemExit("%s: launch -> %d" % (self.name, rc))
        return np.array(ms, np.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--settle-ms", type=float, default=120.0)
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--qp", type=int, default=32)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    w, h, F = a.width, a.height, a.frames
    libs = []
    for spec in a.libs:
        name, rest = spec.split("=", 1)
        parts = rest.split(":")
        libs.append(Lib(name, parts[0], (parts[1] if len(parts) > 1 and parts[1] else "auto"), parts[2] if len(parts) > 2 else None))
    A = libs[0]
    base = [synth.blocky_plane(w, h, seed=1, frame=i) for i in range(4)]
    frames = np.empty((F, h, w), np.uint8)
    for f in range(F):
        k = f // 4
        frames[f] = np.roll(base[f % 4], (8 * (5 * k % (h // 8)), 8 * (7 * k % (w // 8))), axis=(0, 1))
    nb = F * w * h
    src, dst = A.malloc(nb), A.malloc(nb)
    assert A.L.hevcdbk_memcpy_h2d(A.h, src, frames.ctypes.data, nb) == 0
    nv, nh = (w // 8 + 1) * (h // 8), (h // 8 + 1) * (w // 8)
    vb, hb = np.empty(nv, np.uint8), np.empty(nh, np.uint8)
    A.L.hevcdbk_default_bs(w, h, vb.ctypes.data, hb.ctypes.data)
    dv, dh = A.malloc(nv), A.malloc(nh)
    A.L.hevcdbk_memcpy_h2d(A.h, dv, vb.ctypes.data, nv)
    A.L.hevcdbk_memcpy_h2d(A.h, dh, hb.ctypes.data, nh)
    p = L0.DevicePlanes()
    p.src, p.dst = src, dst
    p.pitch, p.frame_stride, p.n_frames = w, w * h, F
    p.plane_w, p.plane_h, p.bit_depth, p.sample_bytes, p.is_chroma = w, h, 8, 1, 0
    p.vert_bs, p.hor_bs, p.vert_bs_stride, p.hor_bs_stride = dv, dh, 0, 0

    # every build must write the same bytes (the copy variant: the source)
    ref = None
    out = np.empty(w * h, np.uint8)
    for lb in libs:
        lb.turn(p, a.qp, 2, 1.0)
        A.L.hevcdbk_memcpy_d2h(A.h, out.ctypes.data, dst + (F - 1) * w * h, w * h)
        if lb.variant == 100:
            assert np.array_equal(out, frames[F - 1].ravel()), "%s: copy variant changed bytes" % lb.name
        elif ref is None:
            ref = out.copy()
        else:
            assert np.array_equal(out, ref), "%s writes other bytes than %s" % (lb.name, libs[0].name)
    rows = {lb.name: [] for lb in libs}
    med = {lb.name: [] for lb in libs}
    for r in range(a.rounds):
        order = libs if r % 2 == 0 else libs[::-1]   # alternate the order: no build always runs behind the same one
        for lb in order:
            ms = lb.turn(p, a.qp, a.steps, a.settle_ms)
            rows[lb.name].append(float(ms.mean()))
            med[lb.name].append(float(np.median(ms)))
        print("round %d  " % r + "  ".join("%s %.4f" % (lb.name, rows[lb.name][-1]) for lb in libs), flush=True)
    res = {"workload": "%dx%d 8-bit luma x %d, QP %d, one process, shared buffers, %d rounds x %d timed launches after %g ms of settling"
                       % (w, h, F, a.qp, a.rounds, a.steps, a.settle_ms), "builds": {}}
    base_rows = np.array(rows[libs[0].name])
    for lb in libs:
        v = np.array(rows[lb.name])
        res["builds"][lb.name] = {"path": lb.path, "mean_ms": float(v.mean()), "median_of_turn_means_ms": float(np.median(v)),
                                  "median_of_turn_medians_ms": float(np.median(med[lb.name])),
                                  "ratio_to_first_mean": float((v / base_rows).mean()), "ratio_to_first_min": float((v / base_rows).min()),
                                  "ratio_to_first_max": float((v / base_rows).max()), "turn_means_ms": [round(x, 4) for x in v]}
    print(json.dumps(res))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as fh:
            json.dump(res, fh, indent=1)
    for lb in libs:
        lb.L.hevcdbk_destroy(lb.h)


if __name__ == "__main__":
    main()
