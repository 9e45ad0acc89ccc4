#!/usr/bin/env python3
"""Does a plain memset of a destination buffer predict how fast the filter writes into it?  (profiles/r04/placement.md)
N destination buffers of one 8K 10-bit 32-frame pool each, one source: for every destination the filter's time (events) and
the time of hipMemset over it (host clock around the synchronous call, best of 7)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gpu_video_codec_amd import deblock, synth, _lib

w, h, bd, F = 7680, 4320, 10, 32
ctx = deblock.Context(0)
b = deblock.DeviceBatch(ctx, w, h, F, bit_depth=bd, per_frame_bs=False)
raw = np.ascontiguousarray(synth.blocky_plane(w, h, seed=3, bit_depth=bd)).view(np.uint8).ravel()
for f in range(F):
    b.src.upload(raw, f * raw.size)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dsts = [b.dst] + [ctx.alloc(F * raw.size) for _ in range(n - 1)]
L = _lib.lib()
import ctypes as C
hip = C.CDLL("libamdhip64.so")
hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
hip.hipEventSynchronize.argtypes = [C.c_void_p]
hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
e0, e1 = C.c_void_p(), C.c_void_p()
hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))
stream = C.c_void_p(L.hevcdbk_compute_stream(ctx.handle))


def timed(fn, reps=7):
    best = 1e9
    for _ in range(reps):
        hip.hipEventRecord(e0, stream)
        fn()
        hip.hipEventRecord(e1, stream)
        hip.hipEventSynchronize(e1)
        ms = C.c_float()
        hip.hipEventElapsedTime(C.byref(ms), e0, e1)
        best = min(best, ms.value)
    return best


rows = []
for rnd in range(2):
    for k, d in enumerate(dsts):
        p = b.planes()
        p.dst = d.ptr
        ms, _i = ctx.replay([p], 32, 12, warmup=2, settle_min_ms=0, settle_max_ms=60)
        nbytes = F * raw.size
        t_set = timed(lambda: hip.hipMemsetAsync(C.c_void_p(d.ptr), 0, nbytes, stream))
        t_cpy = timed(lambda: hip.hipMemcpyAsync(C.c_void_p(d.ptr), C.c_void_p(b.src.ptr), nbytes, 3, stream))
        rows.append({"round": rnd, "dst": k, "filter_ms": round(float(np.mean(ms)), 4), "memset_ms": round(t_set, 4), "d2d_copy_ms": round(t_cpy, 4)})
        print(json.dumps(rows[-1]), flush=True)
f = np.array([r["filter_ms"] for r in rows]); m = np.array([r["memset_ms"] for r in rows])
c = np.array([r["d2d_copy_ms"] for r in rows])
print(json.dumps({"correlation_copy": float(np.corrcoef(f, c)[0, 1]), "copy_spread": float(c.max() / c.min())}))
print(json.dumps({"correlation": float(np.corrcoef(f, m)[0, 1]), "filter_spread": float(f.max() / f.min()), "memset_spread": float(m.max() / m.min())}))
