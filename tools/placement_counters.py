#!/usr/bin/env python3
"""Why identical launches run 5-10 % apart from one frame pool to the next (BASELINE config 5: 7680x4320 10-bit luma, 32 frames
per launch): several pools are allocated in ONE process, each is timed, and the same process runs under
`rocprofv3 --pmc` with the address-translation counters of the vector memory path (TCP_UTCL1_*), so that every pool has a
duration AND its translation hits / misses from the same dispatches.

    python3 tools/placement_counters.py                 # parent: runs itself under rocprofv3, joins counters and durations
    python3 tools/placement_counters.py --child ...     # what is profiled: allocates the pools, launches, prints the plan

Run ON the GPU box (cd /tmp; TMPDIR=/tmp).  Writes gpurun_out/placement/<tag>.json and prints it."""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

COUNTERS = ["TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum", "TCP_UTCL1_REQUEST_sum", "TCP_UTCL1_PERMISSION_MISS_sum"]
COUNTERS2 = ["TCP_UTCL1_SERIALIZATION_STALL", "TCP_UTCL1_THRASHING_STALL", "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS", "TCP_UTCL1_STALL_INFLIGHT_MAX"]


def child(a):
    from gpu_video_codec_amd import deblock, synth, _lib
    variant = _lib.KERNEL_AUTO
    if a.diag is not None or a.variant == "copy":   # the diagnostic build: the kernel's loads and stores with no arithmetic, knobs
        _lib.use_diagnostic_library(a.diag or None)
        if a.variant == "copy":
            variant = _lib.DIAG_KERNEL_COPY
    w, h, bd, F = a.width, a.height, a.bit_depth, a.frames
    sb = 1 if bd == 8 else 2
    ctx = deblock.Context(0)
    frame = synth.blocky_plane(w, h, seed=3, frame=0, bit_depth=bd)
    raw = np.ascontiguousarray(frame).view(np.uint8).ravel()
    fb = w * h * sb
    pools = []
    junk = []
    for k in range(a.pools):
        if a.interleave_junk:  # small allocations between the pools, as a process that has been running for a while has them
            junk.append(ctx.alloc((3 << 20) + 4096 * k))
        b = deblock.DeviceBatch(ctx, w, h, F, bit_depth=bd, per_frame_bs=False)
        for f in range(F):
            b.src.upload(raw, f * fb)
        pools.append(b)
    plan = {"pools": [{"src": b.src.ptr, "dst": b.dst.ptr, "src_mod_2MiB": b.src.ptr % (2 << 20), "dst_mod_2MiB": b.dst.ptr % (2 << 20),
                       "src_mod_1GiB": b.src.ptr % (1 << 30)} for b in pools], "order": [], "event_ms": []}
    for rnd in range(a.rounds):
        for k, b in enumerate(pools):
            ms, _info = ctx.replay([b.planes()], a.qp, a.reps, warmup=2, settle_min_ms=0, settle_max_ms=a.settle_ms, variant=variant)
            plan["order"].append({"pool": k, "launches": int(_info["settle_launches"]) + 2 + a.reps, "timed": a.reps})
            plan["event_ms"].append({"pool": k, "round": rnd, "mean_ms": float(np.mean(ms)), "min_ms": float(np.min(ms))})
    if a.matrix:
        # which side carries the property: every source pool against every destination pool (and in place), event times
        plan["matrix_ms"] = []
        for i, bs_ in enumerate(pools):
            row = []
            for j, bd_ in enumerate(pools):
                p = bs_.planes()
                p.dst = bd_.dst.ptr
                ms, _info = ctx.replay([p], a.qp, a.reps, warmup=2, settle_min_ms=0, settle_max_ms=a.settle_ms, variant=variant)
                row.append(round(float(np.mean(ms)), 4))
            p = bs_.planes()
            p.dst = bs_.src.ptr
            ms, _info = ctx.replay([p], a.qp, a.reps, warmup=2, settle_min_ms=0, settle_max_ms=a.settle_ms, variant=variant)
            row.append(round(float(np.mean(ms)), 4))   # last column: in place (dst = src)
            plan["matrix_ms"].append(row)
    print("PLAN " + json.dumps(plan), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--width", type=int, default=7680)
    ap.add_argument("--height", type=int, default=4320)
    ap.add_argument("--bit-depth", type=int, default=10)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--qp", type=int, default=32)
    ap.add_argument("--pools", type=int, default=6)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--settle-ms", type=float, default=60.0)
    ap.add_argument("--interleave-junk", action="store_true")
    ap.add_argument("--matrix", action="store_true", help="child only, no profiler: every source pool against every destination pool")
    ap.add_argument("--variant", default="filter", choices=["filter", "copy"], help="copy: the diagnostic build's copy variant of the kernel")
    ap.add_argument("--diag", default=None, help="knobs of the diagnostic build (csrc/hevcdbk_diag.h), e.g. align")
    ap.add_argument("--tag", default="cfg5")
    ap.add_argument("--set", type=int, default=1, choices=[1, 2], help="counter set (a process = one placement: one set per run)")
    a = ap.parse_args()
    if a.child:
        return child(a)
    outdir = os.path.join(ROOT, "gpurun_out", "placement")
    os.makedirs(outdir, exist_ok=True)
    d = os.path.join(outdir, a.tag + "_prof")
    ctrs = COUNTERS if a.set == 1 else COUNTERS2
    args = [sys.executable, os.path.abspath(__file__), "--child"] + [x for x in sys.argv[1:] if x != "--child"]
    cmd = ["rocprofv3", "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "--"] + args
    r = subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd="/tmp")
    plan = None
    for line in r.stdout.split("\n"):
        if line.startswith("PLAN "):
            plan = json.loads(line[5:])
    if r.returncode or plan is None:
        print(json.dumps({"error": r.stderr[-1500:], "stdout": r.stdout[-500:]}))
        return 1
    # dispatches of the filter kernel in launch order, one row per (dispatch, counter)
    disp = {}
    for fn in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(fn)):
            if "dbk_packed" not in row["Kernel_Name"]:
                continue
            e = disp.setdefault(int(row["Dispatch_Id"]), {"ns": int(row["End_Timestamp"]) - int(row["Start_Timestamp"])})
            e[row["Counter_Name"]] = float(row["Counter_Value"])
    ids = sorted(disp)
    per_pool = {}
    pos = 0
    for seg in plan["order"]:
        n = seg["launches"]
        timed = ids[pos + n - seg["timed"]:pos + n]   # the last `timed` launches of the segment
        pos += n
        for i in timed:
            per_pool.setdefault(seg["pool"], []).append(disp[i])
    out = {"workload": "%dx%d %d-bit luma, %d frames per launch" % (a.width, a.height, a.bit_depth, a.frames), "counters": ctrs,
           "dispatches_seen": len(ids), "dispatches_planned": sum(s["launches"] for s in plan["order"]), "pools": []}
    for k in sorted(per_pool):
        rows = per_pool[k]
        ent = dict(plan["pools"][k])
        ent["pool"] = k
        ent["kernel_us_under_pmc_median"] = float(np.median([r_["ns"] for r_ in rows])) / 1e3
        ent["event_ms_mean"] = float(np.mean([e["mean_ms"] for e in plan["event_ms"] if e["pool"] == k]))
        for c in ctrs:
            vals = [r_[c] for r_ in rows if c in r_]
            ent[c] = float(np.median(vals)) if vals else None
        if ent.get("TCP_UTCL1_REQUEST_sum"):
            ent["translation_miss_per_request"] = (ent.get("TCP_UTCL1_TRANSLATION_MISS_sum") or 0.0) / ent["TCP_UTCL1_REQUEST_sum"]
        out["pools"].append(ent)
    ts = [p["event_ms_mean"] for p in out["pools"]]
    out["spread_event_ms"] = {"min": min(ts), "max": max(ts), "max_over_min": max(ts) / min(ts)}
    key = ctrs[0]
    xs = [p[key] for p in out["pools"] if p.get(key) is not None]
    if len(xs) == len(ts) and len(ts) > 2 and np.std(xs) > 0 and np.std(ts) > 0:
        out["correlation_of_time_with_" + key] = float(np.corrcoef(xs, ts)[0, 1])
    path = os.path.join(outdir, a.tag + ".json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
