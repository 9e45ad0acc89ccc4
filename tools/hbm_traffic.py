#!/usr/bin/env python3
"""Collect the HBM traffic of the deblocking kernel with rocprofv3 PMC counters (run ON the GPU box).

Method = /opt/skills/guides/MI355X_MICROARCH.md "HBM" + cdna_hip_programming.md section 7:
  * FETCH_SIZE and WRITE_SIZE do not fit one pass (TCC has 4 slots: 3 + 2) -> two separate --pmc runs,
    each with nothing but the counter (no trace domains alongside --pmc).
  * units: KiB  (bytes = counter * 1024).
  * gfx950: FETCH_SIZE under-reports wide coalesced streaming reads by exactly 2x for 16 B/lane;
    "other access widths are uncalibrated: calibrate on a known byte count in your own access
    pattern".  This kernel reads 8 B/lane at 4-byte alignment, so we calibrate with the library's
    diagnostic COPY variant (same loads and stores, dst = src, exactly W*H bytes read and written per
    frame): read_corr = known_read_bytes / FETCH_SIZE_copy, write_corr likewise, then apply both to
    the filter kernel's counters.

Writes profiles/<tag>_hbm_traffic.json, which bench.py folds into its `roofline.traffic` field when
the workload matches.
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_pmc(counter, variant, frames, outdir, steps=3, extra=()):
    d = os.path.join(outdir, "%s_%s" % (counter, variant))
    cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "1",
           "--variant", variant, "--frames", str(frames), "--no-cpu-baseline", "--no-e2e", "--no-extra", "--traffic", "none",
           "--settle-max-ms", "0", "--copy-floor", "off", "--no-telemetry"] + list(extra)
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(cmd, check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT)
    vals = []
    for fn in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == counter and "dbk_packed" in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    assert vals, "no %s rows for %s" % (counter, variant)
    vals.sort()
    return vals[len(vals) // 2]  # median over the dispatches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r01")
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--bit-depth", type=int, default=8)
    ap.add_argument("--outdir", default=os.path.join(ROOT, "gpurun_out", "traffic"))
    ap.add_argument("--no-profiles-copy", action="store_true", help="write the result to --outdir only (bench.py --traffic live)")
    args = ap.parse_args()
    os.makedirs(args.outdir, exist_ok=True)
    w, h, F = args.width, args.height, args.frames
    sb = 1 if args.bit_depth == 8 else 2
    extra = ["--width", str(w), "--height", str(h), "--bit-depth", str(args.bit_depth)]
    known = w * h * F * sb  # bytes read == bytes written by the copy variant
    raw = {}
    for variant in ("copy", "packed"):
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            raw["%s_%s" % (counter, variant)] = run_pmc(counter, variant, F, args.outdir, extra=extra) * 1024.0
    read_corr = known / raw["FETCH_SIZE_copy"]
    write_corr = known / raw["WRITE_SIZE_copy"]
    bs_bytes = ((w // 8 + 1) * (h // 8) + (h // 8 + 1) * (w // 8)) * F
    out = {
        "tag": args.tag, "workload": {"width": w, "height": h, "frames_per_launch": F, "bit_depth": args.bit_depth},
        "raw_bytes_per_launch": raw,
        "calibration": {"known_copy_bytes_each_way": known, "read_corr": read_corr, "write_corr": write_corr,
                        "note": "correction = known bytes / counter on the diagnostic copy variant (8 B/lane, 4-byte aligned)"},
        "hbm_read_bytes_per_launch": raw["FETCH_SIZE_packed"] * read_corr,
        "hbm_write_bytes_per_launch": raw["WRITE_SIZE_packed"] * write_corr,
        "algorithmic_bytes_per_launch": 2 * known + bs_bytes,
    }
    out["hbm_bytes_per_launch"] = out["hbm_read_bytes_per_launch"] + out["hbm_write_bytes_per_launch"]
    out["traffic_over_algorithmic"] = out["hbm_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
    path = os.path.join(ROOT, "profiles", "%s_hbm_traffic.json" % args.tag)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    targets = [os.path.join(args.outdir, os.path.basename(path))] if args.no_profiles_copy else \
        [path, os.path.join(args.outdir, os.path.basename(path))]  # gpurun only returns gpurun_out/
    for p in targets:
        with open(p, "w") as fh:
            json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
