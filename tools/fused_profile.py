#!/usr/bin/env python3
"""The fused deblocking + SAO kernel under rocprofv3 (run ON the GPU box): kernel durations (--kernel-trace --stats) and HBM
bytes per step (FETCH_SIZE / WRITE_SIZE, each in a pass of its own, corrected with the calibration of the newest
profiles/*_hbm_traffic.json: the row loads of all these kernels are 8 bytes per lane) of tools/bench_deblock_sao.py, for the
one-kernel form and for the two launches it replaces.  Prints one JSON object.
   python3 tools/fused_profile.py [--frames 64] [--mode ref|h265]"""
import argparse, csv, glob, json, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows_of(d, suffix):
    out = []
    for fn in glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True):
        out += list(csv.DictReader(open(fn)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--mode", default="ref")
    a = ap.parse_args()
    cal = {"read_corr": 2.0, "write_corr": 1.0}
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json"))):
        try:
            cal = json.load(open(fn))["calibration"]
        except (OSError, ValueError, KeyError):
            pass
    bench = [sys.executable, os.path.join(ROOT, "tools", "bench_deblock_sao.py"), "--frames", str(a.frames), "--mode", a.mode, "--steps", "20"]
    env = dict(os.environ, TMPDIR="/tmp")
    tmp = tempfile.mkdtemp(prefix="fused_prof_")
    res = {"workload": "tools/bench_deblock_sao.py --frames %d --mode %s" % (a.frames, a.mode),
           "calibration": {k: cal[k] for k in ("read_corr", "write_corr")}}
    d = os.path.join(tmp, "stats")
    subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--"] + bench, env=env, cwd="/tmp",
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    res["kernel_stats"] = [{"kernel": r["Name"], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) * 1e-6}
                           for r in rows_of(d, "kernel_stats.csv") if "dbk" in r["Name"] or "sao" in r["Name"]]
    def short(name):
        for k in ("dbk_sao_fused_h265_kernel", "dbk_sao_fused_kernel", "dbk_packed_h265_kernel", "dbk_packed_kernel", "sao8_kernel"):
            if k in name:
                return k
        return None

    def pmc(counters):
        d = os.path.join(tmp, "_".join(counters))
        subprocess.run(["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--"] + bench, env=env, cwd="/tmp",
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        per = {}
        for r in rows_of(d, "_counter_collection.csv"):
            k = short(r["Kernel_Name"])
            if k:
                per.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        return {k: sorted(v)[len(v) // 2] for k, v in per.items()}

    hb = res.setdefault("hbm_bytes_per_launch", {})
    for (k, c), v in pmc(["FETCH_SIZE"]).items():
        hb.setdefault(k, {})["FETCH_SIZE_KiB_raw"] = v
    for (k, c), v in pmc(["WRITE_SIZE"]).items():
        hb.setdefault(k, {})["write_bytes"] = v * 1024.0 * cal["write_corr"]
    # the L2's memory-side read requests by size: bytes without a calibration factor
    try:
        req = pmc(["TCC_EA0_RDREQ", "TCC_EA0_RDREQ_32B", "TCC_EA0_RDREQ_64B", "TCC_EA0_RDREQ_128B"])
        for k in {k for (k, c) in req}:
            n32, n64, n128, tot = (req.get((k, "TCC_EA0_RDREQ_%s" % x), 0.0) for x in ("32B", "64B", "128B", "")) if False else (
                req.get((k, "TCC_EA0_RDREQ_32B"), 0.0), req.get((k, "TCC_EA0_RDREQ_64B"), 0.0), req.get((k, "TCC_EA0_RDREQ_128B"), 0.0),
                req.get((k, "TCC_EA0_RDREQ"), 0.0))
            hb.setdefault(k, {}).update({"rdreq_total": tot, "rdreq_32B": n32, "rdreq_64B": n64, "rdreq_128B": n128,
                                         "read_bytes_from_request_sizes": 32 * n32 + 64 * n64 + 128 * n128 + 64 * max(tot - n32 - n64 - n128, 0.0)})
    except subprocess.CalledProcessError as e:
        res["rdreq_error"] = str(e)
    n = a.frames * 3840 * 2160
    res["algorithmic_bytes_per_step"] = {"read_once_write_once": 2 * n, "two_launches": 4 * n}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
