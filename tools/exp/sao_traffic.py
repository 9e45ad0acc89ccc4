#!/usr/bin/env python3
"""HBM read requests of the SAO pass (run ON the GPU box): rocprofv3 --pmc TCC_EA0_RDREQ by request size over
tools/bench_sao.py, renumbered strips (product numbering) against the plain 3-D grid (diagnostic knob noswz).
   python3 tools/exp/sao_traffic.py [--types mix]"""
import argparse, csv, glob, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--types", default="mix")
    ap.add_argument("--frames", type=int, default=64)
    a = ap.parse_args()
    res = {"plane_bytes_per_launch": a.frames * 3840 * 2160}
    for name, diag in (("renumbered", ""), ("plain_3d_grid", "noswz")):
        d = tempfile.mkdtemp(prefix="sao_rd_")
        cmd = ["rocprofv3", "--pmc", "TCC_EA0_RDREQ", "TCC_EA0_RDREQ_32B", "TCC_EA0_RDREQ_64B", "TCC_EA0_RDREQ_128B", "--output-format", "csv", "-d", d,
               "--", sys.executable, os.path.join(ROOT, "tools", "bench_sao.py"), "--steps", "10", "--frames", str(a.frames), "--types", a.types, "--diag", diag]
        r = subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode:
            res[name] = {"error": r.stderr[-300:]}
            continue
        per = {}
        for fn in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(fn)):
                if "sao" in row["Kernel_Name"]:
                    per.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        m = {k: sorted(v)[len(v) // 2] for k, v in per.items()}
        n32, n64, n128, tot = m.get("TCC_EA0_RDREQ_32B", 0.0), m.get("TCC_EA0_RDREQ_64B", 0.0), m.get("TCC_EA0_RDREQ_128B", 0.0), m.get("TCC_EA0_RDREQ", 0.0)
        rb = 32 * n32 + 64 * n64 + 128 * n128 + 64 * max(tot - n32 - n64 - n128, 0.0)
        res[name] = {"rdreq": m, "read_bytes_from_request_sizes": rb, "read_over_plane": rb / res["plane_bytes_per_launch"]}
        try:
            res[name]["bench"] = json.loads(r.stdout.strip().splitlines()[-1])
        except (ValueError, IndexError):
            pass
    print(json.dumps(res, indent=1))

if __name__ == "__main__":
    main()
