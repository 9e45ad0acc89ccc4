#!/usr/bin/env python3
"""SQ-level picture of ANY tool's kernels (run ON the GPU box): the rocprofv3 --pmc passes of tools/sq_counters.py (counters only,
no trace domains; the profiled program is python3 itself) over a command of this repository, medians over the dispatches of the
kernels whose name contains --kernel.  Writes gpurun_out/sq/<tag>.json and prints it.
   python3 tools/sq_of.py --kernel sao8_kernel --tag sao_edge -- tools/bench_sao.py --types edge --steps 5"""
import argparse, csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from sq_counters import PASSES  # noqa: E402

EXTRA = [["SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS"],
         ["TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_PENDING_STALL_CYCLES_sum"]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--tag", required=True)
    ap.add_argument("--more", action="store_true", help="also the LDS and TCP passes")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = [c for c in a.cmd if c != "--"]
    if not cmd:
        raise SystemExit("sq_of.py: a command after --")
    cmd[0] = os.path.join(ROOT, cmd[0]) if not os.path.isabs(cmd[0]) else cmd[0]
    outdir = os.path.join(ROOT, "gpurun_out", "sq")
    os.makedirs(outdir, exist_ok=True)
    res = {"kernel": a.kernel, "command": " ".join(a.cmd)}
    for i, ctrs in enumerate(PASSES + (EXTRA if a.more else [])):
        d = os.path.join(outdir, "%s_pass%d" % (a.tag, i))
        r = subprocess.run(["rocprofv3", "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "--", sys.executable] + cmd,
                           env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, cwd="/tmp")
        if r.returncode:
            res["pass%d_error" % i] = r.stderr.decode()[-300:]
            continue
        vals = {}
        for fn in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(fn)):
                if a.kernel in row["Kernel_Name"]:
                    vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for k, v in vals.items():
            v.sort()
            res[k] = v[len(v) // 2]
            res.setdefault("dispatches", len(v))
    g = res.get
    if g("SQ_WAVES") and g("SQ_INSTS_VALU"):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1,024 SIMDs, a wave64 VALU instruction = 4 cycles (as profiles/r04/sq/*.json)
        gpu_cycles = g("GRBM_GUI_ACTIVE") / 8.0 if g("GRBM_GUI_ACTIVE") else None
        res["derived"] = {
            "valu_insts_per_wave": g("SQ_INSTS_VALU") / g("SQ_WAVES"),
            "salu_insts_per_wave": (g("SQ_INSTS_SALU") or 0) / g("SQ_WAVES"),
            "vmem_insts_per_wave": ((g("SQ_INSTS_VMEM_RD") or 0) + (g("SQ_INSTS_VMEM_WR") or 0)) / g("SQ_WAVES"),
            "gpu_cycles_per_launch": gpu_cycles,
            "valu_busy_fraction": g("SQ_INSTS_VALU") * 4.0 / 1024.0 / gpu_cycles if gpu_cycles else None,
            "avg_waves_per_simd": g("SQ_WAVE_CYCLES") / g("SQ_BUSY_CU_CYCLES") if g("SQ_BUSY_CU_CYCLES") and g("SQ_WAVE_CYCLES") else None,
            "wait_any_per_wave_cycle": (g("SQ_WAIT_ANY") or 0) / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None,
            "wait_inst_per_wave_cycle": (g("SQ_WAIT_INST_ANY") or 0) / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None,
        }
    json.dump(res, open(os.path.join(outdir, a.tag + ".json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
