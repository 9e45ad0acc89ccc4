#!/usr/bin/env python3
"""SQ-level picture of the deblocking kernel (run ON the GPU box): a few rocprofv3 --pmc passes (counters only, no trace
domains) over a short bench.py run, medians over the kernel's dispatches.  Writes gpurun_out/sq/<tag>.json and prints it.
   python tools/sq_counters.py [--variant packed|copy] [--tag name] [bench.py geometry arguments]"""
import argparse, csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"],
    ["SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM"],
    ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"],
    ["SQ_INST_CYCLES_SALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVES", "SQ_ACTIVE_INST_MISC"],
    ["SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_VMEM_TA_CMD_FIFO_FULL", "SQ_VMEM_WR_TA_DATA_FIFO_FULL", "SQ_IFETCH"],
    ["GRBM_GUI_ACTIVE", "SQ_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_LEVEL_WAVES"],
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="packed")
    ap.add_argument("--tag", default=None)
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--lib", default=None, help="another build of libhevcdbk.so (tools/bench_with_lib.py): same-box comparison of two kernel versions")
    args, extra = ap.parse_known_args()
    tag = args.tag or args.variant
    outdir = os.path.join(ROOT, "gpurun_out", "sq")
    os.makedirs(outdir, exist_ok=True)
    res = {}
    for i, ctrs in enumerate(PASSES):
        d = os.path.join(outdir, "%s_pass%d" % (tag, i))
        prog = [os.path.join(ROOT, "bench.py")] if args.lib is None else [os.path.join(ROOT, "tools", "bench_with_lib.py"), os.path.abspath(args.lib)]
        cmd = ["rocprofv3", "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "--", sys.executable] + prog + ["--steps", "5", "--warmup", "1", "--settle-max-ms", "20", "--copy-floor", "off", "--no-telemetry", "--variant", args.variant,
               "--frames", str(args.frames), "--no-cpu-baseline", "--no-e2e", "--no-extra", "--traffic", "none"] + extra
        r = subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, cwd=ROOT)
        if r.returncode:
            res["pass%d_error" % i] = r.stderr.decode()[-300:]
            continue
        vals = {}
        for fn in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(fn)):
                if "dbk_packed" in row["Kernel_Name"] or "dbk_stripe" in row["Kernel_Name"]:
                    vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for k, v in vals.items():
            v.sort()
            res[k] = v[len(v) // 2]
    g = res.get
    if g("SQ_BUSY_CYCLES") and g("SQ_ACTIVE_INST_VALU"):
        res["derived"] = {
            "valu_active_per_busy_cycle": g("SQ_ACTIVE_INST_VALU") / g("SQ_BUSY_CYCLES"),
            "any_inst_active_per_busy_cycle": (g("SQ_ACTIVE_INST_ANY") or 0) / g("SQ_BUSY_CYCLES"),
            "wait_any_per_wave_cycle": (g("SQ_WAIT_ANY") or 0) / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None,
            "wait_inst_per_wave_cycle": (g("SQ_WAIT_INST_ANY") or 0) / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None,
            "valu_insts_per_wave": g("SQ_INSTS_VALU") / g("SQ_WAVES") if g("SQ_WAVES") else None,
            "salu_insts_per_wave": (g("SQ_INSTS_SALU") or 0) / g("SQ_WAVES") if g("SQ_WAVES") else None,
        }
    path = os.path.join(outdir, tag + ".json")
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
