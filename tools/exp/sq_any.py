#!/usr/bin/env python3
"""SQ counters of ANY kernel of the library (run ON the GPU box): rocprofv3 --pmc passes (counters only, no trace domains)
over a python tool of this repository, medians over the dispatches whose kernel name contains --kernel.
   python3 tools/exp/sq_any.py --kernel dbk_sao_fused_kernel --tag fused8 tools/bench_deblock_sao.py --steps 5
Writes gpurun_out/sq/<tag>.json and prints it."""
import argparse, csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PASSES = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"],
    ["SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM"],
    ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"],
    ["SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAVES"],
    ["GRBM_GUI_ACTIVE", "SQ_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INST_CYCLES_SALU"],
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", required=True, help="substring of the kernel name")
    ap.add_argument("--tag", required=True)
    ap.add_argument("prog", nargs=argparse.REMAINDER, help="python script (relative to the repository root) and its arguments")
    a = ap.parse_args()
    outdir = os.path.join(ROOT, "gpurun_out", "sq")
    os.makedirs(outdir, exist_ok=True)
    res = {"kernel": a.kernel, "command": " ".join(a.prog)}
    for i, ctrs in enumerate(PASSES):
        d = os.path.join(outdir, "%s_pass%d" % (a.tag, i))
        cmd = ["rocprofv3", "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, a.prog[0])] + a.prog[1:]
        r = subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, cwd="/tmp")
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
        res["derived"] = {
            "valu_insts_per_wave": g("SQ_INSTS_VALU") / g("SQ_WAVES"),
            "salu_insts_per_wave": (g("SQ_INSTS_SALU") or 0) / g("SQ_WAVES"),
            "lds_insts_per_wave": (g("SQ_INSTS_LDS") or 0) / g("SQ_WAVES"),
            "vmem_insts_per_wave": ((g("SQ_INSTS_VMEM_RD") or 0) + (g("SQ_INSTS_VMEM_WR") or 0)) / g("SQ_WAVES"),
            # 1024 SIMDs, a wave64 VALU instruction = 4 cycles of one SIMD; GRBM_GUI_ACTIVE is summed over the 8 XCDs
            "valu_cycles_per_simd": g("SQ_INSTS_VALU") * 4 / 1024,
            "gpu_cycles_per_launch": (g("GRBM_GUI_ACTIVE") or 0) / 8,
            "valu_busy_fraction": (g("SQ_INSTS_VALU") * 4 / 1024) / ((g("GRBM_GUI_ACTIVE") or 0) / 8) if g("GRBM_GUI_ACTIVE") else None,
            "avg_waves_per_simd": g("SQ_WAVE_CYCLES") / g("SQ_BUSY_CYCLES") / 8 if g("SQ_WAVE_CYCLES") and g("SQ_BUSY_CYCLES") else None,
        }
    json.dump(res, open(os.path.join(outdir, a.tag + ".json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
