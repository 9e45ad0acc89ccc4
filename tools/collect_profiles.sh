#!/bin/bash
# Collect the round's rocprofv3 evidence ON the GPU box (one gpurun call):
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r03'
# Kernel-trace + stats runs and PMC runs are separate rocprofv3 invocations (never --pmc with a trace domain), and the
# profiled program is python3 itself.  Everything lands in gpurun_out/profiles_<tag>/; copy what should be judged into
# profiles/<tag>/.  The bench runs under the profiler take `--traffic file`: the default (`live`) starts rocprofv3 PMC
# children of its own, which must not nest inside another rocprofv3.
set -e
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp

run_stats() { # name, then the python script and its arguments
    local name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- python3 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
    find "$OUT/$name" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_$name.csv" \;
    find "$OUT/$name" -name '*kernel_trace.csv' -exec cp {} "$OUT/kernel_trace_$name.csv" \;
    echo "== $name"; cat "$OUT/kernel_stats_$name.csv"
}

# (1) the driver's exact command, NOT under the profiler: the headline record of the round
( cd "$R" && python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err" ) || true
# (2) the same bench under rocprofv3 --kernel-trace --stats with 200 timed steps (no copy-floor child, no PMC children: they
#     must not nest inside another rocprofv3)
run_stats bench_default_1gpu "$R/bench.py" --steps 200 --warmup 5 --traffic file --copy-floor off
run_stats bench_8k10_1gpu "$R/bench.py" --traffic none --no-extra --copy-floor off --width 7680 --height 4320 --bit-depth 10 --frames 32
run_stats bench_h265 "$R/tools/bench_h265.py"
# what a decoder's operands cost: bS 0 / 1 / 2 mixed per segment, and a QP per 16x16 quantization group
python3 "$R/tools/bench_h265.py" --only packed --bs mixed > "$OUT/bench_h265_decoder_operands.json" 2>/dev/null || true
python3 "$R/tools/bench_h265.py" --only packed --qp-map 4 >> "$OUT/bench_h265_decoder_operands.json" 2>/dev/null || true
python3 "$R/tools/bench_h265.py" --only packed --qp-map 4 --bs mixed >> "$OUT/bench_h265_decoder_operands.json" 2>/dev/null || true
run_stats e2e_small "$R/tools/e2e_small.py" --file-frames 300 --sequence-frames 512
run_stats bench_sao "$R/tools/bench_sao.py"
python3 - "$OUT" <<'PY'
import csv, json, sys
out = sys.argv[1]
bench = json.loads(open(out + "/bench_default_1gpu.json").read().strip().splitlines()[-1])
settle, warm, steps = bench["config"]["settle_launches"], bench["warmup"], bench["steps"]
rows = [r for r in csv.DictReader(open(out + "/kernel_trace_bench_default_1gpu.csv")) if "dbk_packed_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in rows]
lo, hi = settle + warm, settle + warm + steps
win = sorted(d[lo:hi])
res = {"source": "rocprofv3 --kernel-trace of `python3 bench.py --steps 200 --warmup 5 --traffic file --copy-floor off`: the luma kernel's dispatches in "
                 "launch order are [settle by time][warm-up][timed], then the extra configs' launches of the same kernel",
       "settle_launches": settle, "warmup": warm, "steps": steps, "dispatches_of_the_luma_kernel": len(d),
       "avg_first_32_ms": sum(d[:32]) / 32, "avg_last_32_settle_ms": sum(d[settle - 32:settle]) / 32,
       "timed_window_avg_ms": sum(d[lo:hi]) / steps, "timed_window_p10_p50_p90_ms": [win[len(win) // 10], win[len(win) // 2], win[9 * len(win) // 10]],
       "timed_window_frac_of_8TBps": bench["roofline"]["algorithmic_bytes_per_launch"] / (sum(d[lo:hi]) / steps * 1e-3) / 8e12,
       "bench_hip_events_avg_ms": bench["roofline"]["kernel_avg_ms"]}
json.dump(res, open(out + "/kernel_trace_phases_packed_bench.json", "w"), indent=1)
with open(out + "/kernel_trace_timed_window.csv", "w", newline="") as fh:   # the raw rows of the timed window
    wtr = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
    wtr.writeheader()
    for r in rows[lo:hi]:
        wtr.writerow(r)
print(json.dumps(res))
PY
python3 "$R/tools/hbm_traffic.py" --tag "$TAG" > "$OUT/hbm_traffic.log" 2>&1
python3 "$R/tools/hbm_traffic.py" --tag "${TAG}_8k10" --width 7680 --height 4320 --bit-depth 10 --frames 32 >> "$OUT/hbm_traffic.log" 2>&1
cp "$R"/gpurun_out/traffic/*_hbm_traffic.json "$OUT/" 2>/dev/null || true
python3 "$R/tools/sq_counters.py" --tag packed_4k8 > "$OUT/sq_counters_packed_4k8.log" 2>&1 || true
python3 "$R/tools/sq_counters.py" --tag copy_4k8 --variant copy > "$OUT/sq_counters_copy_4k8.log" 2>&1 || true
cp "$R"/gpurun_out/sq/packed_4k8.json "$OUT/sq_counters_packed_4k8.json" 2>/dev/null || true
cp "$R"/gpurun_out/sq/copy_4k8.json "$OUT/sq_counters_copy_4k8.json" 2>/dev/null || true
python3 "$R/tools/clock_trace.py" 300 packed > "$OUT/clock_trace_packed_4k8.txt" 2>&1 || true
# engine clock / socket power while the kernels run (amdgpu sysfs): the filter, its copy variant, the 8K 10-bit kernel
python3 "$R/tools/clock_power_trace.py" > "$OUT/clock_power_filter_4k8.json" 2>/dev/null || true
python3 "$R/tools/clock_power_trace.py" --variant copy > "$OUT/clock_power_copy_4k8.json" 2>/dev/null || true
python3 "$R/tools/clock_power_trace.py" --width 7680 --height 4320 --bit-depth 10 --frames 32 > "$OUT/clock_power_8k10.json" 2>/dev/null || true
python3 "$R/tools/bench_yuv420.py" --bit-depth 10 --frames 48 > "$OUT/bench_yuv420_10bit.json" 2>/dev/null || true
python3 "$R/tools/bench_deblock_sao.py" > "$OUT/bench_deblock_sao.json" 2>/dev/null || true
python3 "$R/tools/bench_deblock_sao.py" --mode h265 >> "$OUT/bench_deblock_sao.json" 2>/dev/null || true
python3 "$R/tools/bench_deblock_sao.py" --frames 256 --steps 100 >> "$OUT/bench_deblock_sao.json" 2>/dev/null || true
python3 "$R/tools/exp/sao_traffic.py" > "$OUT/sao_read_traffic.json" 2>/dev/null || true
( cd "$R" && python3 tests/soak_gpu.py --cases 3000 --seed 808 > "$OUT/soak_3000_cases.txt" 2>&1 ) || true
python3 "$R/tools/fused_profile.py" > "$OUT/fused_deblock_sao_profile.json" 2>/dev/null || true
python3 "$R/tools/bench_yuv420.py" --bit-depth 10 --frames 48 --diag nofuse > "$OUT/bench_yuv420_10bit_nofuse.json" 2>/dev/null || true
"$R/tools/ubench/valu_rate" > "$OUT/ubench_valu_rate.txt" 2>&1 || true
"$R/tools/ubench/copy_bw" > "$OUT/ubench_copy_bw.txt" 2>&1 || true
rm -rf "$OUT"/bench_default_1gpu "$OUT"/bench_8k10_1gpu "$OUT"/bench_h265 "$OUT"/e2e_small "$OUT"/bench_sao "$OUT"/kernel_trace_bench_*.csv "$OUT"/kernel_trace_e2e_small.csv
ls -la "$OUT"
