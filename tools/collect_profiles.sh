#!/bin/bash
# Collect the round's rocprofv3 evidence ON the GPU box (one gpurun call):
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r01'
# Kernel-trace + stats runs and PMC runs are separate rocprofv3 invocations (never --pmc with a trace domain), and the
# profiled program is python3 itself.  Everything lands in gpurun_out/profiles_<tag>/; copy what should be judged into
# profiles/<tag>/.
set -e
TAG=${1:-r01}
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

run_stats bench_default_1gpu "$R/bench.py"
run_stats bench_8k10_1gpu "$R/bench.py" --width 7680 --height 4320 --bit-depth 10 --frames 48
run_stats bench_h265 "$R/tools/bench_h265.py"
run_stats e2e_small "$R/tools/e2e_small.py" --file-frames 300 --sequence-frames 256
run_stats bench_yuv420 "$R/tools/bench_yuv420.py"
run_stats bench_sao "$R/tools/bench_sao.py"
python3 - "$OUT" <<'PY'
import csv, json, sys
out = sys.argv[1]
rows = [r for r in csv.DictReader(open(out + "/kernel_trace_bench_default_1gpu.csv")) if "dbk_packed" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in rows]
n = len(d)
res = {"source": "rocprofv3 --kernel-trace of `python3 bench.py` (100 settle + 3 warm-up + 200 timed launches, plus the spot check / e2e launches)",
       "dispatches": n, "avg_all_ms": sum(d) / n, "avg_first_100_ms": sum(d[:100]) / 100, "avg_launches_104_to_303_ms": sum(d[103:303]) / 200,
       "min_ms": min(d), "max_ms": max(d)}
json.dump(res, open(out + "/kernel_trace_phases_packed_bench.json", "w"), indent=1)
print(json.dumps(res))
PY
python3 "$R/tools/hbm_traffic.py" --tag "$TAG" > "$OUT/hbm_traffic.log" 2>&1
python3 "$R/tools/hbm_traffic.py" --tag "${TAG}_8k10" --width 7680 --height 4320 --bit-depth 10 --frames 48 >> "$OUT/hbm_traffic.log" 2>&1
cp "$R"/gpurun_out/traffic/*_hbm_traffic.json "$OUT/" 2>/dev/null || true
rm -rf "$OUT"/bench_default_1gpu "$OUT"/bench_8k10_1gpu "$OUT"/bench_h265 "$OUT"/e2e_small "$OUT"/bench_yuv420 "$OUT"/bench_sao "$OUT"/kernel_trace_*.csv
ls -la "$OUT"
