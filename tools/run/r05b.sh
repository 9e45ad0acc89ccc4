mkdir -p gpurun_out/r05b
python3 tools/placement_counters.py --child --matrix --pools 6 --rounds 2 --width 3840 --height 2160 --bit-depth 8 --frames 256 --reps 20 --settle-ms 150 > gpurun_out/r05b/matrix_4k8.log 2>&1
python3 - <<'P'
import json
for l in open("gpurun_out/r05b/matrix_4k8.log"):
    if l.startswith("PLAN "):
        d=json.loads(l[5:])
        print("own dst:",[round(e["mean_ms"],4) for e in d["event_ms"]])
        print("rows = src pool, cols = dst pool, last = in place")
        for r in d["matrix_ms"]: print(r)
P
tail -3 gpurun_out/r05b/matrix_4k8.log | cut -c1-300
