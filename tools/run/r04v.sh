mkdir -p gpurun_out/r04v
python3 tools/placement_counters.py --child --matrix --pools 5 --rounds 1 > gpurun_out/r04v/matrix.log 2>&1
python3 - <<'P'
import json
for l in open("gpurun_out/r04v/matrix.log"):
    if l.startswith("PLAN "):
        d=json.loads(l[5:])
        print("diag (own dst):",[e["mean_ms"] for e in d["event_ms"]])
        print("rows = src pool, cols = dst pool 0..4, last = in place")
        for r in d["matrix_ms"]: print(r)
P
