mkdir -p gpurun_out/r05d
O=gpurun_out/r05d
python3 tools/placement_counters.py --child --matrix --pools 6 --rounds 1 > $O/m_filter.log 2>&1
python3 tools/placement_counters.py --child --matrix --pools 6 --rounds 1 --variant copy > $O/m_copy.log 2>&1
python3 tools/placement_counters.py --child --matrix --pools 6 --rounds 1 --variant copy --diag align > $O/m_copy_align.log 2>&1
python3 - <<'P'
import json
for t in ("m_filter","m_copy","m_copy_align"):
    print(t)
    for l in open("gpurun_out/r05d/%s.log"%t):
        if l.startswith("PLAN "):
            d=json.loads(l[5:])
            for r in d["matrix_ms"]: print("  ",r)
    print(open("gpurun_out/r05d/%s.log"%t).read()[-300:] if "PLAN" not in open("gpurun_out/r05d/%s.log"%t).read() else "")
P
