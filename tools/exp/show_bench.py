#!/usr/bin/env python3
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 4), "kernel_avg", round(d["roofline"]["kernel_avg_ms"], 4),
      "traffic", d["roofline"]["traffic"], str(d["roofline"]["traffic_source"])[:50], "bit_exact", d["bit_exact_vs_oracle"])
for k, v in d.get("extra_configs", {}).items():
    print(" ", k, round(v["ms_per_step"], 4), round(v["frac"], 4), v.get("bit_exact_vs_oracle"))
cb = d.get("cpu_baseline")
if cb:
    print("  cpu", round(cb["value"], 1), [(r["threads"], round(r["frames_per_s"])) for r in cb["thread_ladder"]])
if "e2e_host_frame" in d:
    print("  e2e", {k: (round(v, 6) if isinstance(v, float) else v) for k, v in d["e2e_host_frame"].items()})
if "reference_table_352x288" in d:
    print("  table gpu", d["reference_table_352x288"]["gpu"])
