set -e
mkdir -p gpurun_out/r04t
O=gpurun_out/r04t
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err || { tail -20 $O/bench_driver_cmd.err; exit 1; }
python3 - <<'P'
import json
d=json.load(open("gpurun_out/r04t/bench_driver_cmd.json"))
print("value", d["value"], "frac", d["roofline"]["frac"], "bit_exact", d["bit_exact_vs_oracle"])
e=d.get("e2e_host_frame"); print("e2e", {k:e[k] for k in ("frames_per_s","wall_s","total_s","sequence_frames_per_s")})
for k,v in d.get("extra_configs",{}).items():
    print(k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("ms_per_step","frac","bit_exact_vs_oracle","ms_per_step_two_launches")})
    for kk,vv in v.items():
        if isinstance(vv,dict): print("   ",kk,{a:(round(b,4) if isinstance(b,float) else b) for a,b in vv.items() if a in ("ms_per_step","frac","bit_exact_vs_oracle")})
P
