set -e
mkdir -p gpurun_out/r04k
O=gpurun_out/r04k
true
python3 - <<"P"
import json
d=json.load(open("gpurun_out/r04k/bench_driver_cmd.json"))
print("value", d["value"], "frac", d["roofline"]["frac"], "bit_exact", d["bit_exact_vs_oracle"])
print("e2e", json.dumps(d.get("e2e_host_frame")))
print("cross", d.get("cross_rank")); print("per_rank", d["per_rank"])
for k,v in d.get("extra_configs",{}).items():
    print(k, {kk:vv for kk,vv in v.items() if kk in ("ms_per_step","frac","bit_exact_vs_oracle","scalar_qp30_us_per_call","ctu_qp_map_12x9_us_per_call","cpu_port_1t_median_s","gpu_device_resident_us_per_call")})
    for kk,vv in v.items():
        if isinstance(vv,dict): print("   ",kk,{a:b for a,b in vv.items() if a in ("ms_per_step","frac","bit_exact_vs_oracle")})
P
timeout -k 10 400 python3 bench.py --gpus 2 --oversubscribe --steps 20 --warmup 5 --frames 64 --no-extra --no-cpu-baseline --copy-floor off --traffic none > $O/bench_2ranks_one_gpu_oversubscribed.json 2> $O/bench_2ranks.err || { tail -20 $O/bench_2ranks.err; exit 1; }
python3 - <<"P"
import json
d=json.load(open("gpurun_out/r04k/bench_2ranks_one_gpu_oversubscribed.json"))
print("2 ranks: value", d["value"], "cross", d.get("cross_rank")); print("e2e", json.dumps(d.get("e2e_host_frame")))
for r in d["per_rank"]: print(r)
P
