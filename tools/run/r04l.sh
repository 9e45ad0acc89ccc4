set -e
mkdir -p gpurun_out/r04k
O=gpurun_out/r04k
timeout -k 10 400 python3 bench.py --gpus 2 --oversubscribe --steps 20 --warmup 5 --frames 64 --no-extra --no-cpu-baseline --copy-floor off --traffic none > $O/bench_2ranks_one_gpu_oversubscribed.json 2> $O/bench_2ranks.err || { tail -20 $O/bench_2ranks.err; exit 1; }
python3 - <<"P"
import json
d=json.load(open("gpurun_out/r04k/bench_2ranks_one_gpu_oversubscribed.json"))
print("2 ranks: value", d["value"], "cross", d.get("cross_rank")); print("e2e", json.dumps(d.get("e2e_host_frame")))
for r in d["per_rank"]: print(r)
P
