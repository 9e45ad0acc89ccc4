set -e
mkdir -p gpurun_out/r05e
O=gpurun_out/r05e
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "probed" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for rep in 1 2; do
timeout -k 10 600 python3 bench.py --no-e2e --no-cpu-baseline --traffic none --copy-floor off > $O/bench_$rep.json 2> $O/bench_$rep.err || { tail $O/bench_$rep.err; exit 1; }
python3 -c "
import json,sys;d=json.load(open('gpurun_out/r05e/bench_$rep.json'))['extra_configs']
for k in ('config5_8k_10bit','config5_8k_10bit_fresh_pool','config5_8k_10bit_probed_pool'):
    print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in d[k].items() if a in ('ms_per_step','frac','bit_exact_vs_oracle','probe_best_ms','probe_worst_ms')})"
done
