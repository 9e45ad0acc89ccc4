#!/bin/bash
# A/B runs of bench.py variants in one gpurun call; each line -> gpurun_out/ab/<name>.json
# usage: tools/exp/ab.sh name1 "args1" name2 "args2" ...
mkdir -p gpurun_out/ab
while [ $# -ge 2 ]; do
  n=$1; a=$2; shift 2
  prog="bench.py"
  case "$a" in LIB=*) lib="${a%% *}"; lib="${lib#LIB=}"; a="${a#* }"; [ "$a" = "LIB=$lib" ] && a=""; prog="tools/bench_with_lib.py $lib";; esac
  timeout -k 10 240 python3 $prog --no-e2e --no-cpu-baseline --no-extra --traffic none $a > gpurun_out/ab/$n.json 2> gpurun_out/ab/$n.err || { [ -s gpurun_out/ab/$n.json ] || { echo "FAILED $n"; tail -5 gpurun_out/ab/$n.err; exit 1; }; }
  python3 - "$n" <<'PY'
import json,sys
n=sys.argv[1]
d=json.loads(open('gpurun_out/ab/%s.json'%n).read().strip().splitlines()[-1])
r=d['roofline']
print(n, round(r['kernel_avg_ms'],4), round(r['frac'],4), d['bit_exact_vs_oracle'], 'sclk', r.get('engine_clock_MHz'), 'W', r.get('socket_power_W'), 'cap', r.get('power_cap_W'), 'n', r.get('telemetry_samples'), flush=True)
PY
done
