#!/bin/bash
# A/B runs of bench.py variants in one gpurun call; each line -> gpurun_out/ab/<name>.json
# usage: tools/exp/ab.sh name1 "args1" name2 "args2" ...
mkdir -p gpurun_out/ab
while [ $# -ge 2 ]; do
  n=$1; a=$2; shift 2
  prog="bench.py"
  case "$a" in LIB=*) lib="${a%% *}"; lib="${lib#LIB=}"; a="${a#* }"; [ "$a" = "LIB=$lib" ] && a=""; prog="tools/bench_with_lib.py $lib";; esac
  timeout -k 10 240 python3 $prog --no-e2e --no-cpu-baseline --no-extra --traffic none $a > gpurun_out/ab/$n.json 2> gpurun_out/ab/$n.err || { echo "FAILED $n"; tail -5 gpurun_out/ab/$n.err; exit 1; }
  python3 - "$n" <<'PY'
import json,sys
n=sys.argv[1]
d=json.loads(open('gpurun_out/ab/%s.json'%n).read().strip().splitlines()[-1])
print(n, round(d['roofline']['kernel_avg_ms'],4), round(d['roofline']['frac'],4), d['bit_exact_vs_oracle'], flush=True)
PY
done
