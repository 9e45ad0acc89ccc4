set -e
O=gpurun_out/r02; mkdir -p $O
B="timeout -k 10 300 python bench.py --no-e2e --no-cpu-baseline --no-extra --steps 50 --settle 30"
for k in dummy=0 dummy=16 dummy=24 dummy=24,align; do
  $B --variant copy --map stripe --diag $k 2>$O/g.err | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('copy $k', round(d['roofline']['kernel_avg_ms'],4))"
done
for k in dummy=0 dummy=16; do
$B --map stripe --diag $k | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('filter stripe $k', round(d['roofline']['kernel_avg_ms'],4), d['bit_exact_vs_oracle'])"
done
echo ALLDONE
