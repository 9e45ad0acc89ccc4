set -e
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile_map or device_planes or sharded" > $O/pytest_gpu_e.txt 2>&1 || { tail -40 $O/pytest_gpu_e.txt; exit 1; }
tail -2 $O/pytest_gpu_e.txt
B="timeout -k 10 300 python bench.py --traffic none --no-e2e --no-cpu-baseline --no-extra --steps 100 --settle 50"
$B --variant copy --map tiles 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('copy tiles', round(d['roofline']['kernel_avg_ms'],4), d['bit_exact_vs_oracle'])"
$B --variant copy 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('copy rows', round(d['roofline']['kernel_avg_ms'],4), d['bit_exact_vs_oracle'])"
$B --map tiles | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('filter tiles', round(d['roofline']['kernel_avg_ms'],4), d['bit_exact_vs_oracle'], round(d['roofline']['frac'],4))"
$B | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('filter rows', round(d['roofline']['kernel_avg_ms'],4), d['bit_exact_vs_oracle'], round(d['roofline']['frac'],4))"
echo ALLDONE
