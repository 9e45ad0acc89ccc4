set -e
O=gpurun_out/r02; mkdir -p $O
( time python bench.py > $O/i_full.json 2> $O/i_full.err ) 2> $O/i_full.time
tail -3 $O/i_full.time
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r02/i_full.json") if l.startswith("{")][-1])
print(d["roofline"])
PY
echo ALLDONE
