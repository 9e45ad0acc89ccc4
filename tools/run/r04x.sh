set -e
mkdir -p gpurun_out/r04x
O=gpurun_out/r04x
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "sequence or streaming" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 600 python3 bench.py --no-extra --no-cpu-baseline --traffic none --copy-floor off > $O/bench_e2e.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
python3 -c "
import json;d=json.load(open('gpurun_out/r04x/bench_e2e.json'));print(json.dumps(d['e2e_host_frame']))"
