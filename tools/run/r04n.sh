set -e
mkdir -p gpurun_out/r04n
O=gpurun_out/r04n
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "qp_map or config3 or custom_tables or strip_pipeline" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for m in 6 4 3; do python3 tools/bench_qpmap.py --qp-map $m --bs lcg > $O/qpmap_$m.json; cat $O/qpmap_$m.json; done
python3 tools/bench_qpmap.py --qp-map 0 --bs lcg > $O/qpmap_none.json; cat $O/qpmap_none.json
