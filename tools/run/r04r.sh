set -e
mkdir -p gpurun_out/r04r
O=gpurun_out/r04r
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_h265.py -x -q -k "qp_map or config3 or custom_tables or strip_pipeline or h265 or sao or map" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for m in 6 4; do python3 tools/bench_qpmap.py --qp-map $m --bs lcg > $O/qpmap_$m.json; cat $O/qpmap_$m.json; done
python3 tools/bench_h265.py --qp-map 4 --bs 2 --only packed > $O/h265_map4.json; cat $O/h265_map4.json
python3 tools/bench_h265.py --qp-map 4 --bs mixed --only packed > $O/h265_map4_mixed.json; cat $O/h265_map4_mixed.json
python3 tools/bench_h265.py --qp-map 6 --bs mixed --only packed > $O/h265_map6_mixed.json; cat $O/h265_map6_mixed.json
python3 tools/bench_h265.py --bs mixed --only packed > $O/h265_oneqp_mixed.json; cat $O/h265_oneqp_mixed.json
