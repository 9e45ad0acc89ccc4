set -e
mkdir -p gpurun_out/r05c
O=gpurun_out/r05c
timeout -k 10 900 python3 -m pytest tests/test_gpu_h265.py -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for rep in 1 2; do
python3 tools/bench_h265.py --bs mixed --only packed >> $O/h265_oneqp_mixed.json
python3 tools/bench_h265.py --bs 2 --only packed >> $O/h265_oneqp_bs2.json
done
cat $O/h265_oneqp_mixed.json $O/h265_oneqp_bs2.json
