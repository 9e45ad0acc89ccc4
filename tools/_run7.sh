set -e
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_h265.py -m gpu -x -q -k "sao" > $O/pytest_gpu_g.txt 2>&1 || { tail -40 $O/pytest_gpu_g.txt; exit 1; }
tail -2 $O/pytest_gpu_g.txt
python tools/bench_sao.py
python tools/bench_sao.py --bit-depth 10
echo ALLDONE
