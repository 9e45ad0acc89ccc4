set -e
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_f.txt 2>&1 || { tail -40 $O/pytest_gpu_f.txt; exit 1; }
tail -2 $O/pytest_gpu_f.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail -20 $O/smoke.txt; exit 1; }
tail -2 $O/smoke.txt
( time python bench.py > $O/h_full.json 2> $O/h_full.err ) 2> $O/h_full.time
cat $O/h_full.time | tail -3
python bench.py --gpus 2 --steps 50 > $O/h_2rank.json 2> $O/h_2rank.err
python tools/bench_yuv420.py > $O/h_yuv420.json 2>&1
python tools/bench_sao.py > $O/h_sao.json 2>&1
python tools/bench_sao.py --bit-depth 10 >> $O/h_sao.json 2>&1
echo ALLDONE
