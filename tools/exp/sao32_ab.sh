set -e
python3 -m pytest tests/test_gpu_h265.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do for lib in build/exp/libhevcdbk_presao32.so gpu_video_codec_amd/libhevcdbk.so; do for c in 5 6; do for t in mix edge; do
  echo -n "sao ctb=$c $t $(basename $lib) "; python3 tools/exp/run_with_lib.py $lib tools/bench_sao.py --types $t --ctb-log2 $c --steps 300 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_launch'],4), round(d['frac_of_8TBps'],3))"
done; done; done; done
python3 tests/soak_gpu.py --cases 10000 --seed 20261004 > gpurun_out/r03/soak_10000_cases.txt 2>&1; tail -1 gpurun_out/r03/soak_10000_cases.txt
