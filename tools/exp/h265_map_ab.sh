set -e
python3 -m pytest tests/test_gpu_h265.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
python3 tests/soak_gpu.py --cases 3000 --seed 424242 2>&1 | tail -1
for rep in 1 2 3; do for lib in build/exp/libhevcdbk_base5.so gpu_video_codec_amd/libhevcdbk.so; do for bs in mixed; do for m in 0 4; do
  echo -n "h265 map=$m bs=$bs $(basename $lib) "; python3 tools/exp/run_with_lib.py $lib tools/bench_h265.py --qp-map $m --bs $bs --only packed --steps 300 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_launch'],4), round(d['frac_of_8TBps'],3))"
done; done; done; done
