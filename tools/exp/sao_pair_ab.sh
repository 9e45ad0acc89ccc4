python3 -m pytest tests/test_gpu_h265.py -m gpu -x -q 2>&1 | tail -1
python3 tests/soak_gpu.py --cases 1200 --seed 8086 2>&1 | tail -1
for rep in 1 2 3; do for lib in gpu_video_codec_amd/libhevcdbk.so build/exp/libhevcdbk_base4.so; do for t in "edge" "mix" "mix --merge" "off"; do
  echo -n "sao $t $(basename $lib) "; python3 tools/exp/run_with_lib.py $lib tools/bench_sao.py --types $t --steps 300 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_launch'],4), round(d['frac_of_8TBps'],3))"
done; done; done
