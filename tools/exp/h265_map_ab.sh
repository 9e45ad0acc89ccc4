# one gpurun call: the spec-exact / reference-exact packed kernels with a QP map, build A (eight map look-ups per block)
# against the working tree (four)
set -e
python3 -m pytest tests/test_gpu_h265.py tests/test_gpu_parity.py -m gpu -x -q -k "qp or map or h265 or config3 or sao" 2>&1 | tail -2
for rep in 1 2; do for lib in build/exp/libhevcdbk_map2.so gpu_video_codec_amd/libhevcdbk.so; do for bs in 2 mixed; do for m in 3 4 6; do
  echo -n "h265 map=$m bs=$bs $(basename $lib) "; python3 tools/exp/run_with_lib.py $lib tools/bench_h265.py --qp-map $m --bs $bs --only packed --steps 300 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_launch'],4), round(d['frac_of_8TBps'],3))"
done; done; done; done
