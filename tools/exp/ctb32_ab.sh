set -e
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2
python3 tests/soak_gpu.py --cases 1200 --seed 515 2>&1 | tail -1
for rep in 1 2; do for lib in build/exp/libhevcdbk_prectb32.so gpu_video_codec_amd/libhevcdbk.so; do
  echo "bench extras $(basename $lib)"; python3 tools/exp/run_with_lib.py $lib bench.py --steps 20 --warmup 5 --copy-floor off --no-cpu-baseline --no-e2e 2>/dev/null | python3 tools/exp/show_bench.py /dev/stdin | grep -E "fused|qpmap"
done; done
