set -e
O=gpurun_out/r02
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu_b.txt 2>&1 || { tail -30 $O/pytest_gpu_b.txt; exit 1; }
tail -2 $O/pytest_gpu_b.txt
B="python bench.py --no-e2e --no-cpu-baseline --no-extra"
python tools/bench_with_lib.py gpu_video_codec_amd/libhevcdbk_prev.so --no-e2e --no-cpu-baseline --no-extra > $O/d_prev.json 2>/dev/null
$B > $O/d_new.json 2>/dev/null
python tools/bench_with_lib.py gpu_video_codec_amd/libhevcdbk_prev.so --no-e2e --no-cpu-baseline --no-extra > $O/d_prev2.json 2>/dev/null
$B > $O/d_new2.json 2>/dev/null
$B --map linear > $O/d_new_linear.json 2>/dev/null
$B --variant copy > $O/d_copy_rows.json 2>/dev/null
$B --variant copy --map linear > $O/d_copy_linear.json 2>/dev/null
$B --variant copy --map linear --diag noswz > $O/d_copy_linear_noswz.json 2>/dev/null
$B --map linear --diag noswz > $O/d_new_linear_noswz.json 2>/dev/null
$B --width 7680 --height 4320 --bit-depth 10 --frames 32 > $O/d_8k10_new.json 2>/dev/null
python tools/bench_with_lib.py gpu_video_codec_amd/libhevcdbk_prev.so --no-e2e --no-cpu-baseline --no-extra --width 7680 --height 4320 --bit-depth 10 --frames 32 > $O/d_8k10_prev.json 2>/dev/null
( time python bench.py > $O/d_full.json 2> $O/d_full.err ) 2> $O/d_full.time
python tools/sq_counters.py --tag new > $O/sq_new.txt 2>&1
echo ALLDONE
