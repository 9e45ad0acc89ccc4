# one gpurun call: the SAO index trims (selector constant folded into v_add3_u32 / v_and_or_b32) against the build before
# them, the fused kernel and the SAO pass; then the SAO pass with narrower workgroups (diagnostic library, knob wg=64 / 128)
set -e
python3 -m pytest tests/test_gpu_h265.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do for lib in build/exp/libhevcdbk_presao.so gpu_video_codec_amd/libhevcdbk.so; do for mode in ref h265; do
  echo -n "fused $mode $(basename $lib) "; python3 tools/exp/run_with_lib.py $lib tools/bench_deblock_sao.py --mode $mode --steps 300 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:round(v['ms_per_step'],4) for k,v in d.items() if isinstance(v,dict)})"
done; done; done
for t in mix edge band; do for lib in build/exp/libhevcdbk_presao.so gpu_video_codec_amd/libhevcdbk.so; do
  echo -n "sao $t $(basename $lib) "; python3 tools/exp/run_with_lib.py $lib tools/bench_sao.py --types $t --steps 300 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_launch'],4), round(d['frac_of_8TBps'],3))"
done; done
for t in mix edge off; do for k in wg=64 wg=128 wg=256; do
  echo -n "sao $t diag $k "; python3 tools/bench_sao.py --types $t --steps 300 --diag $k | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_launch'],4), round(d['frac_of_8TBps'],3))"
done; done
