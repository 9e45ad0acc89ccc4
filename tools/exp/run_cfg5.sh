for l in build/exp/libhevcdbk_r02.so build/exp/libhevcdbk_path3.so gpu_video_codec_amd/libhevcdbk.so build/exp/libhevcdbk_r02.so gpu_video_codec_amd/libhevcdbk.so; do
  python3 tools/bench_with_lib.py $l --width 7680 --height 4320 --bit-depth 10 --frames 32 --steps 100 --no-extra --no-e2e --no-cpu-baseline --traffic none --copy-floor off 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$l', round(r['kernel_avg_ms'],4), round(r['frac'],4), d['bit_exact_vs_oracle'])"
done
