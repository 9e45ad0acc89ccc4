#!/bin/bash
for t in off band edge mix; do
  for lib in build/exp/libhevcdbk_oldsao.so gpu_video_codec_amd/libhevcdbk.so; do
    echo -n "$t $(basename $lib) "; python3 tools/exp/run_with_lib.py $lib tools/bench_sao.py --types $t | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_launch'],4), round(d['frac_of_8TBps'],3))"
  done
done
