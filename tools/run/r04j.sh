set -e
mkdir -p gpurun_out/r04j
O=gpurun_out/r04j
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "staging_crew or registered_caller or strip_pipeline or page_locked" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for t in 2 3 4; do python3 tools/host_frame_4k.py --calls 30 --threads $t --affinity near --check > $O/push_near_t$t.json; done
python3 tools/host_frame_4k.py --calls 30 --threads 4 --check > $O/push_none_t4.json
python3 tools/host_frame_4k.py --calls 30 --memory registered --affinity near --check > $O/registered_near.json
python3 tools/host_frame_4k.py --calls 30 --memory pinned --affinity near --check > $O/pinned_near.json
python3 tools/host_frame_4k.py --calls 30 --chroma --affinity near > $O/yuv420_near.json
python3 tools/host_frame_4k.py --calls 30 --width 1920 --height 1080 --affinity near > $O/luma1080_near.json
run() { name=$1; shift; env "$@" python3 tools/host_frame_4k.py --calls 30 --threads 4 --affinity near --diag --check > $O/$name.json; }
for kb in 768 1024 1536 2048; do
  run diag_k1_strip$kb HEVCDBK_HOST_STRIP_KB=$kb
  run diag_k2_strip$kb HEVCDBK_HOST_STRIP_KB=$kb HEVCDBK_HOST_K_STREAMS2=1
done
python3 - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04j/*.json")):
    d=json.load(open(f))
    print(f.split("/")[-1], "wall med %.0f us min %.0f us  total_s %.0f us copy %.0f exec %.0f" % (d["wall_s_median"]*1e6, d["wall_s_min"]*1e6, d["total_s_median"]*1e6, d["copy_s_median"]*1e6, d["exec_s_median"]*1e6), d.get("luma_bit_exact_vs_oracle"))
for n in ("push_near_t4",):
  d=json.load(open("gpurun_out/r04j/%s.json"%n))
  print(n)
  for s in d["last_call_strips"]:
    print({k:(round(v*1e6) if k.endswith("_s") else round(v*1e3) if k.endswith("_ms") else v) for k,v in s.items() if k not in ("plane","row_begin","row_end")})
P
