set -e
mkdir -p gpurun_out/r04g
O=gpurun_out/r04g
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "staging_crew or registered_caller or strip_pipeline or page_locked or streaming or small" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for t in 1 2 3 4 5 6 8; do python3 tools/host_frame_4k.py --calls 30 --threads $t --check > $O/push_t$t.json; done
python3 tools/host_frame_4k.py --calls 30 --threads 4 --fresh --check > $O/push_fresh_t4.json
python3 tools/host_frame_4k.py --calls 30 --memory registered --check > $O/registered.json
python3 tools/host_frame_4k.py --calls 30 --chroma > $O/push_yuv420_t4.json
run() { name=$1; shift; env "$@" python3 tools/host_frame_4k.py --calls 30 --threads 3 --diag --check > $O/$name.json; }
for kb in 512 768 1024 1536 2048; do
  run diag_push_strip$kb HEVCDBK_HOST_STRIP_KB=$kb
done
run diag_nopush_1024 HEVCDBK_HOST_PUSH=0 HEVCDBK_HOST_STRIP_KB=1024
run diag_nopush_2048 HEVCDBK_HOST_PUSH=0 HEVCDBK_HOST_STRIP_KB=2048
python3 - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04g/*.json")):
    d=json.load(open(f))
    print(f.split("/")[-1], "wall med %.0f us min %.0f us  total_s %.0f us copy %.0f exec %.0f" % (d["wall_s_median"]*1e6, d["wall_s_min"]*1e6, d["total_s_median"]*1e6, d["copy_s_median"]*1e6, d["exec_s_median"]*1e6), d.get("luma_bit_exact_vs_oracle"))
for n in ("push_t3",):
  d=json.load(open("gpurun_out/r04g/%s.json"%n))
  print(n)
  for s in d["last_call_strips"]:
    print({k:(round(v*1e6) if k.endswith("_s") else round(v*1e3) if k.endswith("_ms") else v) for k,v in s.items() if k not in ("plane","row_begin","row_end")})
P
