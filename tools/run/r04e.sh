set -e
mkdir -p gpurun_out/r04e
O=gpurun_out/r04e
run() { name=$1; shift; env "$@" python3 tools/host_frame_4k.py --calls 30 --threads 4 --diag --check > $O/$name.json; }
for kb in 1024 2048; do
  run base_$kb HEVCDBK_HOST_STRIP_KB=$kb
  run h2d2_$kb HEVCDBK_HOST_STRIP_KB=$kb HEVCDBK_HOST_H2D_STREAMS2=1
  run h2d2_k2_$kb HEVCDBK_HOST_STRIP_KB=$kb HEVCDBK_HOST_H2D_STREAMS2=1 HEVCDBK_HOST_K_STREAMS2=1
  run din_$kb HEVCDBK_HOST_STRIP_KB=$kb HEVCDBK_HOST_DIRECT_IN=1
  run din_k2_$kb HEVCDBK_HOST_STRIP_KB=$kb HEVCDBK_HOST_DIRECT_IN=1 HEVCDBK_HOST_K_STREAMS2=1
done
run din_512 HEVCDBK_HOST_STRIP_KB=512 HEVCDBK_HOST_DIRECT_IN=1
run din_k2_512 HEVCDBK_HOST_STRIP_KB=512 HEVCDBK_HOST_DIRECT_IN=1 HEVCDBK_HOST_K_STREAMS2=1
python3 - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04e/*.json")):
    d=json.load(open(f))
    print(f.split("/")[-1], "wall med %.0f us min %.0f us  total_s %.0f us copy %.0f exec %.0f" % (d["wall_s_median"]*1e6, d["wall_s_min"]*1e6, d["total_s_median"]*1e6, d["copy_s_median"]*1e6, d["exec_s_median"]*1e6), d.get("luma_bit_exact_vs_oracle"))
for n in ("din_1024","h2d2_1024"):
  d=json.load(open("gpurun_out/r04e/%s.json"%n))
  print(n)
  for s in d["last_call_strips"]:
    print({k:(round(v*1e6) if k.endswith("_s") else round(v*1e3) if k.endswith("_ms") else v) for k,v in s.items() if k not in ("plane","row_begin","row_end")})
P
