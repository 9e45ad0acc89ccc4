mkdir -p gpurun_out/r04k2
O=gpurun_out/r04k2
run() { name=$1; shift; env "$@" python3 tools/host_frame_4k.py --calls 40 --threads 4 --affinity near --diag --check > $O/$name.json; }
for rep in 1 2 3; do
  run k1_$rep HEVCDBK_HOST_THREADS=4
  run k2_$rep HEVCDBK_HOST_K_STREAMS2=1
done
for kb in 1024 2048; do run k2_strip$kb HEVCDBK_HOST_K_STREAMS2=1 HEVCDBK_HOST_STRIP_KB=$kb; run k1_strip$kb HEVCDBK_HOST_STRIP_KB=$kb; done
python3 - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04k2/*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], "wall med %.0f min %.0f" % (d["wall_s_median"]*1e6, d["wall_s_min"]*1e6), d.get("luma_bit_exact_vs_oracle"))
P
