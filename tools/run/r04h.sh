set -e
mkdir -p gpurun_out/r04h
O=gpurun_out/r04h
python3 -c "
from gpu_video_codec_amd import shard
print('gpus', shard.gpu_pci_ids()); print('near', sorted(shard.cpus_near_gpu(0))[:4], len(shard.cpus_near_gpu(0)))"
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "staging_crew or registered_caller or strip_pipeline or page_locked" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for a in near far none; do for t in 2 3 4; do python3 tools/host_frame_4k.py --calls 30 --threads $t --affinity $a --check > $O/push_${a}_t$t.json; done; done
python3 tools/host_frame_4k.py --calls 30 --threads 4 --affinity near --fresh --check > $O/push_near_fresh_t4.json
python3 tools/host_frame_4k.py --calls 30 --memory registered --affinity near --check > $O/registered_near.json
run() { name=$1; shift; env "$@" python3 tools/host_frame_4k.py --calls 30 --threads 4 --affinity near --diag --check > $O/$name.json; }
for kb in 1024 1536 2048; do
  run diag_k1_strip$kb HEVCDBK_HOST_STRIP_KB=$kb
  run diag_k2_strip$kb HEVCDBK_HOST_STRIP_KB=$kb HEVCDBK_HOST_K_STREAMS2=1
done
run diag_k2_first512 HEVCDBK_HOST_FIRST_STRIP_KB=512 HEVCDBK_HOST_K_STREAMS2=1
run diag_k1_first512 HEVCDBK_HOST_FIRST_STRIP_KB=512
run diag_k1_first128 HEVCDBK_HOST_FIRST_STRIP_KB=128
python3 - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04h/*.json")):
    d=json.load(open(f))
    print(f.split("/")[-1], "wall med %.0f us min %.0f us  total_s %.0f us copy %.0f exec %.0f" % (d["wall_s_median"]*1e6, d["wall_s_min"]*1e6, d["total_s_median"]*1e6, d["copy_s_median"]*1e6, d["exec_s_median"]*1e6), d.get("luma_bit_exact_vs_oracle"))
for n in ("push_near_t4","diag_k2_strip1024"):
  d=json.load(open("gpurun_out/r04h/%s.json"%n))
  print(n)
  for s in d["last_call_strips"]:
    print({k:(round(v*1e6) if k.endswith("_s") else round(v*1e3) if k.endswith("_ms") else v) for k,v in s.items() if k not in ("plane","row_begin","row_end")})
P
