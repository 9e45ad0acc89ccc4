set -e
mkdir -p gpurun_out/r04c
O=gpurun_out/r04c
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "staging_crew or registered_caller or strip_pipeline or page_locked or streaming" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for t in 1 2 3 4 6 8; do python3 tools/host_frame_4k.py --calls 30 --threads $t --check > $O/reuse_t$t.json; done
for t in 1 4 8; do python3 tools/host_frame_4k.py --calls 30 --threads $t --fresh > $O/fresh_t$t.json; done
python3 tools/host_frame_4k.py --calls 30 --memory registered > $O/registered.json
python3 tools/host_frame_4k.py --calls 30 --memory pinned > $O/pinned.json
HEVCDBK_HOST_STREAM_STORES=0 python3 tools/host_frame_4k.py --calls 30 --threads 4 --diag > $O/diag_t4_nostream.json
HEVCDBK_HOST_STREAM_STORES=1 python3 tools/host_frame_4k.py --calls 30 --threads 4 --diag > $O/diag_t4_stream.json
HEVCDBK_HOST_AFFINITY=0 python3 tools/host_frame_4k.py --calls 30 --threads 4 --diag > $O/diag_t4_noaffinity.json
python3 tools/host_frame_4k.py --calls 30 --chroma > $O/yuv420_t4.json
python3 - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04c/*.json")):
    d=json.load(open(f))
    print(f.split("/")[-1], "wall med %.0f us min %.0f us  total_s %.0f us copy %.0f exec %.0f" % (d["wall_s_median"]*1e6, d["wall_s_min"]*1e6, d["total_s_median"]*1e6, d["copy_s_median"]*1e6, d["exec_s_median"]*1e6), d.get("luma_bit_exact_vs_oracle"))
d=json.load(open("gpurun_out/r04c/reuse_t4.json"))
for s in d["last_call_strips"]:
    print({k:(round(v*1e6) if k.endswith("_s") else round(v*1e3) if k.endswith("_ms") else v) for k,v in s.items()})
P
