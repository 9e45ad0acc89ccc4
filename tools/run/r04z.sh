mkdir -p gpurun_out/r04z
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/exp/sq_any.py --kernel dbk_packed_kernel --tag qpmap6_final tools/bench_qpmap.py --steps 5 --qp-map 6 > gpurun_out/r04z/sq_map.log 2>&1
python3 tools/exp/sq_any.py --kernel dbk_packed_kernel --tag qpmap0_final tools/bench_qpmap.py --steps 5 --qp-map 0 > gpurun_out/r04z/sq_uni.log 2>&1
python3 - <<'P'
import json
for t in ("qpmap6_final","qpmap0_final"):
    d=json.load(open("gpurun_out/sq/%s.json"%t))
    print(t, {k:(round(v,3) if isinstance(v,float) else v) for k,v in d["derived"].items()}, "wait_inst/wave_cycles %.3f"%(d["SQ_WAIT_INST_ANY"]/d["SQ_WAVE_CYCLES"]), "wait_any %.3f"%(d["SQ_WAIT_ANY"]/d["SQ_WAVE_CYCLES"]), "lds conflict", d.get("SQ_LDS_BANK_CONFLICT"), "active_lds", d.get("SQ_ACTIVE_INST_LDS"))
P
