mkdir -p gpurun_out/r04u
O=gpurun_out/r04u
timeout -k 10 600 python3 -m pytest tests/test_gpu_h265.py -x -q -k "two_caller_streams or 16bit_4k_repeated" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 tools/placement_counters.py --tag cfg5_set1 --set 1 > $O/placement_set1.log 2>&1
timeout -k 10 500 python3 tools/placement_counters.py --tag cfg5_set1_junk --set 1 --interleave-junk > $O/placement_set1_junk.log 2>&1
timeout -k 10 500 python3 tools/placement_counters.py --tag cfg5_set2 --set 2 > $O/placement_set2.log 2>&1
python3 - <<'P'
import json
for t in ("cfg5_set1","cfg5_set1_junk","cfg5_set2"):
    try:
        d=json.load(open("gpurun_out/placement/%s.json"%t))
    except Exception as e:
        print(t,"failed",e); continue
    print(t, d["spread_event_ms"], {k:v for k,v in d.items() if k.startswith("corr")}, d["dispatches_seen"], d["dispatches_planned"])
    for p in d["pools"]:
        print("  ", {k:(round(v,4) if isinstance(v,float) else v) for k,v in p.items() if k not in ("src","dst")})
P
