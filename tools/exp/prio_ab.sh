python3 tools/exp/pytest_with_lib.py build/exp/libhevcdbk_t256.so tests/test_gpu_h265.py -m gpu -x -q 2>&1 | tail -1
for rep in 1 2 3; do for lib in build/exp/libhevcdbk_t192.so build/exp/libhevcdbk_t256.so; do for mode in ref; do
  echo -n "fused $mode $(basename $lib) "; python3 tools/exp/run_with_lib.py $lib tools/bench_deblock_sao.py --mode $mode --steps 300 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:round(v['ms_per_step'],4) for k,v in d.items() if isinstance(v,dict)})"
done; done; done
