#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/qpmap2
timeout -k 10 900 python3 -m pytest tests/test_gpu_h265.py tests/test_gpu_parity.py -x -q -k "qp_map or h265 or sao or fused or map" > gpurun_out/qpmap2/pytest.log 2>&1 || { tail -30 gpurun_out/qpmap2/pytest.log; exit 1; }
tail -2 gpurun_out/qpmap2/pytest.log
for i in 1 2; do
timeout -k 10 300 python3 tools/bench_h265.py --qp-map 4 --bs mixed --only packed | tee -a gpurun_out/qpmap2/h265_map_mixed.json
timeout -k 10 300 python3 tools/bench_h265.py --bs mixed --only packed | tee -a gpurun_out/qpmap2/h265_oneqp_mixed.json
timeout -k 10 300 python3 tools/bench_h265.py --bs 2 --only packed | tee -a gpurun_out/qpmap2/h265_oneqp_bs2.json
timeout -k 10 300 python3 tools/bench_qpmap.py | tee -a gpurun_out/qpmap2/ref_qpmap.json
done
timeout -k 10 300 python3 tools/bench_deblock_sao.py --mode h265 | tee gpurun_out/qpmap2/fused_h265.json
