mkdir -p gpurun_out/r04s
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/exp/sq_any.py --kernel dbk_packed_h265_kernel --tag h265_map4_tab tools/bench_h265.py --steps 5 --qp-map 4 --bs mixed --only packed > gpurun_out/r04s/sq_h265_map4.log 2>&1
python3 tools/exp/sq_any.py --kernel dbk_sao_fused_h265_kernel --tag fused_h265_map_tab tools/bench_deblock_sao.py --steps 5 --mode h265 --qp-map 4 > gpurun_out/r04s/sq_fused_h265_map.log 2>&1
python3 tools/bench_deblock_sao.py --mode h265 --qp-map 4 > gpurun_out/r04s/fused_h265_map4.json 2>&1
python3 tools/bench_deblock_sao.py --mode ref > gpurun_out/r04s/fused_ref.json 2>&1
grep -A12 derived gpurun_out/r04s/sq_h265_map4.log; grep -A12 derived gpurun_out/r04s/sq_fused_h265_map.log; cat gpurun_out/r04s/fused_h265_map4.json gpurun_out/r04s/fused_ref.json
