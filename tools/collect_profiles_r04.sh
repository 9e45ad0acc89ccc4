#!/bin/bash
# Round-4 additions to tools/collect_profiles.sh (run ON the GPU box, its own gpurun call):
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/collect_profiles_r04.sh'
# the host-frame operator, the N-rank bench rehearsal (item 2), the QP-map kernels (item 4).
# Never --pmc together with a trace domain; the profiled program is python3 itself.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/profiles_r04x
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
H="python3 $R/tools/host_frame_4k.py"
# (1) the reference-shaped host call on one pageable 4K luma frame: wall clock, the operator's figures, the per-strip record
$H --calls 30 --affinity near --check > "$OUT/host_frame_4k_pageable.json"
$H --calls 30 --affinity near --fresh --check > "$OUT/host_frame_4k_pageable_new_buffer_per_call.json"
$H --calls 30 --affinity far --check > "$OUT/host_frame_4k_pageable_caller_on_far_socket.json"
$H --calls 30 --check > "$OUT/host_frame_4k_pageable_unpinned_caller.json"
for t in 1 2 3 4 6 8; do $H --calls 30 --affinity near --threads $t > "$OUT/host_frame_4k_threads_$t.json"; done
$H --calls 30 --affinity near --memory registered --check > "$OUT/host_frame_4k_registered.json"
$H --calls 30 --affinity near --memory pinned --check > "$OUT/host_frame_4k_page_locked.json"
$H --calls 30 --affinity near --chroma > "$OUT/host_frame_4k_yuv420.json"
$H --calls 30 --affinity near --width 1920 --height 1080 > "$OUT/host_frame_1080p.json"
$H --calls 30 --affinity near --h265 --check > "$OUT/host_frame_4k_h265.json"
#     ... and what the GPU saw: kernels only, no memory copy (the DMA engines are not used on a large-BAR device)
rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d "$OUT/host_frame_4k_trace" -- python3 "$R/tools/host_frame_4k.py" --calls 5 --affinity near \
    > "$OUT/host_frame_4k_under_rocprof.json" 2> "$OUT/host_frame_4k_under_rocprof.err" || true
find "$OUT/host_frame_4k_trace" -name '*kernel_trace.csv' -exec cp {} "$OUT/host_frame_4k_kernel_trace.csv" \;
find "$OUT/host_frame_4k_trace" -name '*memory_copy_trace.csv' -exec cp {} "$OUT/host_frame_4k_memory_copy_trace.csv" \;
#     ... the same through ring + DMA (diagnostic build: the BAR path switched off), for the DMA engine's view of that form
HEVCDBK_HOST_PUSH=0 python3 "$R/tools/host_frame_4k.py" --calls 30 --affinity near --diag > "$OUT/host_frame_4k_no_bar_ring_dma_in.json" || true
HEVCDBK_HOST_PUSH=0 HEVCDBK_HOST_DIRECT_OUT=0 python3 "$R/tools/host_frame_4k.py" --calls 30 --affinity near --diag > "$OUT/host_frame_4k_no_bar_dma_both_ways.json" || true
"$R/tools/ubench/host_stage" > "$OUT/ubench_host_stage.json" 2>/dev/null || true
"$R/tools/ubench/bar_write" > "$OUT/ubench_bar_write.json" 2>/dev/null || true
# (2) the N-rank bench on one card (rehearsal of the 8-GPU line: every field filled, not a scaling figure)
( cd "$R" && python3 bench.py --gpus 2 --oversubscribe --steps 20 --warmup 5 --frames 64 --no-extra --no-cpu-baseline --copy-floor off --traffic none \
    > "$OUT/bench_2ranks_one_gpu_oversubscribed.json" 2> "$OUT/bench_2ranks.err" ) || true
# (3) the QP-map kernels: timings and SQ counters
for m in 6 4 3; do python3 "$R/tools/bench_qpmap.py" --qp-map $m --bs lcg >> "$OUT/bench_qpmap.json"; done
python3 "$R/tools/bench_qpmap.py" --qp-map 0 --bs lcg >> "$OUT/bench_qpmap.json"
python3 "$R/tools/bench_h265.py" --only packed --qp-map 4 --bs mixed > "$OUT/bench_h265_qpmap.json" 2>/dev/null || true
python3 "$R/tools/bench_h265.py" --only packed --qp-map 6 --bs mixed >> "$OUT/bench_h265_qpmap.json" 2>/dev/null || true
python3 "$R/tools/sq_of.py" --kernel dbk_packed_kernel --tag r04_ref_qpmap --more -- tools/bench_qpmap.py --steps 5 --qp-map 6 > "$OUT/sq_ref_qpmap.log" 2>&1 || true
python3 "$R/tools/sq_of.py" --kernel dbk_packed_kernel --tag r04_ref_one_qp --more -- tools/bench_qpmap.py --steps 5 --qp-map 0 > "$OUT/sq_ref_one_qp.log" 2>&1 || true
python3 "$R/tools/sq_of.py" --kernel dbk_packed_h265_kernel --tag r04_h265_qpmap --more -- tools/bench_h265.py --steps 5 --qp-map 4 --bs mixed --only packed > "$OUT/sq_h265_qpmap.log" 2>&1 || true
cp "$R"/gpurun_out/sq/r04_*.json "$OUT/" 2>/dev/null || true
echo done
