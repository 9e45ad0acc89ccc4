set -e
mkdir -p gpurun_out/r04w
O=gpurun_out/r04w
timeout -k 10 900 python3 -m pytest tests/test_gpu_h265.py tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
python3 tools/host_frame_4k.py --calls 20 --h265 --affinity near --check > $O/h265_host_4k.json; cat $O/h265_host_4k.json | cut -c1-700
python3 tools/host_frame_4k.py --calls 20 --h265 --chroma --affinity near > $O/h265_host_4k_420.json; cat $O/h265_host_4k_420.json | cut -c1-500
python3 tools/host_frame_4k.py --calls 20 --h265 --width 352 --height 288 --chroma --affinity near > $O/h265_host_cif.json; cat $O/h265_host_cif.json | cut -c1-500
