set -e
O=gpurun_out/r02
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stripe or row_major or diagnostic" > $O/pytest_gpu_c.txt 2>&1 || { tail -40 $O/pytest_gpu_c.txt; exit 1; }
tail -2 $O/pytest_gpu_c.txt
B="timeout -k 10 300 python bench.py --traffic none --no-e2e --no-cpu-baseline --no-extra"
$B > $O/e_rows.json 2>/dev/null
$B --map stripe > $O/e_stripe.json 2>$O/e_stripe.err
$B --variant copy > $O/e_copy_rows.json 2>/dev/null
$B --variant copy --map stripe > $O/e_copy_stripe.json 2>/dev/null
$B --map stripe --frames 64 > $O/e_stripe64.json 2>/dev/null
$B --frames 64 > $O/e_rows64.json 2>/dev/null
python tools/sq_counters.py --tag stripe --map stripe > $O/sq_stripe.txt 2>&1
echo ALLDONE
