set -e
O=gpurun_out/r02
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu_a.txt 2>&1 || { tail -30 $O/pytest_gpu_a.txt; exit 1; }
tail -3 $O/pytest_gpu_a.txt
B="python bench.py --no-e2e --no-cpu-baseline"
$B > $O/c_prod.json 2>/dev/null
$B --map linear > $O/c_prod_linear.json 2>/dev/null
for k in mode3 prio=1 prio=2 prio=3 dummy=40 dummy=80 mode3,wg=64 mode3,wg=128 mode3,wg=256 mode3,wg=1024; do
  $B --diag $k > $O/c_diag_$k.json 2>/dev/null
done
$B --diag mode3,wg=128 --map linear > $O/c_diag_wg128_linear.json 2>/dev/null
$B --diag mode3,wg=64 --map linear > $O/c_diag_wg64_linear.json 2>/dev/null
python tools/sq_counters.py --tag rows > $O/sq_rows.txt 2>&1
python tools/sq_counters.py --tag linear --map linear > $O/sq_linear.txt 2>&1
echo ALLDONE
