# is the SAO pass's "mix" time a property of the build (code layout) or of the run?  product and diagnostic library alternating
for rep in 1 2 3 4; do
  echo -n "mix product "; python3 tools/bench_sao.py --types mix --steps 300 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_launch'],4), round(d['frac_of_8TBps'],3))"
  echo -n "mix diag    "; python3 tools/bench_sao.py --types mix --steps 300 --diag "" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_launch'],4), round(d['frac_of_8TBps'],3))"
done
