for p in 1 2 3; do for lag in 0 1 2; do
  if [ $p = 1 ] && [ $lag != 0 ]; then continue; fi
  echo -n "parts=$p lag=$lag: "
  BSRNN_PARTS=$p BSRNN_PART_LAG=$lag python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done; done
