export TMPDIR=/tmp
for o in 0 1; do
export BSRNN_GEMM_SLAB=$o
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_slab$o -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
f=$(ls gpurun_out/prof_slab$o/*/*kernel_stats.csv | head -1)
echo "== slab=$o"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'gemm' in n: print("%-44s calls %5s avg %8.1f us" % (n.replace('bsrnn::','').replace('void ','')[:44], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
