# Sweep of the row-block pipeline (BSRNN_PARTS / BSRNN_PART_LAG) at the metric configuration; run on the GPU box.
for cfg in "1 0" "2 0" "2 2" "2 4" "2 6" "1 0"; do set -- $cfg
BSRNN_PARTS=$1 BSRNN_PART_LAG=$2 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('parts $1 lag $2:', d['ms_per_step'], d['value'])"
done
