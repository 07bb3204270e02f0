#!/bin/bash
# same-box A/B of environment settings at another batch size: tools/ab_rows.sh ROWS "VAR=a" "VAR=b" ... (two rounds)
rows=$1; shift
for r in 1 2; do
  for v in "$@"; do
    env $v python bench.py --rows $rows --steps 100 --warmup 10 --no-cpu-baseline --no-exact-f32 --no-train-step --no-in-flight 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['stages']
print('rows $rows $v', d['ms_per_step'], d['value'], ' '.join('%s=%.4f' % (k, v['ms_per_step']) for k, v in s.items()))"
  done
done
