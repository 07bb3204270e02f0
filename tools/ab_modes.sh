#!/bin/bash
# same-box comparison of several environment settings: tools/ab_modes.sh ROUNDS "VAR=a" "VAR=b" ...
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    env $v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-exact-f32 --no-train-step --no-in-flight 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['stages']
print('$v', d['ms_per_step'], d['roofline']['frac'], d['roofline_dual_path']['ms_per_step'], ' '.join('%s=%.4f' % (k, v['ms_per_step']) for k, v in s.items()))"
  done
done
