#!/bin/bash
# same-box A/B of an environment setting: tools/ab_env.sh "VAR=a" "VAR=b" [rounds]
A=$1; B=$2; R=${3:-3}
for r in $(seq 1 $R); do
  for v in "$A" "$B"; do
    env $v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-exact-f32 --no-train-step --no-in-flight 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['stages']
print('$v', d['ms_per_step'], d['roofline']['frac'], d['roofline_dual_path']['ms_per_step'], ' '.join('%s=%.4f' % (k, v['ms_per_step']) for k, v in s.items()))"
  done
done
