#!/bin/bash
# Same-box A/B of two builds of libbsrnn_hip.so (boxes differ by +-5 %): tools/ab_libs.sh base.so variant.so [rounds]
# Each round runs bench.py once per library, alternating; prints ms/step and the stage times.
set -e
A=$1; B=$2; R=${3:-3}
LIB=speechseparation_amd/lib/libbsrnn_hip.so
cp $LIB /tmp/lib_keep.so
for r in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then cp $A $LIB; else cp $B $LIB; fi
    python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-exact-f32 --no-train-step --no-in-flight 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['stages']
print('$v', d['ms_per_step'], ' '.join('%s=%.4f' % (k, v['ms_per_step']) for k, v in s.items()))"
  done
done
cp /tmp/lib_keep.so $LIB
