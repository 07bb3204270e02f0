#!/bin/bash
# Same-box comparison of several builds of libbsrnn_hip.so: tools/ab_many.sh rounds lib1.so lib2.so ...   (each round runs bench.py once per library)
R=$1; shift
LIB=speechseparation_amd/lib/libbsrnn_hip.so
cp $LIB /tmp/lib_keep.so
for r in $(seq 1 $R); do
  for L in "$@"; do
    cp $L $LIB
    python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-exact-f32 --no-train-step --no-in-flight 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['stages']
print('$L', d['ms_per_step'], ' '.join('%s=%.4f' % (k, v['ms_per_step']) for k, v in s.items()))"
  done
done
cp /tmp/lib_keep.so $LIB
