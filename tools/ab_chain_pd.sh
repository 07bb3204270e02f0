#!/bin/bash
# A/B of the fused chain kernel's prefetch depth on one box: rebuilds mlp_chain.o per value and prints the two MLP stage times.
# usage: tools/ab_chain_pd.sh 6 8 12        (prefetch depths)
#        tools/ab_chain_pd.sh 8:1 8:2 8:3    (depth:ablation - CHAIN_ABL of mlp_chain.hip; the results of those builds are wrong)
set -e
cd "$(dirname "$0")/.."
for pd in "$@"; do
  rm -f build/obj/mlp_chain.o
  abl=0; case "$pd" in *:*) abl=${pd#*:}; pd=${pd%:*};; esac
  make -C speechseparation_amd/csrc CHAIN_PD=$pd CHAIN_ABL=$abl > /dev/null 2>&1
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-exact-f32 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PD=$pd ABL=$abl ms/step %.4f  bandsplit %.4f  mask %.4f' % (d['ms_per_step'], d['stages']['bandsplit_mlp']['ms_per_step'], d['stages']['mask_mlp']['ms_per_step']))"
done
rm -f build/obj/mlp_chain.o; make -C speechseparation_amd/csrc > /dev/null 2>&1    # back to the product build
