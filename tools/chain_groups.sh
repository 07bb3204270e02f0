# measurement: one band group's workgroups of the fused chain launch alone (tools/chain_bench.hip): duration of one workgroup per group
set -e
for c in 0 1; do
for b in 9 10 8 7 0,1,2,3,4,5,6 6 5 0 7,0,1,2,3,4,5,6 all; do
  build/chain_bench 8064 $c $b | tail -1
done; done
