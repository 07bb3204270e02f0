# measurement: dispatch orders of the fused chain launch (tools/chain_bench.hip CHAIN_ORDER=band[:workgroups],...)
for c in 0 1; do
  echo "== chain $c"
  for o in "" "9,10,8,7,6,5,4,3,2,1,0" "9,8,10,7,6,5,4,3,2,1,0" "9,10:88,8,10,7,6,5,4,3,2,1,0" "9,8:88,10,8,7,6,5,4,3,2,1,0" "9,7,6,5,4,10,8,3,2,1,0" "9,10:88,6,5,4,3,2,1,0,7,10,8" "10,9,8,7,6,5,4,3,2,1,0" "9,10:44,8:44,10,8,7,6,5,4,3,2,1,0" "9,6:32,5:32,4:24,10,8,7,6,5,4,3,2,1,0"; do
    printf "%-44s " "order '$o':"; CHAIN_ORDER=$o build/chain_bench 8064 $c all | tail -1 | sed 's/.*ABL 0: *//'
  done
done
