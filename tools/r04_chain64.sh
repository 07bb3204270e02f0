cd $GRAFT_REPO_ROOT
for ch in 0 1; do
  echo "== chain $ch band 10, 32x32 geometry (BSRNN_CHAIN_NO64=1)"; BSRNN_CHAIN_NO64=1 timeout -k 5 60 ./build/cb/chain_bench_pd3tp2 8064 $ch 10 | tail -2
  for v in pd3tp2 pd2tp2 pd8tp1 pd6tp1 pd4tp1; do echo "== chain $ch band 10, four row tiles of 16, $v"; timeout -k 5 60 ./build/cb/chain_bench_$v 8064 $ch 10 | tail -2; done
done
