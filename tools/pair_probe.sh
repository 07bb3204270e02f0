#!/bin/bash
# Paired geometry of the 768-wide band (mlp_chain.hip::chain_body_pair) against the 48-row geometry, band 9 alone and the whole chains:
# durations and agreement of the outputs.  Binaries: build/chain_bench (hand-over behind the first own pass), _early (in front of it),
# _tp2 (two feature tiles per pass), _abl3 (no exchange: durations only).
mkdir -p gpurun_out build
HC="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize"
[ -x build/chain_bench ] || $HC -o build/chain_bench tools/chain_bench.hip || exit 1
[ -x build/chain_bench_early ] || $HC -DCHAIN_PAIR_LATE=0 -o build/chain_bench_early tools/chain_bench.hip || exit 1
[ -x build/chain_bench_tp2 ] || $HC -DCHAIN_PAIR_TPS=2 -DCHAIN_PAIR_TPM=2 -o build/chain_bench_tp2 tools/chain_bench.hip || exit 1
[ -x build/chain_bench_abl3 ] || $HC -DCHAIN_PAIR_ABL=3 -o build/chain_bench_abl3 tools/chain_bench.hip || exit 1
out=gpurun_out/pair_probe.txt
: > $out
for M in 7680 1280; do
for chain in 0 1; do
  echo "== M $M chain $chain band 9: 48-row geometry" >> $out
  CHAIN_DUMP=/tmp/ref_$chain.bin timeout -k 10 120 build/chain_bench $M $chain 9 >> $out 2>&1 || exit 1
  for b in chain_bench chain_bench_early chain_bench_tp2 chain_bench_abl3; do
    echo "== M $M chain $chain band 9: pairs ($b)" >> $out
    CHAIN_PAIR=1 CHAIN_DUMP=/tmp/pair_$chain.bin timeout -k 10 120 build/$b $M $chain 9 >> $out 2>&1 || exit 1
    python3 - $chain >> $out <<'PY'
import sys, numpy as np
c = sys.argv[1]
a = np.fromfile('/tmp/ref_%s.bin' % c, dtype=np.float32); b = np.fromfile('/tmp/pair_%s.bin' % c, dtype=np.float32)
print('   max |pair - ref| = %.3e of max |ref| = %.3e' % (np.abs(a - b).max(), np.abs(a).max()))
PY
  done
done
done
for chain in 0 1; do
  echo "== whole chain $chain: current geometry" >> $out
  CHAIN_DUMP=/tmp/ref_$chain.bin timeout -k 10 120 build/chain_bench 8064 $chain all >> $out 2>&1 || exit 1
  for b in chain_bench chain_bench_early chain_bench_tp2; do
    echo "== whole chain $chain: band 9 as pairs ($b)" >> $out
    CHAIN_PAIR=1 CHAIN_DUMP=/tmp/pair_$chain.bin timeout -k 10 120 build/$b 8064 $chain all >> $out 2>&1 || exit 1
    python3 - $chain >> $out <<'PY'
import sys, numpy as np
c = sys.argv[1]
a = np.fromfile('/tmp/ref_%s.bin' % c, dtype=np.float32); b = np.fromfile('/tmp/pair_%s.bin' % c, dtype=np.float32)
print('   max |pair - ref| = %.3e of max |ref| = %.3e' % (np.abs(a - b).max(), np.abs(a).max()))
PY
  done
done
grep -E "^==|max \||guard|blocks" $out
