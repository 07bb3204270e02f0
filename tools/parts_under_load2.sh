#!/bin/bash
# run-to-run stability of separate() / forward() beside a second process, per flow (400 runs each)
export PYTHONPATH=$PWD
python tests/coresident_check.py load 4000000 > gpurun_out/pul_load.txt 2>/dev/null &
LOAD=$!
for i in $(seq 1 60); do grep -q "load ready" gpurun_out/pul_load.txt 2>/dev/null && break; sleep 1; done
echo "load alive: $(kill -0 $LOAD 2>/dev/null && echo yes || echo no)"
for v in "BSRNN_BAND_FC=part" "BSRNN_BAND_FC=part BSRNN_BAND_GRID=2d" "BSRNN_BAND_FC=gemm" "BSRNN_BAND_FC=gemm BSRNN_BAND_GRID=2d"; do
  echo "== $v"
  env $v timeout -k 10 200 python tests/coresident_check.py separate 400 2>/dev/null | tail -8
done
kill $LOAD 2>/dev/null; wait $LOAD 2>/dev/null
echo done
