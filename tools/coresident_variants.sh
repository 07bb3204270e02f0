#!/bin/bash
# One GPU run for the co-residency hazard of the SLP-vectorised STFT / iSTFT kernels (NOTEBOOK.md R3.4): the victim process
# loads a variant build of the library (BSRNN_HIP_LIB), the load process the product build.
#   bash tools/coresident_variants.sh build/libbsrnn_slp.so build/libbsrnn_slp0.so ...
export PYTHONPATH=.
python tests/coresident_check.py load 400000 > gpurun_out/co_load.txt 2>&1 &
PL=$!
sleep 10
for lib in "$@"; do
  echo "== victim library: $lib"
  for k in stft istft; do BSRNN_HIP_LIB=$lib timeout -k 5 120 python tests/coresident_check.py $k 20 2>&1 | grep "differing\|first result"; done
done
echo "load alive: $(kill -0 $PL 2>/dev/null && echo yes || echo no)"
kill $PL 2>/dev/null; wait $PL 2>/dev/null
exit 0
