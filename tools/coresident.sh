#!/bin/bash
# Determinism / correctness of the entry points while a second process keeps the GPU busy, then of concurrent row blocks
# (run on the GPU box: bash tools/coresident.sh).  Never overwrite the .so while a process has it mapped.
# To reproduce the hazard this guards against, rebuild with the SLP vectorizer first:
#   touch speechseparation_amd/csrc/fft.hip && make -C speechseparation_amd/csrc EXTRA=-fslp-vectorize
# (stft / istft / separate then differ from the quiet result in 15-20 of 20 runs, whole frames of garbage; forward stays clean).
export PYTHONPATH=.
python tests/coresident_check.py load 400000 > gpurun_out/co_load.txt 2>&1 &
PL=$!
sleep 8
for k in stft istft separate forward; do python tests/coresident_check.py $k 20 2>&1 | grep "differing\|first result"; done
echo "load alive: $(kill -0 $PL 2>/dev/null && echo yes || echo no)"
kill $PL 2>/dev/null; wait $PL 2>/dev/null
for cfg in "2 0" "2 3" "3 0"; do set -- $cfg; echo -n "row blocks: parts $1 lag $2: "; BSRNN_PARTS=$1 BSRNN_PART_LAG=$2 python tools/row_block_check.py 30 2>&1 | grep "dirty runs"; done
