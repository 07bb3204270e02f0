#!/bin/bash
# the stand-alone checks of the parts flow (tools/band_parts_check.hip) while a second process keeps the GPU busy with separate()
export PYTHONPATH=$PWD
python tests/coresident_check.py load 400000 > gpurun_out/pul_load.txt 2>/dev/null &
LOAD=$!
for i in $(seq 1 60); do grep -q "load ready" gpurun_out/pul_load.txt 2>/dev/null && break; sleep 1; done
echo "load alive: $(kill -0 $LOAD 2>/dev/null && echo yes || echo no)"
timeout -k 10 200 build/band_parts_check
echo "== dual_path / forward / separate run to run beside the load"
timeout -k 10 120 python tests/coresident_check.py forward 20 2>/dev/null | tail -4
timeout -k 10 120 python tests/coresident_check.py separate 20 2>/dev/null | tail -4
kill $LOAD 2>/dev/null; wait $LOAD 2>/dev/null
echo done
