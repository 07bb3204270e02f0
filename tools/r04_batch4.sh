cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_parity.py -x -q > gpurun_out/r04_batch4_tests.log 2>&1; tail -3 gpurun_out/r04_batch4_tests.log
PROBE_TERSE=1 timeout -k 10 300 python tools/overlap_probe.py 48 101 1500 1 2>&1 | tail -1
PROBE_TERSE=1 PROBE_LOAD=0 timeout -k 10 300 python tools/overlap_probe.py 64 126 1000 1 2>&1 | tail -1
timeout -k 10 500 bash tools/ab_modes.sh 3 BSRNN_OVERLAP=0 BSRNN_OVERLAP=1 > gpurun_out/r04_ovl_ab3.txt 2>&1; cat gpurun_out/r04_ovl_ab3.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 bash tools/overlap_timeline.sh r04c > /dev/null 2>&1; sed -n 14,27p gpurun_out/r04c_timeline.txt
