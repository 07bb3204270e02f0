cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_overlap.py -x -q > gpurun_out/r04_batch5_tests.log 2>&1; tail -3 gpurun_out/r04_batch5_tests.log
BSRNN_OVL_GATE=kernel timeout -k 10 400 python -m pytest tests/test_gpu_overlap.py -x -q > gpurun_out/r04_batch5_tests_k.log 2>&1; tail -3 gpurun_out/r04_batch5_tests_k.log
PROBE_TERSE=1 timeout -k 10 300 python tools/overlap_probe.py 48 101 1500 1 2>&1 | tail -1
timeout -k 10 600 bash tools/ab_modes.sh 3 BSRNN_OVERLAP=0 BSRNN_OVL_GATE=cp BSRNN_OVL_GATE=kernel > gpurun_out/r04_ovl_ab4.txt 2>&1; cat gpurun_out/r04_ovl_ab4.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 bash tools/overlap_timeline.sh r04d > /dev/null 2>&1; sed -n 12,27p gpurun_out/r04d_timeline.txt
timeout -k 10 200 bash tools/overlap_timeline.sh r04dk BSRNN_OVL_GATE=kernel > /dev/null 2>&1; sed -n 14,27p gpurun_out/r04dk_timeline.txt
