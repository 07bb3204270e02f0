cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_train.py tests/test_gpu_entrypoints.py -x -q -k "overlap or two_clip or graphed or ladspa" > gpurun_out/r04_batch3_tests.log 2>&1; tail -4 gpurun_out/r04_batch3_tests.log
PROBE_TERSE=1 timeout -k 10 300 python tools/overlap_probe.py 48 101 1500 1 2>&1 | tail -2
PROBE_TERSE=1 PROBE_LOAD=0 timeout -k 10 300 python tools/overlap_probe.py 64 126 1500 1 2>&1 | tail -2
bash tools/r04_batch2.sh r04a nosuite
