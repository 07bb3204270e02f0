set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_entrypoints.py tests/test_gpu_modes.py tests/test_gpu_overlap.py -x -q -k "ladspa or 16bit or overlap" > gpurun_out/r04_batch1_tests.log 2>&1; tail -5 gpurun_out/r04_batch1_tests.log
timeout -k 10 300 python -m pytest tests/test_named_sizes.py tests/test_gpu_fullsize.py -x -q -k "config2" -s > gpurun_out/r04_batch1_cfg2.log 2>&1; grep -E "passed|failed|range" gpurun_out/r04_batch1_cfg2.log | tail -12
# SIGSEGV hunt: the C loop under the profiler, graph and no graph
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_cloop_g -- ./build/stream_cloop 2 400 > gpurun_out/r04_cloop_prof_graph.log 2>&1; echo "graph rc=$?" ) 2>&1 | tail -1
(BSRNN_NO_GRAPH=1 timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_cloop_n -- ./build/stream_cloop 2 400 > gpurun_out/r04_cloop_prof_nograph.log 2>&1; echo "nograph rc=$?") 2>&1 | tail -1
(timeout -k 10 120 ./build/stream_cloop 2 2000 > gpurun_out/r04_cloop_plain.log 2>&1; echo "plain rc=$?") 2>&1 | tail -1
tail -4 gpurun_out/r04_cloop_prof_graph.log; tail -4 gpurun_out/r04_cloop_prof_nograph.log; cat gpurun_out/r04_cloop_plain.log
