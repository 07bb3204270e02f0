cd $GRAFT_REPO_ROOT
tag=${1:-r04a}
if [ "$2" != "nosuite" ]; then timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_suite.log 2>&1; tail -3 gpurun_out/${tag}_suite.log; fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "profile round"; timeout -k 10 900 bash tools/profile_round.sh $tag 2>&1 | tail -2
unset BSRNN_OVERLAP
echo "bench"; timeout -k 10 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; python -c "
import json;d=json.load(open('gpurun_out/${tag}_bench.json'));print(d['ms_per_step'],d['value'],d['roofline']['frac'],d['roofline'].get('kernels_alone',{}).get('frac'),d['roofline_dual_path']['ms_per_step'])"
echo "configs"; (timeout -k 10 200 python tools/bench_configs.py; BSRNN_GEMM=bf16 timeout -k 10 100 python tools/bench_configs.py; BSRNN_GEMM=fp16 timeout -k 10 100 python tools/bench_configs.py) > gpurun_out/${tag}_other_configs.txt 2>&1; tail -12 gpurun_out/${tag}_other_configs.txt
(echo "== plain launches (the default)"; timeout -k 10 100 ./build/stream_cloop 2 2000; echo "== BSRNN_STREAM_GRAPH=1 (one hipGraph per parity)"; BSRNN_STREAM_GRAPH=1 timeout -k 10 100 ./build/stream_cloop 2 2000) > gpurun_out/${tag}_stream_cloop.txt 2>&1; cat gpurun_out/${tag}_stream_cloop.txt
echo "timeline"; timeout -k 10 200 bash tools/overlap_timeline.sh ${tag}tl > /dev/null 2>&1; head -14 gpurun_out/${tag}tl_timeline.txt
