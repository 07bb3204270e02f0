#!/bin/bash
# Collect the per-round rocprofv3 evidence on the GPU box: kernel trace + stats, then PMC passes, each in its own run
# (counters are never combined with other trace domains).  Usage: bash tools/profile_round.sh r01d
set -e
tag=$1
export TMPDIR=/tmp
B="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact-f32 --no-train-step"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- $B > gpurun_out/prof_$tag.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_fetch -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${tag}_write -- $B > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_${tag}_tcc -- $B > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --output-format csv -d gpurun_out/pmc_${tag}_sq -- $B > /dev/null 2>&1
echo profiled $tag
