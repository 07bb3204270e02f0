#!/bin/bash
# Collect the per-round rocprofv3 evidence on the GPU box: kernel trace + stats, then PMC passes, each in its own run
# (counters are never combined with other trace domains).  Usage: bash tools/profile_round.sh r01d
set -e
tag=$1
export TMPDIR=/tmp
B="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact-f32 --no-train-step --no-in-flight"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- $B > gpurun_out/prof_$tag.log 2>&1
# counter passes serialise the kernels: the overlapped dual path (a consumer launch BESIDE its producer) cannot exist under them and would
# only fall back after its bounded waits; the passes run the launches one after the other from the start
export BSRNN_OVERLAP=0
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_fetch -- $B > gpurun_out/pmc_${tag}.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${tag}_write -- $B > gpurun_out/pmc_${tag}.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_${tag}_tcc -- $B > gpurun_out/pmc_${tag}.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --output-format csv -d gpurun_out/pmc_${tag}_sq -- $B > gpurun_out/pmc_${tag}.log 2>&1
# what bounds the fused chains (VERDICT r3 #2c): requests from the CUs' L1 into L2, L1 stalls on L2, LDS conflicts / stalls, wave cycles
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d gpurun_out/pmc_${tag}_l1l2 -- $B > gpurun_out/pmc_${tag}.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_${tag}_lds -- $B > gpurun_out/pmc_${tag}.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d gpurun_out/pmc_${tag}_wave -- $B > gpurun_out/pmc_${tag}.log 2>&1
echo profiled $tag
