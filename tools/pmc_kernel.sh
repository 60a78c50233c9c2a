#!/bin/bash
# tools/pmc_kernel.sh TAG KERNEL-SUBSTRING <bench args...>: issue / LDS / occupancy counters of one kernel -> gpurun_out/pmck_TAG.txt
tag=$1; kern=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAVES" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout -k 10 100 rocprofv3 --pmc $grp -d $R/gpurun_out/pmck_$tag/p$i -o x --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --profile-steps 5 --no-cpu-baseline "$@" > $R/gpurun_out/pmck_$tag.p$i.log 2>&1 || echo "pass $i failed"
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmck_$tag $kern > $R/gpurun_out/pmck_$tag.txt
cat $R/gpurun_out/pmck_$tag.txt
