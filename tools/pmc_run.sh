#!/bin/bash
# (a pass mixing TA_* and TCP_*_STALL counters aborted inside rocprofv3 on this pool: keep groups to one block)
# usage: tools/pmc_run.sh <outdir-under-gpurun_out> <bench args...>   (one rocprofv3 --pmc pass per counter group)
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp -d $R/gpurun_out/$out/p$i -o x --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --profile-steps 5 --no-cpu-baseline "$@" > $R/gpurun_out/$out.p$i.log 2>&1
  echo "pass $i done: $grp"
done <<GRP
FETCH_SIZE
WRITE_SIZE
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_LEVEL_WAVES
SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
GRP
python3 $R/tools/pmc_summary.py $R/gpurun_out/$out pair_dpd tile_build cell_build brick nve merge_xvt permute k_fr_ > $R/gpurun_out/$out.summary.txt
