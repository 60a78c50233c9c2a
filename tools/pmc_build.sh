#!/bin/bash
# issue-side counters of the list builder: tools/pmc_build.sh <outdir-under-gpurun_out> <bench args...>
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
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_LDS SQ_LEVEL_WAVES SQ_INSTS_SMEM
SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
GRP
python3 $R/tools/pmc_summary.py $R/gpurun_out/$out tile_build > $R/gpurun_out/$out.summary.txt
