#!/bin/bash
# tools/pmc_ab.sh "LIBS": issue / LDS / texture counters of the force kernel under several in-tree builds ('-' = default)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in $1; do
  if [ $v = - ]; then unset MESO_LIB; else export MESO_LIB=$R/meso_amd/libmeso_hip_$v.so; fi
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
             "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES" \
             "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
    i=$((i+1))
    timeout -k 10 100 rocprofv3 --pmc $grp -d $R/gpurun_out/pmcab_$v/p$i -o x --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --profile-steps 5 --no-cpu-baseline > $R/gpurun_out/pmcab_$v.p$i.log 2>&1 || echo "pass $i of $v failed"
  done
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcab_$v pair_dpd > $R/gpurun_out/pmcab_$v.summary.txt
  echo "== $v"; cat $R/gpurun_out/pmcab_$v.summary.txt
done
