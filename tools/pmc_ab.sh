#!/bin/bash
# tools/pmc_ab.sh OUT LIB [bench args]: issue / texture / LDS counters of the force kernel under one in-tree build (LIB '-' = default)
out=$1; lib=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ $lib != - ]; then export MESO_LIB=$R/meso_amd/libmeso_hip_$lib.so; fi
mkdir -p $R/gpurun_out/$out
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout -k 10 100 rocprofv3 --pmc $grp -d $R/gpurun_out/$out/p$i -o x --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --profile-steps 5 --no-cpu-baseline --other-boxes "" "$@" > $R/gpurun_out/$out/p$i.log 2>&1 || echo "pass $i failed: $grp"
done <<GRP
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD
SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAVES
TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum
TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum
GRP
python3 $R/tools/pmc_summary.py $R/gpurun_out/$out pair_dpd_ring > $R/gpurun_out/$out/summary.txt
cat $R/gpurun_out/$out/summary.txt
