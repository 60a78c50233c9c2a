#!/bin/bash
# (TA: two counters per pass, more 'exceeds the capabilities of the hardware'; a GRBM_* group aborts inside rocprofv3 on this pool - signal 6 - and is left out)
# usage: tools/pmc_focus.sh <outdir-under-gpurun_out> <bench args...>: clock, issue and texture-path counters of the force kernel
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout -k 10 100 rocprofv3 --pmc $grp -d $R/gpurun_out/$out/p$i -o x --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --profile-steps 5 --no-cpu-baseline "$@" > $R/gpurun_out/$out.p$i.log 2>&1
  echo "pass $i done: $grp"
done <<GRP
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD
TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
GRP
python3 $R/tools/pmc_summary.py $R/gpurun_out/$out pair_dpd > $R/gpurun_out/$out.summary.txt
