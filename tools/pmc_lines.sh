#!/bin/bash
# tools/pmc_lines.sh TAG <bench args...>: L1 (TCP) line-request counters of the force kernel launched alone -> gpurun_out/pmcl_TAG.txt
# (how many cache-line accesses the texture path makes per gather instruction: the unit the kernel is bound by, profiles/r06_notes.md 2, 9)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum" "TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum" "SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 100 rocprofv3 --pmc $grp -d $R/gpurun_out/pmcl_$tag/p$i -o x --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --profile-steps 5 --no-cpu-baseline --other-boxes "" --opt fuse_pair=0 "$@" > $R/gpurun_out/pmcl_$tag.p$i.log 2>&1 || echo "pass $i failed: $grp"
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcl_$tag pair_dpd > $R/gpurun_out/pmcl_$tag.txt
cat $R/gpurun_out/pmcl_$tag.txt
