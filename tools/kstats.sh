#!/bin/bash
# tools/kstats.sh TAG <bench args...>: rocprofv3 kernel-trace statistics of one bench.py invocation -> gpurun_out/kstats_TAG.txt
# (per kernel: calls, average, share; plus LDS bytes / workgroup size of the first dispatch of each kernel)
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kstats_$tag -o x --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 200 --warmup 50 --profile-steps 20 "$@" > $R/gpurun_out/kstats_$tag.log 2>&1 || { echo failed; tail -5 $R/gpurun_out/kstats_$tag.log; exit 1; }
cd $R
python3 - <<PY
import csv, glob
lds = {}
for f in glob.glob("gpurun_out/kstats_$tag/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        lds.setdefault(r["Kernel_Name"], (r.get("LDS_Block_Size"), r.get("Workgroup_Size"), r.get("Grid_Size"), r.get("VGPR_Count"), r.get("Scratch_Size")))
f = glob.glob("gpurun_out/kstats_$tag/**/*kernel_stats.csv", recursive=True)[0]
with open("gpurun_out/kstats_$tag.txt", "w") as o:
    for r in csv.DictReader(open(f)):
        l = lds.get(r["Name"], ("?",) * 5)
        o.write("%-80s calls %6s avg_us %9.2f pct %6s  lds %s wg %s grid %s vgpr %s scratch %s\n" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"][:6], *l))
print(open("gpurun_out/kstats_$tag.txt").read())
PY
