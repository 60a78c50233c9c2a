#!/bin/bash
# kernel trace of N in-process ranks on the one GPU: launches per rebuild interval and rank (tools/trace_ranks.sh [L])
R=$GRAFT_REPO_ROOT; L=${1:-64}
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
rm -rf $R/gpurun_out/kt_ranks
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_ranks -o x --output-format csv -- python3 $R/tools/local_ranks_bench.py 8 2 2 2 $L 100 > $R/gpurun_out/kt_ranks.log 2>&1
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/kt_ranks/**/*kernel_stats.csv", recursive=True)[0]
tot = 0
rows = list(csv.DictReader(open(f)))
with open("gpurun_out/kt_ranks_stats.txt", "w") as o:
    for r in rows:
        o.write("%-70s calls %7s avg_us %9.2f total_ms %9.2f pct %6s\n" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"][:6]))
PY
