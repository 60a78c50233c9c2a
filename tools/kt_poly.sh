R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt_poly
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_poly -o x --output-format csv -- python3 $R/bench.py --box 64 --polymer 0.1 --steps 300 --warmup 50 --no-cpu-baseline > $R/gpurun_out/kt_poly.log 2>&1
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/kt_poly/**/*kernel_stats.csv", recursive=True)[0]
with open("gpurun_out/kt_poly_stats.txt", "w") as o:
    for r in csv.DictReader(open(f)):
        o.write("%-90s calls %6s avg_us %9.2f total_ms %9.2f pct %6s\n" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"][:6]))
PY
