#!/bin/bash
# kernel trace of a small box: per-kernel durations and the gaps between consecutive kernels inside one rebuild interval
R=$GRAFT_REPO_ROOT; box=${1:-32}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt_small
timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/kt_small -o x --output-format csv -- python3 $R/bench.py --box $box --steps 200 --warmup 50 --profile-steps 10 --no-cpu-baseline > $R/gpurun_out/kt_small.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/kt_small/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the timed region: take a window of 10 steps in the middle
names = [r["Kernel_Name"].split("(")[0].replace("void meso::", "")[:48] for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("k_tile_build")]
a, b = idx[len(idx)//2], idx[len(idx)//2 + 1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = None
tot_busy = 0
for i in range(a, b):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%8.1f us  +gap %5.1f  dur %6.1f  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, names[i]))
    prev_end = e; tot_busy += e - s
print("interval %.1f us, busy %.1f us, kernels %d" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, tot_busy / 1e3, b - a))
PY
