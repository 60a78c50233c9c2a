#!/usr/bin/env python3
"""tools/trace_short_run.py DIR: the kernels of the LAST 20-step run() in a rocprofv3 --kernel-trace of `bench.py --steps 20 --warmup 5`
(the driver's invocation): what a short run spends outside its steady-state steps - first and last step, gaps."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
nm = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("meso::", "")[:40]
# the timed run: the last window of 20 force launches that is followed by the floor kernels / the end; find by the gap structure: take the
# last 20 k_pair_dpd_ring launches before the first k_pair_floor / k_floor_prepare launch (or the end)
stop = next((i for i, r in enumerate(rows) if "floor" in r["Kernel_Name"]), len(rows))
pr = [i for i in range(stop) if "k_pair_dpd_ring" in rows[i]["Kernel_Name"]]
first = pr[-20]
# walk back to the kernel that starts the run (k_nve_initial in front of the first force launch)
a = first
while a > 0 and ("nve_initial" in rows[a - 1]["Kernel_Name"] or "merge_xvt" in rows[a - 1]["Kernel_Name"]):
    a -= 1
b = pr[-1]
while b + 1 < stop and ("nve_final" in rows[b + 1]["Kernel_Name"]):
    b += 1
t0 = int(rows[a]["Start_Timestamp"]); prev = None; busy = 0
for i in range(a, b + 1):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    if i < a + 8 or i > b - 6 or gap > 3.0:
        print("%9.1f us  +gap %6.1f  dur %6.1f  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, nm(rows[i])))
    prev = e; busy += e - s
print("run: %.1f us wall on the device, %.1f us busy, %d kernels" % ((int(rows[b]["End_Timestamp"]) - t0) / 1e3, busy / 1e3, b - a + 1))
