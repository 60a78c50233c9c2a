#!/usr/bin/env python3
"""Per-kernel durations and gaps inside one rebuild interval of a rocprofv3 --kernel-trace CSV (tools/trace_small.sh)."""
import csv, glob, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/kt_small"
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("meso::", "").replace("rocprim::ROCPRIM_400200_NS::detail::", "rp::")[:48] for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("k_tile_build")]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = None
tot_busy = 0
for i in range(a, b):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%8.1f us  +gap %5.1f  dur %6.1f  %-48s grid %s wg %s lds %s vgpr %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, names[i], rows[i]["Grid_Size_X"], rows[i]["Workgroup_Size_X"], rows[i]["LDS_Block_Size"], rows[i]["VGPR_Count"]))
    prev_end = e; tot_busy += e - s
print("interval %.1f us, busy %.1f us, kernels %d" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, tot_busy / 1e3, b - a))
