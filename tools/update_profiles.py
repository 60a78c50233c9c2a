#!/usr/bin/env python3
"""Copy the artefacts of tools/collect_profiles.sh from gpurun_out/ into profiles/ and rebuild profiles/<tag>_traffic.json
(HBM bytes of the dominant kernel from the FETCH_SIZE / WRITE_SIZE passes).   usage: tools/update_profiles.py r01"""
import json, re, shutil, sys
tag = sys.argv[1]
import os
for f in ("bench64_fast.json", "kernel_stats_64_fast.txt", "kernel_stats_64_pair_only.txt", "pmc_64_fast.txt", "pmc_pair_only.txt"):
    shutil.copy("gpurun_out/%s_%s" % (tag, f), "profiles/%s_%s" % (tag, f))
for f in ("bench25_fast.json", "bench32_fast.json", "bench48_fast.json", "bench64_dp.json", "bench25_dp_every1.json", "bench128_fast.json",
          "bench64_polymer.json", "bench128_polymer.json"):
    src = "gpurun_out/%s_%s" % (tag, f)
    if os.path.exists(src) and os.path.getsize(src) > 100:
        shutil.copy(src, "profiles/%s_%s" % (tag, f))
d = json.loads(open("profiles/%s_bench64_fast.json" % tag).read().strip().splitlines()[-1])
vals = {}
# the graded kernel is the force kernel ALONE: its counters come from the passes with the step boundary in its own kernel
for ln in open("profiles/%s_pmc_pair_only.txt" % tag):
    m = re.match(r"(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+mean\s+([0-9.e+]+)", ln)
    if m and "k_pair_dpd_ring" in m.group(1):
        vals[m.group(2)] = float(m.group(3))
t = {
    "source": "profiles/%s_pmc_pair_only.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, force kernel launched alone, mean per dispatch; tools/collect_profiles.sh)" % tag,
    "box": 64, "style": "dpd/fast/meso",
    "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads (MI355X_MICROARCH.md, HBM): calibrated in the same passes on k_merge_xvt (28.7 MB reported for 58.7 MB read) and k_nve_final (30.7 vs 62.9 MB) -> x2, applied to the whole kernel (an upper bound for its 16-byte gathers); WRITE_SIZE is exact",
    "workload": "64^3 rho=4, dpd/fast/meso",
    "kernel": d["roofline"]["kernel"].split(" (")[0],
    "FETCH_SIZE_KB": vals["FETCH_SIZE"],
    "WRITE_SIZE_KB": vals["WRITE_SIZE"],
    "traffic_bytes_per_launch": int(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024),
    "algorithmic_bytes_per_launch": int(d["roofline"]["bytes_per_launch"]),
}
t["ratio"] = round(t["traffic_bytes_per_launch"] / t["algorithmic_bytes_per_launch"], 3)
json.dump(t, open("profiles/%s_traffic.json" % tag, "w"), indent=1)
print(json.dumps(t, indent=1))
