#!/bin/bash
# tools/profile_window_check.sh (on the GPU box, through gpurun): `-profile interval 5 12` observed under rocprofv3.
# (rocprofv3 acts on roctxProfilerPause / roctxProfilerResume only while the marker domain is traced: --marker-trace.)
# The kernel trace must hold exactly the launches of timesteps 6..12 (collection resumes when ntimestep == 5 and pauses when it is 12,
# engine_meso.cu:155-177): 7 force launches, the one list build of step 10, nothing of setup() and nothing of steps 13..20.
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r4/profile_window
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --marker-trace -d $out -o pw --output-format csv -- python3 $R/tools/profile_window_run.py > $out/run.log 2>&1 || { echo "rocprofv3 run failed"; tail -5 $out/run.log; exit 1; }
python3 - "$out" <<'PY' | tee $out/report.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
assert f, "no kernel trace written"
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("meso::", "") for r in rows]
npair = sum(n.startswith("k_pair_dpd_ring") for n in names)
nbuild = sum(n.startswith("k_tile_build") for n in names)
nev = sum(n.startswith("k_pair_dpd<") for n in names)       # setup()'s energy/virial launch: must be outside the window
print("-profile interval 5 12, 20-step run of a 12^3 box (6912 atoms), rocprofv3 --kernel-trace:")
print("kernels in the trace: %d; force launches %d (expected 7: timesteps 6..12), list builds %d (expected 1: step 10), setup's ev kernel %d (expected 0)" % (len(rows), npair, nbuild, nev))
from collections import Counter
for k, c in sorted(Counter(names).items()):
    print("  %4d  %s" % (c, k[:100]))
assert npair == 7 and nbuild == 1 and nev == 0, "the window does not bracket timesteps 6..12"
print("OK: the trace holds exactly the launches of the window")
PY
