#!/bin/bash
# Collect the judged artefacts of one round on the GPU box (run through gpurun), then tools/update_profiles.py <tag> here:
#   tools/collect_profiles.sh r02   -> gpurun_out/r02_bench64_fast.json, r02_kernel_stats_64_fast.txt, r02_pmc_64_fast.txt,
#                                      r02_pmc_pair_only.txt (force kernel launched alone: FETCH_SIZE / WRITE_SIZE), other configs
set -e
tag=$1
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 400 python3 bench.py > gpurun_out/${tag}_bench64_fast.json 2> gpurun_out/${tag}_bench64_fast.err
echo "bench done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_kt -o x --output-format csv -- python3 $R/bench.py --no-cpu-baseline --other-boxes "" > $R/gpurun_out/${tag}_kt.log 2>&1
echo "kernel trace done"
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/${tag}_kt/**/*kernel_stats.csv", recursive=True)[0]
with open("gpurun_out/${tag}_kernel_stats_64_fast.txt", "w") as o:
    o.write("rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline   (64^3 rho=4, dpd/fast/meso, 200+1000+200(+100 force-only) steps)\n")
    for r in csv.DictReader(open(f)):
        o.write("%-90s calls %6s avg_us %9.2f total_ms %9.2f pct %6s\n" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"][:6]))
PY
# the same trace with the step boundary in its own kernel: k_pair_dpd_ring's average is then the force kernel ALONE (the graded figure)
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_ktp -o x --output-format csv -- python3 $R/bench.py --no-cpu-baseline --other-boxes "" --steps 300 --warmup 50 --opt fuse_pair=0 > $R/gpurun_out/${tag}_ktp.log 2>&1
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/${tag}_ktp/**/*kernel_stats.csv", recursive=True)[0]
with open("gpurun_out/${tag}_kernel_stats_64_pair_only.txt", "w") as o:
    o.write("rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 300 --warmup 50 --opt fuse_pair=0   (64^3 rho=4, dpd/fast/meso; the force kernel is launched alone on every step)\n")
    for r in csv.DictReader(open(f)):
        o.write("%-90s calls %6s avg_us %9.2f total_ms %9.2f pct %6s\n" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"][:6]))
PY
echo "pair-only kernel trace done"
bash tools/pmc_run.sh ${tag}_pmc --other-boxes ""
( echo "# rocprofv3 --pmc <counters> -- python3 bench.py --steps 20 --warmup 5 --profile-steps 5 --no-cpu-baseline   (64^3 rho=4, dpd/fast/meso; one pass per counter group; mean per dispatch)"; cat gpurun_out/${tag}_pmc.summary.txt ) > gpurun_out/${tag}_pmc_64_fast.txt
echo "pmc done"
# the force kernel launched alone (step boundary in its own kernel): HBM traffic of the graded kernel
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --pmc $c -d $R/gpurun_out/${tag}_pmcp/$c -o x --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --profile-steps 5 --no-cpu-baseline --other-boxes "" --opt fuse_pair=0 > $R/gpurun_out/${tag}_pmcp.$c.log 2>&1
done
cd $R
( echo "# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 20 --warmup 5 --profile-steps 5 --no-cpu-baseline --opt fuse_pair=0   (force kernel alone; mean per dispatch, KB)"; python3 tools/pmc_summary.py gpurun_out/${tag}_pmcp pair_dpd merge_xvt nve ) > gpurun_out/${tag}_pmc_pair_only.txt
echo "pair-only pmc done"
# issue / texture-path counters of the force kernel launched alone (the limiter block of the bench line)
bash tools/pmc_focus.sh ${tag}_focus --other-boxes "" --opt fuse_pair=0 > gpurun_out/${tag}_focus.log 2>&1 || echo "focus counters failed"
# ... and of the fp64 style (dpd/meso: configs[1] and configs[3])
bash tools/pmc_focus.sh ${tag}_focus_dp --other-boxes "" --style dpd/meso --opt fuse_pair=0 > gpurun_out/${tag}_focus_dp.log 2>&1 || echo "fp64 focus counters failed"
bash tools/pmc_kernel.sh ${tag}_tb tile_build --other-boxes "" > gpurun_out/${tag}_tb.log 2>&1 || echo "list builder counters failed"
echo "focus pmc done"
# the other configurations of BASELINE.json (parity-test cases; timed for the record)
timeout -k 10 300 python3 bench.py --box 25 --no-cpu-baseline > gpurun_out/${tag}_bench25_fast.json 2>/dev/null
timeout -k 10 300 python3 bench.py --box 32 --no-cpu-baseline > gpurun_out/${tag}_bench32_fast.json 2>/dev/null
timeout -k 10 300 python3 bench.py --box 48 --no-cpu-baseline > gpurun_out/${tag}_bench48_fast.json 2>/dev/null
timeout -k 10 300 python3 bench.py --box 64 --style dpd/meso --no-cpu-baseline > gpurun_out/${tag}_bench64_dp.json 2>/dev/null
timeout -k 10 300 python3 bench.py --box 25 --style dpd/meso --every 1 --no-cpu-baseline > gpurun_out/${tag}_bench25_dp_every1.json 2>/dev/null
timeout -k 10 300 python3 bench.py --box 128 --steps 300 --warmup 50 --no-cpu-baseline > gpurun_out/${tag}_bench128_fast.json 2>/dev/null
timeout -k 10 300 python3 bench.py --box 64 --polymer 0.1 --no-cpu-baseline > gpurun_out/${tag}_bench64_polymer.json 2>/dev/null
timeout -k 10 400 python3 bench.py --box 128 --polymer 0.1 --steps 200 --warmup 50 --no-cpu-baseline > gpurun_out/${tag}_bench128_polymer.json 2>/dev/null
echo "other configs done"
