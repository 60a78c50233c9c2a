#!/bin/bash
# kernel trace of a small box: per-kernel durations and the gaps between consecutive kernels inside one rebuild interval
# usage: tools/trace_small.sh [box] [extra bench args...]
R=$GRAFT_REPO_ROOT; box=${1:-32}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt_small
timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/kt_small -o x --output-format csv -- python3 $R/bench.py --box $box --steps 200 --warmup 50 --profile-steps 10 --no-cpu-baseline "$@" > $R/gpurun_out/kt_small.log 2>&1
cd $R
python3 tools/trace_window.py gpurun_out/kt_small
