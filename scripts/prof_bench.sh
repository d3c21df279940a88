#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py; copies the kernel-stats CSV to gpurun_out/<tag>_kernel_stats.csv
# usage: scripts/prof_bench.sh <tag> [bench args...]     (env vars pass through)
TAG=$1; shift
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_$TAG
timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o p --output-format csv -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $ROOT/gpurun_out/${TAG}_bench.json 2> $ROOT/gpurun_out/${TAG}_bench.err
f=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $ROOT/gpurun_out/${TAG}_kernel_stats.csv
t=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)
[ -n "$t" ] && python3 $ROOT/scripts/trace_gaps.py "$t" > $ROOT/gpurun_out/${TAG}_last_step.txt 2>&1
cd $ROOT
head -c 400 gpurun_out/${TAG}_bench.json; echo
head -25 gpurun_out/${TAG}_last_step.txt
