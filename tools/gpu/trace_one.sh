#!/bin/bash
# usage: bash tools/gpu/trace_one.sh <outdir-name> <opener-regex> [bench.py args...]: one-step kernel timeline of a bench command
set -e
NAME=$1; FIRST=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$NAME; mkdir -p $O
rocprofv3 --kernel-trace -d $O/t -o t --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $O/t.log 2>&1
cd $R
python3 tools/trace_timeline.py --first "$FIRST" $(find $O/t -name "*kernel_trace.csv") > $O/timeline.txt
find $O -name "*.csv" -size +3M -delete
cat $O/timeline.txt
