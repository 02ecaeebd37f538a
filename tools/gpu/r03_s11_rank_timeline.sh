#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03s11; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/rk -o rk -- python3 $R/tools/rank_profile.py 24 2 4 > $O/rk.log 2>&1
cd $R
python3 tools/trace_timeline.py $(find $O/rk -name "*kernel_trace.csv") > $O/rank_2_4.timeline.txt
find $O -name "*.csv" -size +3M -delete
cat $O/rank_2_4.timeline.txt
