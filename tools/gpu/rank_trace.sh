#!/bin/bash
# one rank of 8 of the window-sharded 2^24 MSM under rocprofv3 --kernel-trace: timeline of its last call
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
rocprofv3 --kernel-trace -d $O/t -o t --output-format csv -- python3 $R/tools/rank_profile.py 24 2 4 > $O/t.log 2>&1
cd $R
python3 tools/trace_timeline.py --first "k_pip_digits" $(find $O/t -name "*kernel_trace.csv") > $O/timeline.txt
find $O -name "*.csv" -size +3M -delete
cat $O/timeline.txt
