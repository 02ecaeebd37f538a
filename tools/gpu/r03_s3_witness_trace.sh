#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03s3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/wt -o wt -- python3 $R/tools/lhs_witness_profile.py 20 > $O/wt.log 2>&1
cd $R
python3 tools/per_launch.py $(find $O/wt -name "*kernel_trace.csv") "dw::k_plan" "k_pw_rootinv" "k_pw_prefix" "k_merge_sum" "k_pw_apply" "k_lhs_gather" > $O/per_launch.txt
find $O -name "*.csv" -size +3M -delete
cat $O/wt.log | tail -2; cat $O/per_launch.txt
