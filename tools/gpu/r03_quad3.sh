#!/bin/bash
set -e
bash tools/gpu/bench_trio.sh r03z2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03z2
rocprofv3 --kernel-trace -d $O/t20q2 -o t20q2 --output-format csv -- python3 $R/bench.py --logn 20 --steps 6 --warmup 2 --no-cpu-baseline --option pyr_quad=2 > $O/t20q2.log 2>&1
cd $R
python3 tools/trace_timeline.py $(find $O/t20q2 -name "*kernel_trace.csv") > $O/t20q2.timeline.txt
find $O -name "*.csv" -size +3M -delete
echo "== quad"; grep -E "k_pyramid|step:" $O/t20.timeline.txt
echo "== one lane per addition"; grep -E "k_pyramid|step:" $O/t20q2.timeline.txt
