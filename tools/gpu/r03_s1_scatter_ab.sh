#!/bin/bash
# Round 3, second session: lean k_scatter1 and the bucket clear beside the sort -- targeted parity tests, interleaved A/B
# in one process, kernel statistics of both forms of the kernel.  Output: gpurun_out/r03s1/
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03s1; mkdir -p $O
cd $R
timeout -k 10 420 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scatter_form or window_and_chunk or msm_matches_oracle or skew or golden or ragged_sizes or multi_slab or share_one_tail" > $O/tests.txt 2>&1
AB_ROUNDS=9 timeout -k 10 200 python3 tools/ab_bench.py 20 "" "scatter_lean=2" "clear_beside=2" "scatter_lean=2,clear_beside=2" > $O/ab20.txt 2>&1
AB_ROUNDS=7 timeout -k 10 200 python3 tools/ab_bench.py 24 "" "scatter_lean=2" > $O/ab24.txt 2>&1
AB_ROUNDS=7 timeout -k 10 200 python3 tools/ab_bench.py 22 "" "scatter_lean=2" "clear_beside=2" > $O/ab22.txt 2>&1
cd /tmp && export TMPDIR=/tmp
AB_ROUNDS=5 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks24 -o ks -- python3 $R/tools/ab_bench.py 24 "" "scatter_lean=2" > $O/ks24.log 2>&1
AB_ROUNDS=5 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks20 -o ks -- python3 $R/tools/ab_bench.py 20 "" "scatter_lean=2,clear_beside=2" > $O/ks20.log 2>&1
cd $R
for t in ks24 ks20; do cp $(find $O/$t -name "*kernel_stats.csv") $O/$t.kernel_stats.csv; python3 tools/trace_timeline.py --step -3 $(find $O/$t -name "*kernel_trace.csv") > $O/$t.timeline_default.txt; python3 tools/trace_timeline.py --step -2 $(find $O/$t -name "*kernel_trace.csv") > $O/$t.timeline_knobs.txt; done
find $O -name "*.csv" -size +3M -delete
tail -2 $O/tests.txt; cat $O/ab20.txt $O/ab22.txt $O/ab24.txt; grep -h scatter1 $O/ks24.kernel_stats.csv $O/ks20.kernel_stats.csv
