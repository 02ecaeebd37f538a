#!/bin/bash
# Round 3, second session: vector loads of the negabase digits in pass 1, call-start clears as kernels, deferred range-check read-back.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03s5; mkdir -p $O
cd $R
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lhs or golden or msm_matches_oracle or share_one_tail or sharded_sim" > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
AB_WORKLOAD=lhs AB_ROUNDS=9 timeout -k 10 200 python3 tools/ab_bench.py 20 "" "neg_vec=2" "own_clear=2" "neg_vec=2,own_clear=2" > $O/ab_lhs20.txt 2>&1
AB_ROUNDS=9 timeout -k 10 200 python3 tools/ab_bench.py 20 "" "own_clear=2" > $O/ab_msm20.txt 2>&1
AB_ROUNDS=7 timeout -k 10 200 python3 tools/ab_bench.py 24 "" "own_clear=2" > $O/ab_msm24.txt 2>&1
cd /tmp && export TMPDIR=/tmp
AB_WORKLOAD=lhs AB_ROUNDS=5 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ksl -o ks -- python3 $R/tools/ab_bench.py 20 "" "neg_vec=2,own_clear=2" > $O/ksl.log 2>&1
AB_ROUNDS=5 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ksm -o ks -- python3 $R/tools/ab_bench.py 20 "" "own_clear=2" > $O/ksm.log 2>&1
cd $R
python3 tools/trace_timeline.py --first "k_negbase_digits" --step -3 $(find $O/ksl -name "*kernel_trace.csv") > $O/ksl.timeline_a.txt || true
python3 tools/trace_timeline.py --first "k_negbase_digits" --step -2 $(find $O/ksl -name "*kernel_trace.csv") > $O/ksl.timeline_b.txt || true
python3 tools/trace_timeline.py --step -3 $(find $O/ksm -name "*kernel_trace.csv") > $O/ksm.timeline_a.txt || true
python3 tools/trace_timeline.py --step -2 $(find $O/ksm -name "*kernel_trace.csv") > $O/ksm.timeline_b.txt || true
cp $(find $O/ksl -name "*kernel_stats.csv") $O/ksl.kernel_stats.csv
find $O -name "*.csv" -size +3M -delete
tail -2 $O/tests.txt; cat $O/ab_lhs20.txt $O/ab_msm20.txt $O/ab_msm24.txt; head -12 $O/ksl.timeline_a.txt; head -12 $O/ksl.timeline_b.txt
