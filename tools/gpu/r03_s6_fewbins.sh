#!/bin/bash
# Round 3, second session: replicated LDS cursors for the few-bin (negabase) pass 1 -- tests, bench line, kernel statistics, timeline.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03s6; mkdir -p $O
cd $R
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lhs or golden or negbase" > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
python3 bench.py --workload lhs --logn 20 --steps 20 --warmup 5 > $O/bench_lhs_2p20.json 2> $O/err.txt
python3 bench.py --workload lhs --curve grumpkin --logn 20 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_lhs_grumpkin_2p20.json 2>> $O/err.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ksl -o ks -- python3 $R/bench.py --workload lhs --logn 20 --steps 12 --warmup 2 --no-cpu-baseline > $O/ksl.log 2>&1
cd $R
python3 tools/trace_timeline.py --first "k_negbase_digits" $(find $O/ksl -name "*kernel_trace.csv") > $O/ksl.timeline.txt || true
cp $(find $O/ksl -name "*kernel_stats.csv") $O/ksl.kernel_stats.csv
find $O -name "*.csv" -size +3M -delete
tail -2 $O/tests.txt; python3 -c "
import json
for f in ('bench_lhs_2p20','bench_lhs_grumpkin_2p20'):
    d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); print(f, d['ms_per_step'], 'ms/step', '%.4g' % d['value'], d['config'].get('bit_exact'))"
cat $O/ksl.timeline.txt
