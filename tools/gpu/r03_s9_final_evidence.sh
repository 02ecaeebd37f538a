#!/bin/bash
# Round 3, end of the second session: bench lines of every BASELINE config that fits one GPU, kernel statistics and one-step
# timelines of the same commands on the final code.  Output: gpurun_out/r03s9/ (copied into profiles/r03/ as zz_* afterwards).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03s9; mkdir -p $O
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_msm_2p24.json 2> $O/err.txt
python3 bench.py --logn 20 --steps 30 --warmup 5 > $O/bench_msm_2p20.json 2>> $O/err.txt
python3 bench.py --workload lhs --logn 20 --steps 20 --warmup 5 > $O/bench_lhs_2p20.json 2>> $O/err.txt
python3 bench.py --curve grumpkin --logn 22 --steps 10 --warmup 3 > $O/bench_grumpkin_2p22.json 2>> $O/err.txt
python3 bench.py --workload lhs_witness --logn 20 --steps 3 --warmup 1 --cpu-sample-log 14 > $O/bench_lhs_witness_2p20.json 2>> $O/err.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks24 -o ks -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/ks24.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks20 -o ks -- python3 $R/bench.py --logn 20 --steps 12 --warmup 2 --no-cpu-baseline > $O/ks20.log 2>&1
cd $R
for t in ks24 ks20; do python3 tools/trace_timeline.py $(find $O/$t -name "*kernel_trace.csv") > $O/$t.timeline.txt; cp $(find $O/$t -name "*kernel_stats.csv") $O/$t.kernel_stats.csv; done
python3 tools/sharded_sim_timing.py 24 1 8 > $O/sharded_sim_timing.txt 2>&1 || true
find $O -name "*.csv" -size +3M -delete
for f in bench_msm_2p24 bench_msm_2p20 bench_lhs_2p20 bench_grumpkin_2p22 bench_lhs_witness_2p20; do python3 -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], 'ms/step', '%.4g' % d['value'], d['unit'], 'bit_exact', d['config'].get('bit_exact'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))"; done
cat $O/sharded_sim_timing.txt
