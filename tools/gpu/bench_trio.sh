#!/bin/bash
# usage: bash tools/gpu/bench_trio.sh <outdir-name> [extra bench.py args...]
# bench lines + one-step kernel timelines for the three sizes the round's targets name (2^20 MSM, 2^20 lhs, 2^24 MSM)
set -e
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$NAME; mkdir -p $O
cd $R
python3 bench.py --logn 20 --steps 30 --warmup 5 --no-cpu-baseline "$@" > $O/b20.json 2> $O/err.txt
python3 bench.py --workload lhs --logn 20 --curve grumpkin --steps 20 --warmup 5 --no-cpu-baseline "$@" > $O/l20.json 2>> $O/err.txt
python3 bench.py --logn 24 --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/b24.json 2>> $O/err.txt
cd /tmp
rocprofv3 --kernel-trace -d $O/t20 -o t20 --output-format csv -- python3 $R/bench.py --logn 20 --steps 6 --warmup 2 --no-cpu-baseline "$@" > $O/t20.log 2>&1
rocprofv3 --kernel-trace -d $O/tl20 -o tl20 --output-format csv -- python3 $R/bench.py --workload lhs --curve grumpkin --logn 20 --steps 6 --warmup 2 --no-cpu-baseline "$@" > $O/tl20.log 2>&1
rocprofv3 --kernel-trace -d $O/t24 -o t24 --output-format csv -- python3 $R/bench.py --logn 24 --steps 4 --warmup 1 --no-cpu-baseline "$@" > $O/t24.log 2>&1
cd $R
for t in t20 t24; do python3 tools/trace_timeline.py $(find $O/$t -name "*kernel_trace.csv") > $O/$t.timeline.txt; done
python3 tools/trace_timeline.py --first "k_negbase_digits" $(find $O/tl20 -name "*kernel_trace.csv") > $O/tl20.timeline.txt || true
find $O -name "*.csv" -size +3M -delete
for f in b20 l20 b24; do python3 -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f ms/step', d['ms_per_step'], 'pairs/s %.4g' % d['value'], 'accum ms', d['roofline']['kernel_ms'], 'bit_exact', d['config'].get('bit_exact'))"; done
