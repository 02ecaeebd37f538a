set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
python3 bench.py --logn 20 --steps 30 --warmup 5 --no-cpu-baseline > $O/b20.json 2> $O/b20.err
python3 bench.py --logn 20 --steps 30 --warmup 5 --no-cpu-baseline --option pyr_fuse=1 > $O/b20_nofuse.json 2>> $O/b20.err
python3 bench.py --workload lhs --logn 20 --curve grumpkin --steps 20 --warmup 5 --no-cpu-baseline > $O/l20.json 2>> $O/b20.err
python3 bench.py --logn 24 --steps 10 --warmup 3 --no-cpu-baseline > $O/b24.json 2>> $O/b20.err
python3 bench.py --logn 24 --steps 10 --warmup 3 --no-cpu-baseline --option pyr_fuse=1 > $O/b24_nofuse.json 2>> $O/b20.err
cd /tmp
rocprofv3 --kernel-trace -d $O/t20 -o t20 --output-format csv -- python3 $R/bench.py --logn 20 --steps 6 --warmup 2 --no-cpu-baseline > $O/t20.log 2>&1
rocprofv3 --kernel-trace -d $O/t20nf -o t20nf --output-format csv -- python3 $R/bench.py --logn 20 --steps 6 --warmup 2 --no-cpu-baseline --option pyr_fuse=1 > $O/t20nf.log 2>&1
rocprofv3 --kernel-trace -d $O/tl20 -o tl20 --output-format csv -- python3 $R/bench.py --workload lhs --curve grumpkin --logn 20 --steps 6 --warmup 2 --no-cpu-baseline > $O/tl20.log 2>&1
rocprofv3 --kernel-trace -d $O/t24 -o t24 --output-format csv -- python3 $R/bench.py --logn 24 --steps 4 --warmup 1 --no-cpu-baseline > $O/t24.log 2>&1
cd $R
for t in t20 t20nf t24; do python3 tools/trace_timeline.py $(find $O/$t -name "*kernel_trace.csv") > $O/$t.timeline.txt; done
python3 tools/trace_timeline.py --first "k_negbase_digits|k_count1" $(find $O/tl20 -name "*kernel_trace.csv") > $O/tl20.timeline.txt || true
find $O -name "*.csv" -size +3M -delete
for f in b20 b20_nofuse l20 b24 b24_nofuse; do python3 -c "
import json,sys
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])"; done
