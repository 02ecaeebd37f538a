#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03q; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "msm and not slab and not host_entry" > $O/pytest_msm.txt 2>&1 || { tail -40 $O/pytest_msm.txt; exit 1; }
tail -2 $O/pytest_msm.txt
for v in 0 2 0 2; do
python3 bench.py --logn 20 --steps 40 --warmup 5 --no-cpu-baseline --option pyr_quad=$v > $O/b20_quad$v.json 2>> $O/err.txt
python3 -c "
import json
d=json.loads(open('$O/b20_quad$v.json').read().strip().splitlines()[-1]); print('2^20 pyr_quad=$v ms/step', d['ms_per_step'], 'bit_exact', d['config'].get('bit_exact'))"
done
for v in 0 2; do
python3 bench.py --workload lhs --logn 20 --steps 20 --warmup 5 --no-cpu-baseline --option pyr_quad=$v > $O/l20_quad$v.json 2>> $O/err.txt
python3 -c "
import json
d=json.loads(open('$O/l20_quad$v.json').read().strip().splitlines()[-1]); print('lhs 2^20 pyr_quad=$v ms/step', d['ms_per_step'], 'bit_exact', d['config'].get('bit_exact'))"
SIM_OPTIONS=pyr_quad=$v python3 tools/sharded_sim_timing.py 24 8
done
cd /tmp
rocprofv3 --kernel-trace -d $O/t20 -o t20 --output-format csv -- python3 $R/bench.py --logn 20 --steps 6 --warmup 2 --no-cpu-baseline > $O/t20.log 2>&1
cd $R
python3 tools/trace_timeline.py $(find $O/t20 -name "*kernel_trace.csv") > $O/t20.timeline.txt
sed -n '/k_merge_final/,$p' $O/t20.timeline.txt
find $O -name "*.csv" -size +3M -delete
