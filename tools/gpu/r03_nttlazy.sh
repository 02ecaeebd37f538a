#!/bin/bash
# lazy-field transform passes: witness parity tests, then A/B against the strict-field kernel at 2^20 and 2^18
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03n; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "witness or divisor or reuse or lhs" > $O/pytest_witness.txt 2>&1 || { tail -30 $O/pytest_witness.txt; exit 1; }
tail -3 $O/pytest_witness.txt
for v in 0 2 0 2; do
python3 bench.py --workload lhs_witness --curve grumpkin --logn 20 --steps 4 --warmup 1 --no-cpu-baseline --option dw_ntt_lazy=$v > $O/w20_nttlazy$v.json 2>> $O/err.txt
python3 -c "
import json
d=json.loads(open('$O/w20_nttlazy$v.json').read().strip().splitlines()[-1]); print('dw_ntt_lazy=$v ms/step', d['ms_per_step'], 'bit_exact', d['config'].get('bit_exact'))"
done
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/st -o st --output-format csv -- python3 $R/bench.py --workload lhs_witness --curve grumpkin --logn 20 --steps 3 --warmup 1 --no-cpu-baseline > $O/st.log 2>&1
f=$(find $O/st -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-60,200-400
python3 - <<P
import csv,glob
f=glob.glob('$O/st/*kernel_trace.csv')[0]
seen={}
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name'].split('(')[0]
    if 'ntt_tile' in n and n not in seen:
        seen[n]=(r.get('VGPR_Count'), r.get('Scratch_Size'), r.get('LDS_Block_Size'))
print(seen)
P
find $O -name "*.csv" -size +3M -delete
