#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03h3; mkdir -p $O
for v in 0 2; do
rocprofv3 --kernel-trace -d $O/tr$v -o tr --output-format csv -- python3 $R/bench.py --workload lhs_witness --curve grumpkin --logn 20 --steps 1 --warmup 1 --no-cpu-baseline --option dw_halves=$v > $O/tr$v.log 2>&1
python3 - <<P > $O/levels$v.txt
import csv,glob
f=glob.glob('$O/tr$v/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_plan' in r['Kernel_Name']]
for lev in (-15,-10,-4):
    a=idx[lev]; b=idx[lev+1]
    t0=int(rows[a]['Start_Timestamp'])
    print('level', lev)
    for r in rows[a:b+1]:
        n=r['Kernel_Name'].split('(')[0][-40:]
        print(f"{n:42s} q={r.get('Queue_Id','?'):>3} start {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} end {(int(r['End_Timestamp'])-t0)/1e3:9.1f}")
a=idx[-19]
print('forest span ms', (int(rows[-1]['End_Timestamp'])-int(rows[a]['Start_Timestamp']))/1e6)
P
cat $O/levels$v.txt
done
find $O -name "*.csv" -size +3M -delete
