#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c2; mkdir -p $O
cd $R
run() { # logn chunk steps workload
python3 bench.py --workload $4 --logn $1 --steps $3 --warmup 5 --no-cpu-baseline --option chunk=$2 > $O/$4_$1_chunk$2.json 2>> $O/err.txt
python3 -c "
import json
d=json.loads(open('$O/$4_$1_chunk$2.json').read().strip().splitlines()[-1]); print('$4 2^$1 chunk=$2 ms/step', d['ms_per_step'], 'accum', d['roofline'].get('kernel_ms'))"
}
for rep in 1 2; do
run 18 0 40 msm; run 18 16 40 msm
run 19 0 40 msm; run 19 32 40 msm
run 20 0 40 msm; run 20 64 40 msm
run 21 0 30 msm; run 21 128 30 msm
run 22 0 20 msm; run 22 256 20 msm
run 20 0 20 lhs; run 20 132 20 lhs; run 20 112 20 lhs
done
