#!/bin/bash
# Round 3, second session: fuzz soak (scatter_lean among the sampled options) and the headline bench lines on the same code.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03s4; mkdir -p $O
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_msm_2p24.json 2> $O/err.txt
python3 bench.py --logn 20 --steps 30 --warmup 5 > $O/bench_msm_2p20.json 2>> $O/err.txt
python3 bench.py --workload lhs --logn 20 --steps 20 --warmup 5 > $O/bench_lhs_2p20.json 2>> $O/err.txt
timeout -k 10 400 python3 tests/fuzz_gpu.py 300 910005 > $O/fuzz300.txt 2>&1 || { tail -30 $O/fuzz300.txt; exit 1; }
tail -2 $O/fuzz300.txt
for f in bench_msm_2p24 bench_msm_2p20 bench_lhs_2p20; do python3 -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], 'ms/step', '%.4g' % d['value'], d['unit'], 'cpu', (d.get('cpu_baseline') or {}).get('value'))"; done
