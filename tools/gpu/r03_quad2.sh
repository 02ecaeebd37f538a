#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03q2; mkdir -p $O
cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quad_additions" > $O/pytest_quad.txt 2>&1 || { tail -40 $O/pytest_quad.txt; exit 1; }
tail -2 $O/pytest_quad.txt
for v in 0 2 0 2; do
python3 bench.py --logn 20 --steps 40 --warmup 5 --no-cpu-baseline --option pyr_quad=$v > $O/b20_quad$v.json 2>> $O/err.txt
python3 -c "
import json
d=json.loads(open('$O/b20_quad$v.json').read().strip().splitlines()[-1]); print('2^20 pyr_quad=$v ms/step', d['ms_per_step'], 'bit_exact', d['config'].get('bit_exact'))"
done
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu_full.txt 2>&1 || { tail -40 $O/pytest_gpu_full.txt; exit 1; }
tail -2 $O/pytest_gpu_full.txt
timeout -k 10 300 python3 tests/fuzz_gpu.py 200 424242 > $O/fuzz200.txt 2>&1 || { tail -30 $O/fuzz200.txt; exit 1; }
tail -1 $O/fuzz200.txt
