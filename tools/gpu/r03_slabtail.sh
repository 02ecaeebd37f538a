#!/bin/bash
# shared tail across the slabs of a call: parity tests, fuzz, then timing at 2^26 on one GPU and as one of 8 simulated ranks
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03t; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "slab or sharded or multi_slab or skew or long_bucket" > $O/pytest_slab.txt 2>&1 || { tail -40 $O/pytest_slab.txt; exit 1; }
tail -3 $O/pytest_slab.txt
timeout -k 10 300 python3 tests/fuzz_gpu.py 120 3031 > $O/fuzz120.txt 2>&1 || { tail -30 $O/fuzz120.txt; exit 1; }
tail -3 $O/fuzz120.txt
for v in 0 2; do
python3 bench.py --logn 26 --steps 3 --warmup 1 --no-cpu-baseline --option slab_tail=$v > $O/b26_tail$v.json 2>> $O/err.txt
python3 -c "
import json
d=json.loads(open('$O/b26_tail$v.json').read().strip().splitlines()[-1]); print('2^26 slab_tail=$v ms/step', d['ms_per_step'], 'bit_exact', d['config'].get('bit_exact'))"
done
for v in 0 2; do
SIM_OPTIONS=slab_tail=$v python3 tools/sharded_sim_timing.py 26 1 8 > $O/sim26_tail$v.txt 2>> $O/err.txt
echo "slab_tail=$v"; cat $O/sim26_tail$v.txt
done
