#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03x; mkdir -p $O
cd $R
LEMSM_DEBUG_STAMPS=1 timeout -k 10 500 python3 tools/gpu/r03_batchdbg.py > $O/dbg.txt 2>&1 || { tail -20 $O/dbg.txt; exit 1; }
grep -v "lemsm batch\|host tail\|amdgpu.ids" $O/dbg.txt
for cfg in "20 1 30" "20 8 8" "24 1 8" "24 4 3"; do
set -- $cfg
python3 bench.py --logn $1 --batch $2 --steps $3 --warmup 2 --no-cpu-baseline > $O/b$1_batch$2.json 2>> $O/err.txt
python3 -c "
import json
d=json.loads(open('$O/b$1_batch$2.json').read().strip().splitlines()[-1]); print('2^$1 batch $2: ms/step', d['ms_per_step'], 'pairs/s %.4g' % d['value'], 'bit_exact', d['config'].get('bit_exact'))"
done
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch" > $O/pytest_batch.txt 2>&1 || { tail -40 $O/pytest_batch.txt; exit 1; }
tail -2 $O/pytest_batch.txt
