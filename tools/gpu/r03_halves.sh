#!/bin/bash
# A/B of the two-half pointwise chains of the divisor-witness levels, the witness parity tests, and a ROCTX sanity run
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03h; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "witness or divisor or reuse or lhs" > $O/pytest_witness.txt 2>&1 || { tail -30 $O/pytest_witness.txt; exit 1; }
tail -3 $O/pytest_witness.txt
for v in 0 2 0 2; do
python3 bench.py --workload lhs_witness --curve grumpkin --logn 20 --steps 4 --warmup 1 --no-cpu-baseline --option dw_halves=$v > $O/w20_halves$v.json 2>> $O/err.txt
python3 -c "
import json
d=json.loads(open('$O/w20_halves$v.json').read().strip().splitlines()[-1]); print('dw_halves=$v ms/step', d['ms_per_step'], 'bit_exact', d['config'].get('bit_exact'))"
done
cd /tmp
LEMSM_ROCTX=1 rocprofv3 --kernel-trace --marker-trace -d $O/roctx -o roctx --output-format csv -- python3 $R/bench.py --logn 20 --steps 3 --warmup 1 --no-cpu-baseline > $O/roctx.log 2>&1
ls $O/roctx/* | head
f=$(find $O/roctx -name "*marker*trace.csv" | head -1); echo "marker file: $f"; head -12 "$f" | cut -c1-200
find $O -name "*.csv" -size +3M -delete
