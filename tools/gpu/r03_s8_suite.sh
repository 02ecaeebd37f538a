#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03s8; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu_full.txt 2>&1 || { tail -40 $O/pytest_gpu_full.txt; exit 1; }
tail -12 $O/pytest_gpu_full.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; cat $O/smoke.txt | tail -2
