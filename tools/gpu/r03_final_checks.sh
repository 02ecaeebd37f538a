#!/bin/bash
# final code: full GPU suite, then a 600-second fuzz soak
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_gpu_full.txt 2>&1 || { tail -40 $O/pytest_gpu_full.txt; exit 1; }
tail -12 $O/pytest_gpu_full.txt
timeout -k 10 700 python3 tests/fuzz_gpu.py 600 777001 > $O/fuzz600.txt 2>&1 || { tail -30 $O/fuzz600.txt; exit 1; }
tail -2 $O/fuzz600.txt
