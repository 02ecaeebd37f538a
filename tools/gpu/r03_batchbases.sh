#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03v; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch" > $O/pytest_batch.txt 2>&1 || { tail -40 $O/pytest_batch.txt; exit 1; }
tail -3 $O/pytest_batch.txt
timeout -k 10 900 python3 tools/host_path_timing.py 24 > $O/host_path_2p24.txt 2>&1 || { tail -20 $O/host_path_2p24.txt; exit 1; }
cat $O/host_path_2p24.txt
