#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03s10; mkdir -p $O
cd $R
timeout -k 10 500 python3 tests/fuzz_gpu.py 400 20261005 > $O/fuzz400.txt 2>&1 || { tail -30 $O/fuzz400.txt; exit 1; }
tail -2 $O/fuzz400.txt
