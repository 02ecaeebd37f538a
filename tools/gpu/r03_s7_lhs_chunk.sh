#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03s7; mkdir -p $O
cd $R
AB_WORKLOAD=lhs AB_ROUNDS=7 timeout -k 10 300 python3 tools/ab_bench.py 20 "" "chunk=64" "chunk=128" "chunk=176" "chunk=256" "merge_slice=256" "merge_slice=1024" "chunk=176,merge_slice=256" "chunk=128,merge_slice=256" > $O/ab_lhs_chunk.txt 2>&1
cat $O/ab_lhs_chunk.txt
