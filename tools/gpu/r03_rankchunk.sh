#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O
cd $R
for v in 96 112 128 144 160 192 0 128; do
SIM_OPTIONS=chunk=$v python3 tools/sharded_sim_timing.py 24 8 > $O/sim24_chunk$v.txt 2>> $O/err.txt
echo "chunk=$v $(cat $O/sim24_chunk$v.txt)"
done
