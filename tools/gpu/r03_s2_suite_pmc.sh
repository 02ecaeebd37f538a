#!/bin/bash
# Round 3, second session: LDS rank-rate microbenchmark, counter passes of the sort kernels (shipped k_scatter1), full GPU suite.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03s2; mkdir -p $O
cd $R
timeout -k 10 120 tools/ubench/bin/lds_rank_rates > $O/lds_rank_rates.txt 2>&1
bash tools/pmc_sort_kernels.sh > $O/pmc_sort.log 2>&1 || true
cp gpurun_out/pmc/sort_kernels_summary.txt $O/ || true
timeout -k 10 800 python3 -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu_full.txt 2>&1
tail -3 $O/pytest_gpu_full.txt; cat $O/lds_rank_rates.txt
