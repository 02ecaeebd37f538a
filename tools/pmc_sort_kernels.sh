set -e
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/pmc
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  AB_ROUNDS=2 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/q$i -o q -- python3 $GRAFT_REPO_ROOT/tools/ab_bench.py 24 field=0 > /dev/null 2>&1
  cp /tmp/q$i/q_counter_collection.csv $GRAFT_REPO_ROOT/gpurun_out/pmc/sc1_$i.csv
done
python3 $GRAFT_REPO_ROOT/tools/pmc_sort_summary.py $GRAFT_REPO_ROOT/gpurun_out/pmc > $GRAFT_REPO_ROOT/gpurun_out/pmc/sort_kernels_summary.txt
