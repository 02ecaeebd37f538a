#!/bin/bash
# Counter passes over the kernels of compute_lhs_witness in full (2^LOGN points, base 16): separate --pmc passes with
# --kernel-trace only (the pool refuses --pmc together with the runtime / memory-copy trace domains), FETCH_SIZE and
# WRITE_SIZE in passes of their own (MI355X_MICROARCH.md, HBM section).  usage: bash tools/gpu/pmc_witness.sh <outdir-name> [LOGN]
set -e
NAME=$1; LOGN=${2:-18}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$NAME; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/w$i -o w -- python3 $R/tools/lhs_witness_profile.py $LOGN > $O/pass$i.log 2>&1
  cp /tmp/w$i/w_counter_collection.csv $O/pmc_pass$i.csv
done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ws -o w -- python3 $R/tools/lhs_witness_profile.py $LOGN > $O/stats.log 2>&1
cp /tmp/ws/w_kernel_stats.csv $O/kernel_stats.csv
cd $R
python3 tools/pmc_witness_summary.py $O $LOGN > $O/summary.txt
find $O -name "pmc_pass*.csv" -size +2M -delete
cat $O/summary.txt
