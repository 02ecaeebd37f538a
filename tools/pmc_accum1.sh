#!/bin/bash
# SQ issue / wait counters of the accumulate kernel (two passes: 8 SQ slots each); run from the repo root on the GPU box.
# usage: tools/pmc_accum1.sh OUTDIR
set -e
OUT=$(realpath "$1"); mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE \
  -d "$OUT/passA" -o a --output-format csv -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/passA.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
  -d "$OUT/passB" -o b --output-format csv -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/passB.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_IOPS SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
  -d "$OUT/passC" -o c --output-format csv -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/passC.log" 2>&1
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in sorted(glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True)):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if "k_accum1" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(p.replace(out + "/", ""))
    for k, v in d.items():
        print("  %-26s n=%d avg=%.6g" % (k, len(v), sum(v) / len(v)))
PY
