#!/bin/bash
# Refreshes the PMC-derived inputs of bench.py's roofline for the default workload (2^24 BN254 MSM): HBM traffic of
# k_accum1 (FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes) and its VALU instruction
# count / clock.  Run from the repo root on the GPU box:  tools/pmc_refresh.sh OUTDIR   -> OUTDIR/{traffic,valu}_accum1.json
set -e
OUT=$(realpath "$1"); mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp; export TMPDIR=/tmp
CMD="python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- $CMD > "$OUT/write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/valu" -o v -- $CMD > "$OUT/valu.log" 2>&1
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
def avg(pass_dir, counter):
    vals, durs = [], []
    for p in glob.glob(out + "/" + pass_dir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "k_accum1" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"])); durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return (sum(vals) / len(vals), sum(durs) / len(durs), len(vals)) if vals else (None, None, 0)
f, _, nf = avg("fetch", "FETCH_SIZE"); w, _, nw = avg("write", "WRITE_SIZE")
v, dur, nv = avg("valu", "SQ_INSTS_VALU"); g, _, _ = avg("valu", "GRBM_GUI_ACTIVE")
key = "msm/bn254_g1/2^24/x1"
if f is not None and w is not None:
    json.dump({key: {"bytes_per_launch": int(f * 1024 + w * 1024), "fetch_size_kb": f, "write_size_kb": w, "launches_averaged": [nf, nw],
                     "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (one run) and --pmc WRITE_SIZE (another run) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline; k_accum1 dispatch average; bytes = FETCH_SIZE*1024*f + WRITE_SIZE*1024 with f = 1.00: the gfx950 FETCH_SIZE halving of MI355X_MICROARCH.md applies to wide coalesced streams, not to this kernel's 64-byte row gathers (calibration: profiles/r01/fetch_size_calibration.txt, tools/ubench/gather_calib.hip)"}},
              open(out + "/traffic_accum1.json", "w"), indent=1)
if v is not None:
    madds = 15 * (1 << 24)
    json.dump({key: {"sq_insts_valu_per_launch": v, "madds_per_launch": madds, "instr_per_madd": round(v * 64 / madds, 1),
                     "grbm_gui_active_per_launch": g, "kernel_ns": dur, "clock_ghz_from_pmc": round(g / 8 / dur, 4), "launches_averaged": nv,
                     "how": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline, k_accum1 dispatch average; instr_per_madd = SQ_INSTS_VALU x 64 lanes / (15 windows x 2^24 pairs); clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time"}},
              open(out + "/valu_accum1.json", "w"), indent=1)
print(open(out + "/traffic_accum1.json").read() if f is not None else "no traffic")
print(open(out + "/valu_accum1.json").read() if v is not None else "no valu")
PY
