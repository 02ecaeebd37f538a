#!/usr/bin/env python3
"""HBM traffic of k_accum1 per launch (FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes, as
MI355X_MICROARCH.md prescribes) for the bench workloads beyond the default one, and for the window-sharded ranks
rehearsed on one GPU (tools/sharded_sim_timing.py: the G ranks' launches one after the other).  This script never touches
the GPU itself; every pass is a child process.  usage (repo root, GPU box): tools/pmc_traffic.py OUTDIR
 -> OUTDIR/traffic_more.json, to be merged into profiles/traffic_accum1.json"""
import csv, glob, json, os, subprocess, sys
out = os.path.realpath(sys.argv[1]); os.makedirs(out, exist_ok=True)
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bench = ["python3", os.path.join(repo, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
sim = ["python3", os.path.join(repo, "tools", "sharded_sim_timing.py")]
cfgs = [
    ("lhs/bn254_g1/2^20/x1", bench + ["--workload", "lhs"], "bench.py --workload lhs"),
    ("msm/bn254_g1/2^20/x1", bench + ["--logn", "20"], "bench.py --logn 20"),
    ("msm/grumpkin/2^22/x1", bench + ["--curve", "grumpkin", "--logn", "22"], "bench.py --curve grumpkin --logn 22"),
    ("msm/bn254_g1/2^24/x2", sim + ["24", "2"], "tools/sharded_sim_timing.py 24 2 (ranks rehearsed on one GPU)"),
    ("msm/bn254_g1/2^24/x4", sim + ["24", "4"], "tools/sharded_sim_timing.py 24 4 (ranks rehearsed on one GPU)"),
    ("msm/bn254_g1/2^24/x8", sim + ["24", "8"], "tools/sharded_sim_timing.py 24 8 (ranks rehearsed on one GPU)"),
    ("msm/bn254_g1/2^26/x8", sim + ["26", "8"], "tools/sharded_sim_timing.py 26 8 (ranks rehearsed on one GPU; 4 slabs per rank: bytes per launch = per slab)"),
]
env = dict(os.environ, TMPDIR="/tmp")
res = {}
for key, cmd, label in cfgs:
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out, key.replace("/", "_").replace("^", "p"), counter)
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "t", "--"] + cmd,
                           cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if r.returncode:
            print("pass failed:", key, counter, r.returncode, flush=True); sys.exit(1)
        v = []
        for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(p)):
                if "k_accum1" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    v.append(float(row["Counter_Value"]))
        # the first dispatches of a process are warm-up of the same shape: all are averaged
        vals[counter] = (sum(v) / len(v), len(v)) if v else (None, 0)
    f, nf = vals["FETCH_SIZE"]; w, nw = vals["WRITE_SIZE"]
    if f is None or w is None:
        print("no k_accum1 dispatches:", key, flush=True); continue
    res[key] = {"bytes_per_launch": int(f * 1024 + w * 1024), "fetch_size_kb": f, "write_size_kb": w, "launches_averaged": [nf, nw],
                "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (one run) and --pmc WRITE_SIZE (another run) -- python3 " + label +
                       "; k_accum1 dispatch average; bytes = (FETCH_SIZE + WRITE_SIZE) * 1024, no gfx950 doubling for this kernel's 64-byte row gathers (profiles/r01/fetch_size_calibration.txt)"}
    print(key, res[key]["bytes_per_launch"], flush=True)
    json.dump(res, open(os.path.join(out, "traffic_more.json"), "w"), indent=1)
