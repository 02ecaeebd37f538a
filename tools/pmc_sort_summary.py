#!/usr/bin/env python3
"""Per-kernel averages of the counter passes tools/pmc_sort_kernels.sh wrote (gpurun_out/pmc/sc1_*.csv): one line per
counter, averaged over the dispatches of each sort kernel of a 2^24-point MSM, plus the kernel's average duration.
usage: pmc_sort_summary.py [DIR]"""
import csv, glob, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "pmc")
kern = ["k_pip_digits", "k_scatter1", "k_binsort", "k_count2", "k_scatter2", "k_segreduce", "k_segwave"]
acc = {}
dur = {}
for f in sorted(glob.glob(os.path.join(d, "sc1_*.csv"))):
    for r in csv.DictReader(open(f)):
        name = next((k for k in kern if k in r["Kernel_Name"]), None)
        if not name:
            continue
        a = acc.setdefault(name, {}).setdefault(r["Counter_Name"], [0.0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += 1
        t = dur.setdefault(name, [0.0, 0])
        t[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); t[1] += 1
print("rocprofv3 --kernel-trace --pmc <4 counters per pass> -- python3 tools/ab_bench.py 24 field=0 (tools/pmc_sort_kernels.sh); 2^24-point BN254 MSM, 17-bit windows")
for k in kern:
    if k not in acc:
        continue
    print("%s   (avg duration under the counter passes: %.1f us)" % (k, dur[k][0] / dur[k][1] / 1e3))
    for c, (s, n) in acc[k].items():
        print("   %-28s %12.4g   (avg over %d dispatches)" % (c, s / n, n))
