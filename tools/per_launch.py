#!/usr/bin/env python3
"""Per-launch durations (us, in launch order) of the kernels whose name matches REGEX, out of a rocprofv3 --kernel-trace CSV.
usage: per_launch.py <kernel_trace.csv> REGEX [REGEX ...]"""
import csv, re, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", "")) for r in csv.DictReader(open(sys.argv[1]))), key=lambda t: t[0])
for pat in sys.argv[2:]:
    sel = [(e - s) / 1e3 for s, e, n, g in rows if re.search(pat, n)]
    grids = [g for s, e, n, g in rows if re.search(pat, n)]
    print("%s: %d launches, total %.1f us" % (pat, len(sel), sum(sel)))
    print("   us  : " + " ".join("%.0f" % v for v in sel))
    print("   grid: " + " ".join(grids))
