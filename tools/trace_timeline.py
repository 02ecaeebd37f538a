#!/usr/bin/env python3
"""Timeline of ONE step out of a rocprofv3 --kernel-trace CSV: kernel, start (us from the step's first
kernel), duration, gap to the previous kernel's end, grid.  A step is delimited by the digit kernel
(k_pip_digits / k_negbase_digits / k_count1) that opens every pipeline pass; the LAST complete step of the
trace is printed (warm caches, warm clocks).

    python tools/trace_timeline.py <kernel_trace.csv> [--first k_pip_digits] [--step -2]
"""
import argparse
import csv
import re


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"([A-Za-z0-9_:]+)", name)
    return m.group(1) if m else name[:40]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--first", default="k_pip_digits|k_negbase_digits", help="regex of the kernel that opens a step")
    ap.add_argument("--step", type=int, default=-2, help="which step to print (index into the list of steps; -2 = last complete one)")
    args = ap.parse_args()
    rows = []
    with open(args.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]), int(r["Workgroup_Size_X"]),
                         r.get("VGPR_Count", ""), r.get("Scratch_Size", ""), r.get("LDS_Block_Size", "")))
    rows.sort()
    opener = re.compile(args.first)
    starts = [i for i, r in enumerate(rows) if opener.search(r[2])]
    if len(starts) < 2:
        raise SystemExit("fewer than two steps in the trace")
    k = args.step % len(starts)
    i0 = starts[k]
    i1 = starts[k + 1] if k + 1 < len(starts) else len(rows)
    t0 = rows[i0][0]
    prev_end = None
    busy = 0
    print("%-34s %10s %9s %8s %10s %5s %5s %7s %6s" % ("kernel", "start us", "dur us", "gap us", "grid", "wg", "vgpr", "scratch", "lds"))
    for s, e, name, grid, wg, vg, sc, lds in rows[i0:i1]:
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        print("%-34s %10.1f %9.1f %8.1f %10d %5d %5s %7s %6s" % (short(name)[:34], (s - t0) / 1e3, (e - s) / 1e3, gap, grid, wg, vg, sc, lds))
        busy += e - s
        prev_end = e if prev_end is None else max(prev_end, e)
    print("step: %.1f us first-start to last-end, %.1f us of kernel time, %d launches" % ((prev_end - t0) / 1e3, busy / 1e3, i1 - i0))


if __name__ == "__main__":
    main()
