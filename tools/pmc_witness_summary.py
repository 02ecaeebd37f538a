#!/usr/bin/env python3
"""Per-kernel totals of the counter passes tools/gpu/pmc_witness.sh wrote: for every kernel of the second (warm)
compute_lhs_witness call -- summed over its dispatches -- duration, VALU instructions, HBM bytes (FETCH_SIZE / WRITE_SIZE in KB
as rocprofv3 reports them; FETCH_SIZE doubled for the transform kernels' wide 16-byte-per-lane streaming reads per
MI355X_MICROARCH.md), LDS bank-conflict share, wait share.  usage: pmc_witness_summary.py DIR LOGN"""
import csv, glob, os, re, sys
d = sys.argv[1]; logn = sys.argv[2] if len(sys.argv) > 2 else "?"
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", n); return (m.group(1) if m else n)[:60]
tot = {}
for f in sorted(glob.glob(os.path.join(d, "pmc_pass*.csv"))):
    rows = list(csv.DictReader(open(f)))
    if not rows: continue
    # the second call only: dispatches in the later half of the trace by Dispatch_Id
    ids = sorted(int(r["Dispatch_Id"]) for r in rows)
    mid = ids[len(ids) // 2]
    seen = set()
    for r in rows:
        if int(r["Dispatch_Id"]) < mid: continue
        k = short(r["Kernel_Name"]); c = r["Counter_Name"]
        t = tot.setdefault(k, {})
        t[c] = t.get(c, 0.0) + float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            t["_ns_" + os.path.basename(f)] = t.get("_ns_" + os.path.basename(f), 0.0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            t["_n"] = t.get("_n", 0) + 1
print("compute_lhs_witness in full, 2^%s points, base 16, second call; per kernel, summed over its dispatches (tools/gpu/pmc_witness.sh)" % logn)
print("%-62s %9s %12s %10s %10s %8s %8s" % ("kernel", "ms", "VALU inst", "fetch MB", "write MB", "LDSconf", "wait"))
order = sorted(tot.items(), key=lambda kv: -max([v for k, v in kv[1].items() if k.startswith("_ns_")] or [0]))
for k, t in order:
    ns = max([v for kk, v in t.items() if kk.startswith("_ns_")] or [0])
    if ns < 2e5: continue
    conf = t.get("SQ_LDS_BANK_CONFLICT", 0) / t["SQ_ACTIVE_INST_LDS"] if t.get("SQ_ACTIVE_INST_LDS") else float("nan")
    wait = t.get("SQ_WAIT_INST_ANY", 0) / t["SQ_WAVE_CYCLES"] if t.get("SQ_WAVE_CYCLES") else float("nan")
    print("%-62s %9.2f %12.4g %10.1f %10.1f %8.3f %8.3f" % (k, ns / 1e6, t.get("SQ_INSTS_VALU", float("nan")), t.get("FETCH_SIZE", float("nan")) / 1024, t.get("WRITE_SIZE", float("nan")) / 1024, conf, wait))
