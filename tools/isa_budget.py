#!/usr/bin/env python3
"""Static ISA budget of a kernel's hot loop: instruction classes per loop iteration.

    python tools/isa_budget.py <file.s> <kernel-name-regex> [--whole]

--whole: histogram of the whole kernel (for straight-line bodies such as tools/ubench/madd_body.hip).

Reads hipcc -S output, finds the kernel, takes its largest innermost loop (a backward branch to a label with the
most instructions in between) and prints a histogram by class -- the table DESIGN.md's "instruction budget of one
mixed addition" is built from.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mad_i64_i32") or op.startswith("v_mad_u64_u32"):
        return "mad64 (29x29 product / reduction)"
    if op.startswith(("v_mul_lo", "v_mul_hi")):
        return "v_mul_lo/hi (Montgomery m_i)"
    if op.startswith(("v_ashrrev_i64", "v_lshrrev_b64", "v_lshlrev_b64", "v_lshl_add_u64")):
        return "64-bit shift/add (column carry)"
    if op.startswith("v_"):
        return "other VALU (and/add/sub/cndmask/mov/...)"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "VMEM"
    if op.startswith("scratch_"):
        return "scratch (spill)"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "SALU/other scalar"
    return "other"


def main():
    path, pat = sys.argv[1], re.compile(sys.argv[2])
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and ":" in l and pat.search(l.split(":")[0]))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {}
    instrs = []   # (index in body, op, text)
    for i, l in enumerate(body):
        t = l.strip()
        if not t or t.startswith((";", "//")):
            continue
        m = re.match(r"^(\.LBB[0-9_]+):", t)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        if t.startswith(".") or t.endswith(":"):
            continue
        instrs.append((i, t.split()[0], t))
    # backward branches
    best = None
    for k, (_, op, t) in enumerate(instrs):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] <= k:
                span = (labels[tgt], k)
                if best is None or span[1] - span[0] > best[1] - best[0]:
                    best = span
    print("kernel:", body[0].split(":")[0][:120])
    if "--whole" in sys.argv:
        best = (0, len(instrs) - 1)
    print("static instructions in kernel: %d; largest loop: %d instructions" % (len(instrs), best[1] - best[0] + 1))
    cnt = collections.Counter(classify(op) for _, op, _ in instrs[best[0]:best[1] + 1])
    tot = sum(cnt.values())
    valu = sum(v for k, v in cnt.items() if k.startswith(("mad64", "v_mul", "64-bit", "other VALU")))
    for k, v in cnt.most_common():
        print("  %-45s %5d  %5.1f %%" % (k, v, 100.0 * v / tot))
    print("  %-45s %5d" % ("VALU total", valu))
    print("  %-45s %5d" % ("all", tot))


if __name__ == "__main__":
    main()
