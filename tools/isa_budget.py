#!/usr/bin/env python3
"""Instruction budget of a loop in hipcc's assembly output: counts per class between two line numbers (or labels).

    hipcc -O3 --offload-arch=gfx950 -std=c++17 --cuda-device-only -S -o kx6.s proba-v_amd/csrc/kernels_x6.hip
    python tools/isa_budget.py kx6.s <kernel-name-substring> [<first label> <last label>]

Without labels: the innermost loop that contains an s_barrier and the most v_mfma instructions (the interior tile body)."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "MFMA"
    if op.startswith(("ds_read", "ds_load")):
        return "DS_READ"
    if op.startswith(("ds_write", "ds_store")):
        return "DS_WRITE"
    if op.startswith("ds_"):
        return "DS_OTHER"
    if op.startswith(("global_load", "buffer_load", "flat_load")):
        return "VMEM_LOAD"
    if op.startswith(("global_store", "buffer_store", "flat_store")):
        return "VMEM_STORE"
    if op.startswith(("global_atomic", "buffer_atomic", "flat_atomic")):
        return "VMEM_ATOMIC"
    if op.startswith("scratch_"):
        return "SCRATCH"
    if op.startswith("s_waitcnt"):
        return "S_WAITCNT"
    if op.startswith("s_barrier"):
        return "S_BARRIER"
    if op.startswith(("s_load", "s_buffer_load")):
        return "SMEM"
    if op.startswith(("s_cbranch", "s_branch")):
        return "BRANCH"
    if op.startswith("s_nop"):
        return "S_NOP"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("v_"):
        return "VALU"
    return "OTHER"


def main():
    path, kern = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kern in l and l.rstrip().endswith(tuple(":")) or (l.startswith("_Z") and kern in l and ": " in l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    if len(sys.argv) >= 5:
        lo, hi = labels[sys.argv[3]], labels[sys.argv[4]]
    else:
        # back edges: a branch at line i to a label at line j < i closes a loop [j, i]
        loops = []
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        best = None
        for lo_, hi_ in loops:
            seg = body[lo_:hi_ + 1]
            nb = sum("s_barrier" in s for s in seg)
            nm = sum(s.strip().startswith("v_mfma") for s in seg)
            if nb >= 1 and nm and (best is None or (hi_ - lo_) < (best[1] - best[0])):
                best = (lo_, hi_)
        lo, hi = best
    cnt, ops = collections.Counter(), collections.Counter()
    for l in body[lo:hi + 1]:
        s = l.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":") or re.match(r"^\.LBB", s):
            continue
        op = s.split()[0]
        cnt[classify(op)] += 1
        ops[op] += 1
    total = sum(cnt.values())
    print("kernel %s, lines %d..%d of its body (%s .. %s)" % (kern, lo, hi, body[lo].split(":")[0], body[hi].strip()))
    print("total instructions %d, of which MFMA %d -> non-MFMA %d" % (total, cnt["MFMA"], total - cnt["MFMA"]))
    for k, v in cnt.most_common():
        print("  %-12s %4d" % (k, v))
    print("opcodes:")
    for k, v in ops.most_common(40):
        print("  %-28s %4d" % (k, v))


if __name__ == "__main__":
    main()
