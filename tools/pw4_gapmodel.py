#!/usr/bin/env python3
"""Static issue model of a one-wave-per-SIMD MFMA stream (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'): the loop's instructions are cut into gaps at every
v_mfma; a gap takes max(32, 8 + sum of the issue costs of its other instructions).  Usage: pw4_gapmodel.py kernels_pw4.s (the whole unit: the basic block with the most MFMAs is taken) or a loop body"""
import re, sys
COST = [(r"v_fma_mix", 8.8), (r"v_cvt_pk_f16_f32", 8.0), (r"v_cmp", 8.3), (r"v_dot2", 10.0), (r"v_accvgpr", 4.9), (r"v_", 4.9), (r"ds_write_b128|ds_write2st64_b64", 13.0), (r"ds_write_b64", 6.0),
        (r"ds_write", 4.0), (r"ds_read_b128", 4.0), (r"ds_", 2.0), (r"buffer_|global_", 4.0), (r"s_nop", None), (r"s_waitcnt", 1.0), (r"s_memtime", 4.0), (r"s_", 1.0)]
def cost(op, arg):
    for pat, c in COST:
        if re.match(pat, op):
            if c is None:
                return 4.0 * (int(arg.split()[0]) + 1)
            return c
    return 4.0
text = open(sys.argv[1]).read()
blocks = re.findall(r"(\.LBB\d+_\d+:.*?)(?=\n\.LBB\d+_\d+:|\Z)", text, re.S)
if blocks:                                  # a whole .s file: the tile loop is the basic block with the most MFMAs
    text = max(blocks, key=lambda b: b.count("v_mfma"))
gaps, cur, nm = [], [], 0
for line in text.splitlines():
    m = re.match(r"\s+([a-z_0-9]+)\s*(.*)", line)
    if not m or line.lstrip().startswith(";"):
        continue
    op, arg = m.group(1), m.group(2)
    if op.startswith("v_mfma"):
        gaps.append(cur); cur = []; nm += 1
    else:
        cur.append((op, arg))
gaps.append(cur)
tot = sum(max(32.0, 8.0 + sum(cost(o, a) for o, a in g)) for g in gaps[1:]) + sum(cost(o, a) for o, a in gaps[0])
raw = sum(sum(cost(o, a) for o, a in g) for g in gaps)
print("MFMAs %d; sum of non-MFMA issue costs %.0f; modelled cycles per loop iteration %.0f (%.0f per chunk iteration); MFMA floor %d" % (nm, raw, tot, tot / 8, 32 * nm))
hist = {}
for g in gaps[1:]:
    c = 8.0 + sum(cost(o, a) for o, a in g)
    b = int(c // 16) * 16
    hist[b] = hist.get(b, 0) + 1
print("gap lengths (cycles, incl. the MFMA's 8): " + ", ".join("%d-%d: %d" % (k, k + 15, v) for k, v in sorted(hist.items())))
