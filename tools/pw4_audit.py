#!/usr/bin/env python3
"""Audit of the code hipcc emits for kernels_pw4.hip: the fused pointwise backward issues its persistent-accumulator MFMAs as inline asm ("+a" operands), and hipcc pads
no hazard inside an asm statement.  Checked here, on the assembly of the unit built with the product's flags:
  * no VALU instruction writes an A / B operand of an asm MFMA within the two instructions in front of it (VALU write -> MFMA operand read: two wait states),
    unless the statement carries its own `s_nop 1`;
  * the kernel uses no scratch memory, spills nothing and holds no v_accvgpr move inside its tile loop (the 256 accumulator registers stay where they are).
Usage: pw4_audit.py [file.s]   (without a file: compiles proba-v_amd/csrc/kernels_pw4.hip to a temporary directory first).  Exit code 1 on a finding."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_unit():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    d = tempfile.mkdtemp(prefix="pw4_audit_")
    src = os.path.join(ge.CSRC, "kernels_pw4.hip")
    out = os.path.join(d, "kernels_pw4.s")
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + ge.HIPCC_FLAGS + ge.UNIT_FLAGS["kernels_pw4.hip"] +
                          ["-I", os.path.join(ROOT, "include"), "--cuda-device-only", "-S", src, "-o", out], stderr=subprocess.DEVNULL)
    return out


def regs(tok):
    tok = tok.strip()
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def audit(path):
    text = open(path).read()
    findings = []
    ins = []
    for i, l in enumerate(text.splitlines()):
        m = re.match(r"\s+([a-z_0-9]+)\s+(.*)", l)
        if m and not l.lstrip().startswith((";", ".")):
            ins.append((i + 1, m.group(1), m.group(2)))
    n_asm = 0
    for k, (ln, op, a) in enumerate(ins):
        if not (op.startswith("v_mfma") and a.startswith("a[")):
            continue
        n_asm += 1
        ops = a.split(",")
        src = regs(ops[1]) | regs(ops[2])
        padded = ins[k - 1][1] == "s_nop" and int(ins[k - 1][2].split()[0]) >= 1
        if padded:
            continue
        for back in (1, 2):
            _, op2, a2 = ins[k - back]
            if op2.startswith("v_") and not op2.startswith("v_mfma") and regs(a2.split(",")[0]) & src:
                findings.append("line %d: %s %s reads an operand written %d instruction(s) earlier by %s %s" % (ln, op, a[:48], back, op2, a2[:40]))
    for key in ("vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size"):
        for m in re.finditer(r"\.%s:\s*(\d+)" % key, text):
            if int(m.group(1)) != 0:
                findings.append(".%s = %s" % (key, m.group(1)))
    # the tile loop: the basic block(s) between the loop header that holds the first in-loop MFMA and its back edge
    loop = re.search(r"Loop Header: Depth=1\n(.*?)s_cbranch_scc\d \.LBB\d+_\d+", text[text.find("v_mfma"):], re.S)
    big = max(re.findall(r"(\.LBB\d+_\d+:.*?)(?=\n\.LBB\d+_\d+:|\Z)", text, re.S), key=lambda b: b.count("v_mfma"))
    if "v_accvgpr" in big:
        findings.append("v_accvgpr move inside the tile loop")
    if "scratch_" in text:
        findings.append("scratch access")
    return n_asm, big.count("v_mfma"), findings


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else compile_unit()
    n_asm, n_loop, findings = audit(path)
    print("%s: %d asm MFMAs, %d MFMAs in the tile loop, %d finding(s)" % (path, n_asm, n_loop, len(findings)))
    for f in findings:
        print("  " + f)
    sys.exit(1 if findings else 0)
