"""The fused pointwise backward of round 5 (proba-v_amd/csrc/kernels_pw4.hip) issues its persistent-accumulator MFMAs as inline asm, where hipcc inserts no wait
states: the code it emits for the unit is audited on every build (tools/pw4_audit.py: no VALU write of an MFMA operand within two instructions of an unpadded asm
MFMA, no scratch, no spill, no v_accvgpr move in the tile loop).  Cross-compiles for gfx950 without a GPU (a few seconds)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_emitted_code_of_the_one_wave_per_simd_kernel():
    spec = importlib.util.spec_from_file_location("pw4_audit", os.path.join(ROOT, "tools", "pw4_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    path = mod.compile_unit()
    n_asm, n_loop, findings = mod.audit(path)
    assert not findings, findings
    assert n_loop == 240, n_loop                     # 8 chunks x 30 MFMAs per tile, all in ONE basic block (a branch inside the loop drains every counter)
    assert n_asm >= 96 + 12
    # the audit itself must see a hazard when there is one: a VALU write of an operand right in front of an unpadded asm MFMA
    text = open(path).read().splitlines()
    k = next(i for i, l in enumerate(text) if "v_mfma_f32_32x32x16_f16 a[" in l and "s_nop" not in text[i - 1])
    ops = text[k].split("v_mfma_f32_32x32x16_f16")[1].split(",")
    lo = int(ops[1].strip()[2:].split(":")[0])
    bad = text[:k] + ["\tv_mov_b32_e32 v%d, 0" % lo] + text[k:]
    tmp = path + ".bad.s"
    with open(tmp, "w") as fh:
        fh.write("\n".join(bad))
    assert mod.audit(tmp)[2]
