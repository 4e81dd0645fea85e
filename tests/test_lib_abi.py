"""The C-ABI library builds for gfx950, loads, and exports every symbol include/probav_hip.h declares
(no compute calls: there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "probav_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(probav_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built_lib):
    from probav_amd import _lib
    L = ctypes.CDLL(built_lib)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libprobav_hip.so does not export %s" % n
    assert set(names) == set(_lib.SIGNATURES), "ctypes binding and header disagree"
    assert _lib.lib().probav_abi_version() == 7


def test_no_torch_types_in_the_abi():
    txt = open(os.path.join(ROOT, "include", "probav_hip.h")).read()
    assert "torch" not in txt.lower().replace("pytorch", "") and "at::" not in txt and "#include <hip" not in txt


def test_product_path_fails_loudly_without_a_device(built_lib):
    """No CPU fallback: a CPU tensor is refused instead of being computed some other way."""
    from probav_amd.modelsTF import WDSRConv3D
    from probav_amd.loss import Losses
    model = WDSRConv3D("t", "NIR", 8075.2045, 3160.7272, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True, seed=0)
    assert model.flat.numel() == 535267 and len(model.trainable_variables) == 132
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.zeros(1, 22, 22, 9, 1))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Losses(targetShape=(48, 48, 1)).shiftCompensatedL1Loss(torch.zeros(1, 48, 48, 1), torch.ones(1, 48, 48, 1, dtype=torch.bool), torch.zeros(1, 48, 48, 1))
    with pytest.raises(ValueError):
        WDSRConv3D("t", "NIR", 1.0, 1.0, 6).build(3, 32, (5, 5, 5), 12, 8, 0.8, 9, 16, True)


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    from probav_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="missing"):
        _lib.lib()


def test_initial_state_is_the_reference_first_call_state():
    """TFA WeightNormalization(data_init=False): after the first call g == ||v|| and bias == 0, so the
    effective kernel equals v."""
    from probav_amd.modelsTF import WDSRConv3D
    model = WDSRConv3D("t", "RED", 5266.2245, 3431.8614, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True, seed=1)
    names, tv = model.variable_names, model.trainable_variables
    assert names[:3] == ["mainConv1/g", "mainConv1/v", "mainConv1/bias"] and names[-1] == "residConv3/bias"
    for k in range(0, len(tv), 3):
        g, v, b = tv[k], tv[k + 1], tv[k + 2]
        torch.testing.assert_close(g, v.reshape(-1, v.shape[-1]).pow(2).sum(0).sqrt(), rtol=1e-5, atol=1e-7)
        assert float(b.abs().max()) == 0.0
        fan = (v.numel() // (v.shape[-1] * v.shape[-2])) * (v.shape[-1] + v.shape[-2])
        assert float(v.abs().max()) <= (6.0 / fan) ** 0.5 + 1e-7


def test_build_knows_the_units_that_include_other_units():
    """kernels_wg4b.hip / kernels_wg4c.hip are the backward-filter kernel's other instances: they include kernels_wg4.hip (three translation units for the build time's sake),
    so build() must rebuild them when THAT file changes and compile them with its flags."""
    import __graft_entry__ as ge
    for name in ("kernels_wg4b.hip", "kernels_wg4c.hip"):
        deps = ge._included_units(os.path.join(ge.CSRC, name))
        assert [os.path.basename(d) for d in deps] == ["kernels_wg4.hip"], (name, deps)
        assert ge.UNIT_FLAGS[name] == ge.UNIT_FLAGS["kernels_wg4.hip"]
    assert ge._included_units(os.path.join(ge.CSRC, "kernels_wg4.hip")) == []
