"""Host tables and the cfg reader (no GPU)."""
import os

import pytest

from probav_amd.arch import layer_table, reducer_plan
from probav_amd.parseConfig import parseConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_variable_inventory_matches_reference_checkpoint():
    """modelInfo/ckpt_p16t9c85r12/NIR/ckpt-124.index: 44 weight-normalised layers, 535 267 fp32
    (SURVEY.md F3, A.1), in Keras topological order."""
    layers, total = layer_table()
    assert len(layers) == 44 and total == 535267 and total * 4 == 2141068
    names = [L.name for L in layers]
    assert names[0] == "mainConv1"
    assert names[1:4] == ["expConv_0", "decConv_0", "normConv_0"]
    assert names[37:] == ["convReducer_1", "convReducer_2", "convReducer_3", "residConv1", "upscaleConv1", "residConv2", "residConv3"]
    per = {L.name: 2 * L.cout + (L.b_off - L.v_off) for L in layers}
    assert per["mainConv1"] == 928 and per["expConv_3"] == 8704 and per["decConv_3"] == 6450
    assert per["normConv_3"] == 21664 and per["convReducer_2"] == 27712 and per["residConv1"] == 99
    assert per["upscaleConv1"] == 7794 and per["residConv2"] == 747
    assert layers[2].vshape == (1, 1, 1, 256, 25) and layers[40].vshape == (3, 3, 1, 9)
    # contiguous, non-overlapping flat layout
    off = 0
    for L in layers:
        assert (L.g_off, L.v_off) == (off, off + L.cout)
        off = L.b_off + L.cout
    assert off == total


def test_three_channel_branch_and_kernel_size():
    """isGrayScale=False touches the two input-facing layers only (models/modelsTF.py:19-20); kernelSize != 3 is refused like the reference's
    own graph refuses it (its valid reducers and residual path only close for 3: the docstring of WDSRConv3D.build)."""
    from probav_amd.modelsTF import WDSRConv3D
    layers, total = layer_table(inChannels=3)
    assert total == 535267 + 27 * 2 * 32 + 9 * 2 * 9
    assert layers[0].vshape == (3, 3, 3, 3, 32) and layers[40].vshape == (3, 3, 3, 9) and layers[1].vshape == (1, 1, 1, 32, 256)
    b = WDSRConv3D("t", "NIR", 8075.2045, 3160.7272, 6)
    m = b.build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, False, seed=0)
    assert m.inChannels == 3 and m.flat.numel() == total and len(m.trainable_variables) == 132
    assert b.build(3, 32, 3, 12, 8, 0.8, 9, 16, True, seed=0).inChannels == 1
    with pytest.raises(ValueError):
        b.build(3, 32, (5, 5, 5), 12, 8, 0.8, 9, 16, True)
    # why: the reference's 9-frame graph with k = 5 -- depth 9 -> 5 -> 1 -> negative at the third valid reducer (models/modelsTF.py:152-164)
    t = 9
    for _ in range(3):
        t -= 5 - 1
    assert t < 1


def test_reducer_plans():
    a, b = (3, 1, 0), (3, 0, 0)                              # (kernel, mirrored H/W pad, mirrored depth pad)
    assert reducer_plan(9) == (a, b, b)                      # models/modelsTF.py:152-164
    assert reducer_plan(13) == (a, a, a, b, b)               # :123-150
    assert reducer_plan(7) == (b, b)                         # :166-175
    assert reducer_plan(19) == ((5, 2, 2), (3, 2, 1), (3, 2, 0), (3, 2, 0), a, b, b, b, b, b)   # :76-121
    with pytest.raises(ValueError):
        reducer_plan(12)                                     # a literal 12-frame net is undefined
    assert layer_table(numImgLR=13)[1] == 535267 + 2 * 27712
    # 19 frames: seven more reducers than t9, the first with a 5x5x5 kernel (125*32*32 weights + g + bias)
    assert layer_table(numImgLR=19)[1] == 535267 + 6 * 27712 + (125 * 32 * 32 + 64)
    h, t = 22, 19                                            # the 19-frame chain must end at P x P x 1 before the Reshape (:71)
    for k, p, pt in reducer_plan(19) + ((3, 0, 0),):
        h, t = h + 2 * p - k + 1, t + 2 * pt - k + 1
    assert (h, t) == (16, 1)


@pytest.mark.parametrize("name", ["p16t9c85r12", "p16t12c85r12"])
def test_shipped_cfgs_parse(name):
    cfg = parseConfig(os.path.join(ROOT, "cfg", name + ".cfg"))
    assert cfg["num_low_res_imgs"] == 9 and cfg["num_res_blocks"] == 12 and cfg["num_filters"] == 32
    assert cfg["batch_size"] == 128 and cfg["learning_rate"] == 0.0005 and cfg["optimizer"] == "nadam" and cfg["loss"] == "l1"
    assert cfg["decay_rate"] == 0.8 and cfg["is_grayscale"] is True and cfg["max_shift"] == 6 and cfg["patch_size"] == 16
    assert cfg["ckpt"] == [1, 2, 3, 4, 5] and cfg["to_flip"] is False and isinstance(cfg["model_out"], str)
    assert "type" not in cfg


def test_cfg_typing_rules_and_whitelist(tmp_path):
    p = tmp_path / "x.cfg"
    p.write_text("# comment\n[Directories]\nanything_goes=here\nmodel_out = out dir \n\n[Train]\nbatch_size= 8\nsplit=0.2\nloss= l2 \n"
                 "[Net]\ndecay_rate=0.5\nscale=3\n[Preprocessing]\nlow_res_patch_thresholds=0.85,0.9\nhigh_res_threshold=0.85\nto_rotate=1\nckpt=2,3\n")
    cfg = parseConfig(str(p)[:-4])                       # '.cfg' is appended when missing
    assert cfg["anything_goes"] == "here" and cfg["model_out"] == "out dir"      # first section is not whitelisted
    assert cfg["batch_size"] == 8 and cfg["split"] == 0.2 and cfg["loss"] == "l2"
    assert cfg["low_res_patch_thresholds"] == [0.85, 0.9] and cfg["to_rotate"] is True and cfg["ckpt"] == [2, 3]
    bad = tmp_path / "bad.cfg"
    bad.write_text("[Directories]\nraw_data=x\n[Train]\nbatch_sizes=8\n")
    with pytest.raises(AssertionError, match="Unsupported fields"):
        parseConfig(str(bad))


# ---- pinned by the reference itself: tests/golden/cfg_cases.json holds what /root/reference/utils/parseConfig.py returned (or raised) for
# each cfg text when tests/golden/make_cfg_fixture.py ran it in the build container -- the one piece of the reference that imports here
def _cfg_cases():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "cfg_cases.json")) as fh:
        return json.load(fh)["cases"]


@pytest.mark.parametrize("case", _cfg_cases(), ids=lambda c: c["name"])
def test_parseConfig_matches_the_reference_parser(case, tmp_path):
    import builtins
    p = tmp_path / (case["name"] + ".cfg")
    p.write_text(case["text"])
    if "error" in case:
        with pytest.raises(getattr(builtins, case["error"])):
            parseConfig(str(p))
        return
    got = parseConfig(str(p))
    assert got == case["result"]
    for k, v in case["result"].items():                  # == lets True == 1 and 3 == 3.0 through: the types must agree too
        assert type(got[k]) is type(v), (k, got[k], v)
        if isinstance(v, list):
            assert [type(a) for a in got[k]] == [type(a) for a in v], k


def test_repo_cfgs_equal_the_reference_cfgs_outside_the_directory_paths():
    """cfg/*.cfg of this repo parse to the same dict as the reference's shipped files, except for the site-local [Directories] paths."""
    cases = {c["name"]: c for c in _cfg_cases()}
    for name in ("p16t9c85r12", "p16t12c85r12"):
        ours = parseConfig(os.path.join(ROOT, "cfg", name + ".cfg"))
        ref = cases["shipped_" + name]["result"]
        dirs = {"raw_data", "preprocessing_out", "model_out", "test_out", "train_out"}
        assert set(ours) == set(ref)
        assert {k: v for k, v in ours.items() if k not in dirs} == {k: v for k, v in ref.items() if k not in dirs}
