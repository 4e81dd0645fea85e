"""TensorFlow checkpoint-bundle reader / writer (no TensorFlow): the reference's own index table and a round trip."""
import json
import os

import numpy as np
import pytest
import torch

from probav_amd import tfckpt
from probav_amd.arch import layer_table
from probav_amd.modelsTF import WDSRConv3D

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ckpt124_index.json")
REF = "/root/reference/modelInfo/ckpt_p16t9c85r12/NIR/ckpt-124"


def test_reference_index_table_matches_the_engine_layout():
    """tests/golden/ckpt124_index.json = the tensor table of the reference's ckpt-124.index: every model variable the
    engine holds exists there with the same shape, in the same K order, and nothing else is trainable."""
    idx = json.load(open(GOLD))
    layers, total = layer_table()
    n = 0
    for (k, name, key), L in zip(tfckpt.model_variable_keys(len(layers)), [L for L in layers for _ in range(3)]):
        want = {"g": [L.cout], "v": list(L.vshape), "bias": [L.cout]}[name]
        assert idx[key]["shape"] == want and idx[key]["dtype"] == 1, key
        n += int(np.prod(want))
    assert n == total == 535267
    assert "model/layer_with_weights-44/g/.ATTRIBUTES/VARIABLE_VALUE" not in idx
    slots = [k for k in idx if ".OPTIMIZER_SLOT/optimizer/" in k]
    assert len(slots) == 2 * 132                                      # Nadam m and v for each of the 132 trainables
    assert "optimizer/momentum_cache/.ATTRIBUTES/VARIABLE_VALUE" in idx and idx[""]["num_shards"] == 2


@pytest.mark.skipif(not os.path.exists(REF + ".index"), reason="reference checkout not present on this machine")
def test_reader_parses_the_reference_index_file():
    idx = tfckpt.read_index(REF)
    gold = json.load(open(GOLD))
    assert set(idx) == set(gold)
    for k, v in gold.items():
        if k:
            assert (idx[k]["dtype"], list(idx[k]["shape"]), idx[k]["shard"], idx[k]["offset"], idx[k]["size"]) == \
                   (v["dtype"], v["shape"], v["shard"], v["offset"], v["size"]), k
    assert int(tfckpt.read_tensor(REF, idx, "step/.ATTRIBUTES/VARIABLE_VALUE")) > 0        # shard 0 ships with the repo
    model = WDSRConv3D("t", "NIR", 8075.2045, 3160.7272, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True, seed=0)
    with pytest.raises(FileNotFoundError, match="missing"):          # the weight shard is a missing large blob
        tfckpt.load_reference_checkpoint(model, REF)


def test_bundle_round_trip_and_load(tmp_path):
    a = WDSRConv3D("a", "NIR", 1.0, 2.0, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True, seed=11)
    with torch.no_grad():
        a.flat.add_(torch.randn_like(a.flat) * 0.01)
    prefix = str(tmp_path / "ckpt-7")
    tfckpt.save_reference_checkpoint(a, prefix, step=1234, psnr=48.5)
    idx = tfckpt.read_index(prefix)
    assert idx[""]["num_shards"] == 1 and len([k for k in idx if k.endswith("/v/.ATTRIBUTES/VARIABLE_VALUE")]) == 44
    np.testing.assert_array_equal(tfckpt.read_tensor(prefix, idx, "model/layer_with_weights-3/v/.ATTRIBUTES/VARIABLE_VALUE"),
                                  a.trainable_variables[3 * 3 + 1].detach().numpy())
    b = WDSRConv3D("b", "NIR", 1.0, 2.0, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True, seed=12)
    assert tfckpt.load_reference_checkpoint(b, prefix) == 1234
    assert torch.equal(a.flat.detach(), b.flat.detach())
    # corruption is detected (crc32c per tensor), wrong architecture is refused
    with open(tfckpt.shard_path(prefix, 0, 1), "r+b") as fh:
        fh.seek(100); fh.write(b"\xff\xff\xff\xff")
    with pytest.raises(ValueError, match="crc32c"):
        tfckpt.load_reference_checkpoint(b, prefix)
    c = WDSRConv3D("c", "NIR", 1.0, 2.0, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 13, 16, True, seed=13)
    with pytest.raises((KeyError, ValueError)):
        tfckpt.load_reference_checkpoint(c, prefix)


def test_crc32c_known_answers():
    assert tfckpt.crc32c(b"") == 0 and tfckpt.crc32c(b"123456789") == 0xE3069283      # RFC 3720 check value
    assert tfckpt.crc32c(bytes(32)) == 0x8A9136AA


def test_trainer_restores_from_a_reference_format_directory(tmp_path):
    """A checkpoint directory as the reference writes it (TF CheckpointState file + bundle) is restored by ModelTrainer."""
    from probav_amd.trainClass import ModelTrainer
    a = WDSRConv3D("a", "NIR", 1.0, 2.0, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True, seed=21)
    ck = tmp_path / "ckpt_p16t9c85r12" / "NIR"
    ck.mkdir(parents=True)
    tfckpt.save_reference_checkpoint(a, str(ck / "ckpt-124"), step=184532)
    (ck / "checkpoint").write_text('model_checkpoint_path: "ckpt-124"\nall_model_checkpoint_paths: "ckpt-123"\nall_model_checkpoint_paths: "ckpt-124"\n')
    b = WDSRConv3D("b", "NIR", 1.0, 2.0, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True, seed=22)
    tr = ModelTrainer(b, None, None, None, str(ck), str(tmp_path / "logs"))
    assert tr.step == 184532 and torch.equal(a.flat.detach(), b.flat.detach())


def test_trainer_restores_nadam_slots_and_leaves_the_tf_state_file_alone(tmp_path):
    """a reference bundle carries Nadam's m / v slots, `optimizer/iter` and `momentum_cache`; the trainer resumes with them
    (instead of t = 0 and zero moments), and its own saves do not overwrite TensorFlow's `checkpoint` state file."""
    from probav_amd.trainClass import ModelTrainer, make_optimizer
    a = WDSRConv3D("a", "NIR", 1.0, 2.0, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True, seed=31)
    rng = np.random.default_rng(3)
    n = a.flat.numel()
    opt_state = {"iter": 4321, "momentum_cache": 0.0123, "m": rng.normal(size=n).astype(np.float32) * 1e-3,
                 "v": (rng.normal(size=n).astype(np.float32) * 1e-3) ** 2}
    ck = tmp_path / "ck"
    ck.mkdir()
    tfckpt.save_reference_checkpoint(a, str(ck / "ckpt-9"), step=77, psnr=47.25, optimizer=opt_state)
    tf_state = 'model_checkpoint_path: "ckpt-9"\nall_model_checkpoint_paths: "ckpt-9"\n'
    (ck / "checkpoint").write_text(tf_state)
    got = tfckpt.load_reference_optimizer(a, str(ck / "ckpt-9"))
    assert got["iter"] == 4321 and abs(got["momentum_cache"] - 0.0123) < 1e-7 and got["psnr"] == 47.25
    np.testing.assert_array_equal(got["m"], opt_state["m"])
    np.testing.assert_array_equal(got["v"], opt_state["v"])
    b = WDSRConv3D("b", "NIR", 1.0, 2.0, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True, seed=32)
    opt = make_optimizer("nadam", b, 5e-4)                           # HipNadam (its state loads on any device; only step() needs the GPU)
    tr = ModelTrainer(b, None, None, opt, str(ck), str(tmp_path / "logs"))
    assert tr.step == 77 and tr.psnr == 47.25
    st = opt.state[b.flat]
    assert float(st["step"]) == 4321 and abs(float(st["momentum_cache"]) - 0.0123) < 1e-7
    np.testing.assert_array_equal(st["m"].numpy(), opt_state["m"])
    np.testing.assert_array_equal(st["v"].numpy(), opt_state["v"])
    tr.save()
    assert (ck / "checkpoint").read_text() == tf_state               # still the reference's own state file
    assert (ck / "checkpoint.pt-index").read_text().split() == ["ckpt-1.pt"]
    # a later restore prefers this trainer's newer .pt checkpoint over the TF bundle
    c = WDSRConv3D("c", "NIR", 1.0, 2.0, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True, seed=33)
    tr2 = ModelTrainer(c, None, None, make_optimizer("nadam", c, 5e-4), str(ck), str(tmp_path / "logs2"))
    assert tr2.step == 77 and torch.equal(c.flat.detach(), b.flat.detach())
    # weights-only bundle: no slots -> None
    tfckpt.save_reference_checkpoint(a, str(tmp_path / "w-1"), step=1)
    assert tfckpt.load_reference_optimizer(a, str(tmp_path / "w-1")) is None
