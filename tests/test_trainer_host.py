"""Host logic of ModelTrainer / the input pipeline on CPU with the HIP forward swapped for a tiny torch
stand-in (test-only: it exercises step counting, evaluation cadence, checkpoint rotation and restore,
not the arithmetic of the hot path)."""
import json
import os

import numpy as np
import pytest
import torch

from probav_amd.modelsTF import WDSRModel
from probav_amd.trainClass import HipNadam, Mean, ModelTrainer, make_optimizer, shuffle_repeat_batch


class StubModel(WDSRModel):
    def forward(self, x, training=False):
        k = self.flat[:48 * 48].view(1, 48, 48, 1)
        return x.mean(dim=(1, 2, 3, 4)).view(-1, 1, 1, 1) * 0 + k * 1e-3 + 5.0


def _stub():
    return StubModel("stub", "NIR", 0.0, 1.0, 6, 3, 32, 12, 8, 0.8, 9, 16, seed=0)


def cpu_optimizer(name, model, lr):
    """Test-only stand-in for make_optimizer: the product's optimizers are HIP launches and refuse CPU parameters, the host-logic
    tests step the stub model with the algebraically identical torch.optim classes."""
    params = list(model.parameters())
    if name == "adam":
        return torch.optim.Adam(params, lr=lr, betas=(0.9, 0.999), eps=1e-7)
    if name == "nadam":
        return torch.optim.NAdam(params, lr=lr, betas=(0.9, 0.999), eps=1e-7, momentum_decay=0.004)
    return torch.optim.SGD(params, lr=lr)


def _l1(hr, mask, pred):
    return ((hr - pred).abs() * mask).mean()


def _metric(hr, mask, pred):
    return -((hr - pred) ** 2).mean(dim=(1, 2, 3))


def test_shuffle_repeat_batch_semantics():
    rng = np.random.default_rng(0)
    batches = list(shuffle_repeat_batch(10, 3, 4, 5, rng))
    flat = np.concatenate(batches)
    assert len(flat) == 30 and [len(b) for b in batches] == [4] * 7 + [2]
    for e in range(3):                                   # every epoch is a permutation (repeat AFTER shuffle)
        assert sorted(flat[10 * e:10 * (e + 1)]) == list(range(10))
    assert any(not np.array_equal(flat[:10], flat[10 * e:10 * (e + 1)]) for e in (1, 2))   # reshuffled each iteration
    first = np.concatenate(list(shuffle_repeat_batch(100, 1, 100, 5, np.random.default_rng(1))))
    assert max(first[:3]) < 5 + 3                        # a 5-element buffer can only emit early indices first


def test_mean_metric():
    m = Mean()
    m(torch.tensor([1.0, 3.0]))
    m(torch.tensor(5.0))
    assert m.result() == 3.0
    m.reset_states()
    assert m.result() == 0.0


def test_fit_counts_steps_evaluates_and_rotates_checkpoints(tmp_path):
    model = _stub()
    hip = make_optimizer("nadam", model, 5e-4)               # the product's optimizer: Keras defaults, and no CPU arithmetic behind it
    assert isinstance(hip, HipNadam) and hip.defaults["epsilon"] == 1e-7 and hip.defaults["schedule_decay"] == 0.004
    model.flat.grad = torch.zeros_like(model.flat)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hip.step()
    model.flat.grad = None
    opt = cpu_optimizer("nadam", model, 5e-4)
    tr = ModelTrainer(model, _l1, _metric, opt, str(tmp_path / "ckpt"), str(tmp_path / "logs"), evalStep=2)
    n = 16
    X = np.zeros((n, 22, 22, 9, 1), np.float32)
    y = np.ones((n, 48, 48, 1), np.float32)
    msk = np.ones((n, 48, 48, 1), bool)
    before = model.flat.detach().clone()
    tr.fitTrainData(X, [y, msk], 4, 4, [X[:8], y[:8], msk[:8]], valSteps=1, saveBestOnly=False)
    assert tr.step == 16                                  # 16 samples / batch 4 * 4 epochs
    assert not torch.equal(before, model.flat.detach())
    names = open(tmp_path / "ckpt" / "checkpoint.pt-index").read().split()
    assert names == ["ckpt-%d.pt" % k for k in range(4, 9)]          # 8 saves, max_to_keep = 5
    assert sorted(os.listdir(tmp_path / "ckpt")) == sorted(names + ["checkpoint.pt-index"])      # TF's `checkpoint` state file is never written
    events = [json.loads(l) for l in open(tmp_path / "logs" / "events.jsonl")]
    tags = [e["tag"] for e in events]
    assert tags.count("Train loss") == 16 and tags.count("Test PSNR") == 8
    # the per-step scalars are emitted late (no host sync per step) but complete, in order, and before the evaluation that follows them
    assert [e["step"] for e in events if e["tag"] == "Train loss"] == list(range(1, 17))
    assert tags.index("Test loss") > [i for i, e in enumerate(events) if e["tag"] == "Train loss" and e["step"] == 2][0]
    # restore picks up step, psnr and the weights
    model2 = _stub()
    tr2 = ModelTrainer(model2, _l1, _metric, cpu_optimizer("adam", model2, 1e-3), str(tmp_path / "ckpt"), str(tmp_path / "logs2"))
    assert tr2.step == 16 and torch.equal(model2.flat.detach(), model.flat.detach())
    # saveBestOnly keeps the checkpoint only when the validation metric improves (models/trainClass.py:117-122)
    tr3 = ModelTrainer(_stub(), _l1, _metric, None, str(tmp_path / "c3"), str(tmp_path / "l3"), evalStep=1)
    tr3.optimizer = cpu_optimizer("sgd", tr3.model, 0.0)
    tr3.psnr = 1e9
    tr3.fitTrainData(X, [y, msk], 8, 1, [X[:8], y[:8], msk[:8]], valSteps=1, saveBestOnly=True)
    assert not os.path.exists(tmp_path / "c3" / "checkpoint.pt-index")


def test_batch_prefetcher_preserves_order_and_values():
    """trainClass.BatchPrefetcher (the pipeline's .prefetch): same batches, same order as the plain index stream."""
    from probav_amd.trainClass import BatchPrefetcher, shuffle_repeat_batch
    rng = np.random.default_rng(0)
    X = rng.normal(size=(37, 4, 3)).astype(np.float32)
    Y = rng.integers(0, 2, size=(37, 5)).astype(bool)
    idx = list(shuffle_repeat_batch(37, 2, 8, 16, np.random.default_rng(5)))
    got = list(BatchPrefetcher((X, Y), (torch.float32, torch.bool), iter(idx), "cpu", depth=2))
    assert len(got) == len(idx)
    for (xb, yb), i in zip(got, idx):
        np.testing.assert_array_equal(xb.numpy(), X[i])
        np.testing.assert_array_equal(yb.numpy(), Y[i])

    def boom():
        yield np.arange(4)
        raise RuntimeError("index stream failed")
    with pytest.raises(RuntimeError):
        list(BatchPrefetcher((X,), (torch.float32,), boom(), "cpu"))


def test_legacy_checkpoint_directories_are_recognised(tmp_path):
    """the .pt index moved from `checkpoint` to `checkpoint.pt-index`.  A directory of the earlier revision (index lines
    `ckpt-N.pt` in `checkpoint`), or one with .pt files and no index at all, restores its newest checkpoint, continues the numbering
    and keeps pruning -- instead of starting from scratch and overwriting ckpt-1.pt."""
    model = _stub()
    tr = ModelTrainer(model, _l1, _metric, cpu_optimizer("sgd", model, 0.1), str(tmp_path / "ck"), str(tmp_path / "lg"))
    with torch.no_grad():
        model.flat.add_(1.0)
    tr.step = 7
    for _ in range(3):
        tr.save()
    idx = tmp_path / "ck" / "checkpoint.pt-index"
    names = idx.read_text().split()
    assert names == ["ckpt-1.pt", "ckpt-2.pt", "ckpt-3.pt"]
    # (a) earlier revision: the index lived in `checkpoint`
    (tmp_path / "ck" / "checkpoint").write_text(idx.read_text())
    idx.unlink()
    m2 = _stub()
    tr2 = ModelTrainer(m2, _l1, _metric, cpu_optimizer("sgd", m2, 0.1), str(tmp_path / "ck"), str(tmp_path / "lg2"))
    assert tr2.step == 7 and tr2.save_counter == 3 and torch.equal(m2.flat.detach(), model.flat.detach())
    assert tr2.save() == "ckpt-4.pt"
    assert idx.read_text().split() == ["ckpt-1.pt", "ckpt-2.pt", "ckpt-3.pt", "ckpt-4.pt"]
    # (b) no index at all: the files themselves, by number
    idx.unlink()
    (tmp_path / "ck" / "checkpoint").unlink()
    m3 = _stub()
    tr3 = ModelTrainer(m3, _l1, _metric, cpu_optimizer("sgd", m3, 0.1), str(tmp_path / "ck"), str(tmp_path / "lg3"))
    assert tr3.step == 7 and tr3.save_counter == 4
    tr3.save(); tr3.save()
    assert idx.read_text().split() == ["ckpt-2.pt", "ckpt-3.pt", "ckpt-4.pt", "ckpt-5.pt", "ckpt-6.pt"]
    assert not (tmp_path / "ck" / "ckpt-1.pt").exists()
