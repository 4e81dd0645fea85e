"""The tensor plumbing around the kernels against fixtures the REFERENCE'S OWN CODE produced (tests/golden/make_ref_fixtures.py lifted
the functions out of /root/reference with `ast` and executed them in the build container; the fixture holds inputs and outputs only):
test.py:149-160 reconstruct_from_patches, test.py:125-134 resolveByBatch, models/testClass.py:31-39 Enhancer.reconstruct,
utils/dataGenerator.py:106-121 + 553-596 (reflect pad, generatePatches, reshape) followed by test.py:38's transpose.
This pins the data movement; the arithmetic of the network stays "parity unpinned" (oracle/__init__.py)."""
import os

import numpy as np
import torch

from probav_amd import testClass

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_plumbing.npz"))


def _ids(name):
    return np.cumsum(Z[name + "_diff"].astype(np.int64)).reshape(tuple(Z[name + "_shape"]))


def test_fixture_names_the_reference_lines_it_executed():
    lines = set(Z["reference_lines"].tolist())
    assert {"test.py::reconstruct_from_patches:149-160", "test.py::resolveByBatch:125-134", "models/testClass.py::Enhancer.reconstruct:32-39",
            "utils/dataGenerator.py::generatePatchesPerImgSet:569-596", "test.py::main[transpose]:38-38"} <= lines, lines


def test_reconstruct_from_patches_and_device_stitch_match_the_reference():
    ids = np.arange(64 * 48 * 48, dtype=np.float64).reshape(64, 48, 48, 1)
    want = _ids("rec64_out")
    got = testClass.reconstruct_from_patches(ids)
    assert got.shape == want.shape == (384, 384, 1) and got.dtype == np.float64
    np.testing.assert_array_equal(got, want)
    # the device pipeline's stitch (a reshape / permute on whatever device the tensor lives on), several image sets at once
    both = torch.as_tensor(np.concatenate([ids, ids + 1e6]).astype(np.float32))
    st = testClass.stitch_device(both, 2).numpy()
    np.testing.assert_array_equal(st[0], want[..., 0])
    np.testing.assert_array_equal(st[1], want[..., 0] + 1e6)
    # 16 patches of 96 pixels: the reference's function derives the grid from the count, Enhancer.reconstruct hard-codes it
    ids16 = np.arange(16 * 96 * 96, dtype=np.float64).reshape(16, 96, 96, 1)
    np.testing.assert_array_equal(testClass.reconstruct_from_patches(ids16), _ids("rec16_out"))
    np.testing.assert_array_equal(testClass.Enhancer(None, None).reconstruct(ids16), _ids("enh_out"))
    np.testing.assert_array_equal(testClass.stitch_device(torch.as_tensor(ids16), 1).numpy()[0], _ids("enh_out")[..., 0])


def test_resolve_by_batch_slices_like_the_reference(monkeypatch):
    calls = []

    def stub_resolve(model, lr_batch):
        calls.append(int(lr_batch.shape[0]))
        return model(lr_batch)
    monkeypatch.setattr(testClass, "resolve", stub_resolve)
    model = lambda b: np.asarray(b)[:, :2, :2, 0, :] * 2.0 + 1.0
    for n, bs in Z["rbb_cases"].tolist():
        lr = np.arange(n * 3 * 3 * 2, dtype=np.float32).reshape(n, 3, 3, 2, 1)
        del calls[:]
        got = testClass.resolveByBatch(model, lr, batch_size=bs)
        assert calls == Z["rbb_%d_%d_calls" % (n, bs)].tolist(), (n, bs, calls)
        np.testing.assert_array_equal(got, Z["rbb_%d_%d_out" % (n, bs)])
    del calls[:]
    testClass.resolveByBatch(model, np.zeros((37, 3, 3, 2, 1), np.float32))
    assert calls == Z["rbb_default_calls"].tolist() == [16, 16, 5]


def test_unfold_frames_matches_the_reference_patch_generator():
    """utils/dataGenerator.py:106-121 (reflect pad by max_shift // 2, generatePatches with stride = patch_size, reshape) + test.py:38."""
    s, T, _, H, W = Z["unfold_frames_shape"].tolist()
    frames = np.arange(s * T * H * W, dtype=np.float32).reshape(s, T, H, W)
    got = testClass.unfold_frames(torch.as_tensor(frames)).numpy()
    want = _ids("unfold_patches")
    assert got.shape == want.shape == (2, 16, 22, 22, 9, 1)
    np.testing.assert_array_equal(got, want)
    frames128 = np.arange(3 * 128 * 128, dtype=np.float32).reshape(1, 3, 128, 128)
    want128 = _ids("unfold128_patches")
    assert want128.shape == (1, 64, 22, 22, 3, 1)
    np.testing.assert_array_equal(testClass.unfold_frames(torch.as_tensor(frames128)).numpy(), want128)
