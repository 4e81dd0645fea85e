"""Host-side mirror of the reference's inference wrappers: models/testClass.py::Enhancer and the helpers of
test.py (``resolve``, ``resolveByBatch``, ``evaluate``, ``reconstruct_from_patches``; test.py:103-160).

Patch-wise evaluation is kept exactly as the reference does it -- the network is NOT run on the whole
frame, because its 'same'-padded convolutions see zeros beyond each 22x22 patch.
The forward pass, the clip to [0, 2**16] and the round-half-even run on the device (HIP kernels);
only the stitched uint16-range image returns to the host.
"""
import numpy as np
import torch

from . import _lib


def _device_of(model):
    return next(model.parameters()).device


def resolve_device(model, lr_batch):
    """test.py:114-122 without the final host copy: float32 cast -> model -> clip_by_value(0, 2**16) -> round."""
    dev = _device_of(model)
    x = torch.as_tensor(np.ascontiguousarray(lr_batch) if isinstance(lr_batch, np.ndarray) else lr_batch)
    x = x.to(device=dev, dtype=torch.float32)
    with torch.no_grad():
        sr = model(x, training=False)
        out = torch.empty_like(sr)
        _lib.check(_lib.lib().probav_clip_round(_lib.ptr(sr), _lib.ptr(out), sr.numel(), 0.0, float(2 ** 16),
                                                _lib.current_stream()), "probav_clip_round")
    return out


def resolve(model, lr_batch):
    """test.py:114-122 -> numpy float32 [b, 3P, 3P, 1]."""
    return resolve_device(model, lr_batch).cpu().numpy()


def resolveByBatch(model, lr_batch, batch_size=16):
    """test.py:125-134: micro-batches of `batch_size` plus the remainder, concatenated."""
    n, rem = divmod(lr_batch.shape[0], batch_size)
    cache = [resolve(model, lr_batch[batch_size * i: batch_size * (i + 1)]) for i in range(n)]
    if rem:
        cache.append(resolve(model, lr_batch[batch_size * n: batch_size * n + rem]))
    return np.concatenate(cache)


def reconstruct_from_patches(images):
    """test.py:149-160: row-major n x n stitch of square patches into a [384, 384, 1] image
    (float64 zeros, like np.zeros in the reference)."""
    rec = np.zeros((384, 384, 1))
    n = int(len(images) ** 0.5)
    ps = images.shape[1]
    k = 0
    for i in range(n):
        for j in range(n):
            rec[i * ps:(i + 1) * ps, j * ps:(j + 1) * ps] = images[k]
            k += 1
    return rec.reshape((384, 384, 1))


def evaluate(model, X_test_patches, batch_size=16):
    """test.py:103-111: one stitched prediction per image set."""
    return [reconstruct_from_patches(resolveByBatch(model, X_test_patches[i], batch_size))
            for i in range(X_test_patches.shape[0])]


class Enhancer:
    """models/testClass.py:11-39 (unused by the reference's own test.py; kept for API parity,
    including its hard-coded 4 x 4 grid of 96-pixel blocks)."""

    def __init__(self, model, patchLR):
        self.model = model
        self.patchLR = patchLR

    def enhance(self):
        return [self.reconstruct(np.array(self.enhancePatch(s).cpu())) for s in self.patchLR]

    def enhancePatch(self, set):
        return resolve_device(self.model, set)

    def reconstruct(self, patches):
        img = np.zeros((384, 384, 1))
        k = 0
        for i in range(4):
            for j in range(4):
                img[i * 96:(i + 1) * 96, j * 96:(j + 1) * 96] = patches[k]
                k += 1
        return img.reshape((384, 384, 1))
