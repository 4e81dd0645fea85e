"""Host-side mirror of the reference's inference wrappers: models/testClass.py::Enhancer and the helpers of
test.py (``resolve``, ``resolveByBatch``, ``evaluate``, ``reconstruct_from_patches``; test.py:103-160).

Patch-wise evaluation is kept exactly as the reference does it -- the network is NOT run on the whole
frame, because its 'same'-padded convolutions see zeros beyond each 22x22 patch.
The forward pass, the clip to [0, 2**16] and the round-half-even run on the device (HIP kernels);
only the stitched uint16-range image returns to the host.
"""
import numpy as np
import torch

from . import _lib, ops            # noqa: F401  (ops registers torch.ops.probav.*)


def _device_of(model):
    return next(model.parameters()).device


def resolve_device(model, lr_batch):
    """test.py:114-122 without the final host copy: float32 cast -> model -> clip_by_value(0, 2**16) -> round."""
    dev = _device_of(model)
    x = torch.as_tensor(np.ascontiguousarray(lr_batch) if isinstance(lr_batch, np.ndarray) else lr_batch)
    x = x.to(device=dev, dtype=torch.float32)
    with torch.no_grad():
        sr = model(x, training=False)
        out = torch.ops.probav.clip_round(sr, 0.0, float(2 ** 16))
    return out


def resolve(model, lr_batch):
    """test.py:114-122 -> numpy float32 [b, 3P, 3P, 1]."""
    return resolve_device(model, lr_batch).cpu().numpy()


LAUNCH_BATCH = 2048        # patches per launch set of the coalesced paths (5.0 GB of workspace at T = 9)


def _is_engine_model(model):
    return hasattr(model, "_handle") and hasattr(model, "flat")


def resolveByBatch(model, lr_batch, batch_size=16):
    """test.py:125-134: micro-batches of `batch_size` plus the remainder, concatenated.

    The reference slices because its GPU could not hold more; the slices are invisible in the result (they are concatenated in order, and the
    network has no cross-sample term: models/modelsTF.py:15-43).  On the engine every kernel family is bitwise independent of the batch
    (tests/test_gpu_h3_range.py::test_forward_is_bitwise_independent_of_the_batch, tests/test_gpu_parity.py::test_config4_*), so the
    micro-batches are coalesced into launch sets of up to LAUNCH_BATCH patches and the result is the same array, bit for bit, at the
    batched rate (a 16-patch launch set fills a sixteenth of the device).  Any other callable takes the reference's loop as written
    (tests/test_ref_plumbing.py holds its call pattern to the reference's own function)."""
    if _is_engine_model(model) and resolve is _RESOLVE:
        return resolve_coalesced(model, lr_batch, batch_size).cpu().numpy()
    n, rem = divmod(lr_batch.shape[0], batch_size)
    cache = [resolve(model, lr_batch[batch_size * i: batch_size * (i + 1)]) for i in range(n)]
    if rem:
        cache.append(resolve(model, lr_batch[batch_size * n: batch_size * n + rem]))
    return np.concatenate(cache)


_RESOLVE = resolve


def resolve_coalesced(model, lr_batch, batch_size=16, launch_batch=None):
    """`resolveByBatch` on the device: the reference's micro-batches of `batch_size` grouped into launch sets of whole micro-batches
    (at most `launch_batch` patches, default LAUNCH_BATCH); returns the device tensor [n, 3P, 3P, 1]."""
    launch_batch = LAUNCH_BATCH if launch_batch is None else launch_batch
    per = max(1, launch_batch // max(1, batch_size)) * max(1, batch_size)          # whole micro-batches per launch set
    n = lr_batch.shape[0]
    outs = [resolve_device(model, lr_batch[i:i + per]) for i in range(0, n, per)]
    return outs[0] if len(outs) == 1 else torch.cat(outs)


def resolveBySampleAveraging(model, lr_batch, rng=None):
    """test.py:137-146: the mean over 20 predictions, each on a cumulative random permutation of the LR frames (axis 3); every
    prediction is clipped and rounded (it goes through `resolve`) before the mean.  `rng`: numpy Generator (the reference draws
    from the global numpy state).  Returns a device tensor [b, 3P, 3P, 1]."""
    rng = np.random.default_rng() if rng is None else rng
    x = torch.as_tensor(np.ascontiguousarray(lr_batch) if isinstance(lr_batch, np.ndarray) else lr_batch).to(_device_of(model))
    acc = None
    for _ in range(20):
        idx = torch.as_tensor(rng.permutation(x.shape[3])).to(x.device)
        x = x.index_select(3, idx)                            # the permutations compound, as in the reference
        sr = resolve_device(model, x.contiguous())
        acc = sr.double() if acc is None else acc + sr.double()
    return (acc / 20.0).float()


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
    """test.py:103-111: one stitched prediction per image set.  On the engine the image sets and their micro-batches are coalesced
    (see `resolveByBatch`): same pixels, one copy back."""
    if _is_engine_model(model) and resolve is _RESOLVE:
        return evaluate_device(model, X_test_patches, micro_batch=batch_size)
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


# ---------------------------------------------------------------------------------------------------------------------
# Device-side image pipeline around the forward kernels.  Tensor re-arrangement only (torch plumbing);
# the arithmetic stays in the HIP engine and in probav_clip_round.
# ---------------------------------------------------------------------------------------------------------------------
def unfold_frames(frames, patchSizeLR=16, maxShift=6):
    """Registered LR frames [sets, T, H, W] (H = W = 128) -> patches [sets, n*n, P+s, P+s, T, 1] on the frames' device.

    Restates what the reference's offline preprocessing does before test.py sees the data
    (utils/dataGenerator.py:108-121: reflect-pad every frame by maxShift//2, then unfold (P+maxShift)-sized windows with
    stride P, row-major) followed by test.py:38's transpose to [sets, patch, H, W, T, C]."""
    import torch.nn.functional as F
    f = torch.as_tensor(frames)
    S, T, H, W = f.shape
    pad, win = maxShift // 2, patchSizeLR + maxShift
    fp = F.pad(f.reshape(S * T, 1, H, W).float(), (pad, pad, pad, pad), mode="reflect")
    u = fp.unfold(2, win, patchSizeLR).unfold(3, win, patchSizeLR)           # [S*T, 1, n, n, win, win]
    n = u.shape[2]
    u = u.reshape(S, T, n * n, win, win).permute(0, 2, 3, 4, 1).contiguous()  # [S, n*n, win, win, T]
    return u.unsqueeze(-1)


def stitch_device(sr, sets):
    """[sets*n*n, ps, ps, 1] -> [sets, n*ps, n*ps]: the row-major block layout of test.py:149-160, on the device."""
    nn_, ps = sr.shape[0] // sets, sr.shape[1]
    n = int(round(nn_ ** 0.5))
    return sr.reshape(sets, n, n, ps, ps).permute(0, 1, 3, 2, 4).reshape(sets, n * ps, n * ps)


def resolve_images(model, patches, micro_batch=2048, launch_batch=None):
    """All image sets at once: patches [sets, n*n, P+s, P+s, T, 1] -> uint16-range images [sets, 3nP, 3nP] (device tensor).
    Samples are independent in every kernel family (models/modelsTF.py:15-43 has no cross-sample term; the H3 kernels scale their
    operands per sample), so any micro-batch gives bit-identical pixels to the reference's batches of 16
    (tests/test_gpu_h3_range.py::test_forward_is_bitwise_independent_of_the_batch).
    `micro_batch` is the reference-visible slicing (test.py:125: 16); `launch_batch` is how many patches one launch set of the engine takes:
    None (default) coalesces whole micro-batches up to max(micro_batch, LAUNCH_BATCH); `launch_batch=micro_batch` launches every micro-batch
    on its own, as the reference's loop does (the parity tests compare the two bit for bit)."""
    dev = _device_of(model)
    p = torch.as_tensor(patches)
    sets = p.shape[0]
    flat = p.reshape((-1,) + tuple(p.shape[2:]))
    if launch_batch is None:
        launch_batch = max(micro_batch, LAUNCH_BATCH)
    per = max(1, launch_batch // max(1, micro_batch)) * max(1, micro_batch)
    outs = []
    for i in range(0, flat.shape[0], per):
        outs.append(resolve_device(model, flat[i:i + per].to(dev)))
    return stitch_device(torch.cat(outs) if len(outs) > 1 else outs[0], sets)


def evaluate_device(model, X_test_patches, micro_batch=2048, launch_batch=None):
    """test.py:103-111 through the device pipeline: every image set in micro-batches of `micro_batch` patches (16 = the reference's
    resolveByBatch; coalesced into launch sets unless `launch_batch` says otherwise), clip / round and the 8 x 8 stitch on the device, ONE
    copy back.  Returns a list of [384, 384, 1] float64 arrays, element for element what the reference's `evaluate` returns."""
    imgs = resolve_images(model, X_test_patches, micro_batch=micro_batch, launch_batch=launch_batch).cpu().numpy().astype(np.float64)
    return [im[:, :, None] for im in imgs]
