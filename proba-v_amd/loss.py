"""Host-side mirror of the reference's models/loss.py::Losses (shift-compensated L1 / L2 / cPSNR).

Argument order is the reference's: ``(patchHR, maskHR, predPatchHR)`` (models/loss.py:37,55,73).  One fused
HIP launch evaluates all (2*cropBorder+1)^2 candidate registrations of every sample and returns the
per-sample minima of the L1 and L2 terms, the cPSNR maximum and the arg-min shifts; the backward
launch differentiates the arg-min shift including the brightness-bias term.  Both are
torch custom ops (`torch.ops.probav.shift_loss` with its registered autograd formula, `probav.shift_metrics`).  The
reference unrolls the same work into ~600 TensorFlow ops per step (models/loss.py:79-81).
"""
import torch

from . import _lib, ops             # noqa: F401  (ops registers torch.ops.probav.*)


def _prep(patchHR, maskHR, predPatchHR):
    pred = _lib.require_device(predPatchHR, "predPatchHR")
    dev = pred.device
    hr = torch.as_tensor(patchHR).to(device=dev, dtype=torch.float32).contiguous()
    m = torch.as_tensor(maskHR).to(device=dev)
    m = (m if m.dtype == torch.bool else m != 0).contiguous().view(torch.uint8)   # True = clear pixel (train.py:43)
    pred = pred.contiguous().float()
    if hr.shape != pred.shape or m.shape != pred.shape or pred.dim() != 4 or pred.shape[3] != 1 or pred.shape[1] != pred.shape[2]:
        raise ValueError("expected patchHR, maskHR, predPatchHR all [B, S, S, 1]; got %s %s %s"
                         % (tuple(hr.shape), tuple(m.shape), tuple(pred.shape)))
    return hr, m, pred


def _launch_forward(hr, m, pred, border, bit_depth):
    """One launch for all shifts: torch.ops.probav.shift_metrics -> (f [3,B] = l1 | l2 | cpsnr, arg [2,B], means [2])."""
    return torch.ops.probav.shift_metrics(hr, m, pred, border, bit_depth)


class _ShiftL1Edge(torch.autograd.Function):
    """cfg loss = sobel_l1_mix (models/loss.py:86-97): one launch over all shifts, one for the gradient of the arg-min shift."""

    @staticmethod
    def forward(ctx, pred, hr, m, border, pi):
        B, S = pred.shape[0], pred.shape[1]
        loss = torch.empty(B, dtype=torch.float32, device=pred.device)
        arg = torch.empty(B, dtype=torch.int32, device=pred.device)
        mean = torch.empty(2, dtype=torch.float32, device=pred.device)
        _lib.check(_lib.lib().probav_shift_l1edge_forward(_lib.ptr(hr), _lib.ptr(m), _lib.ptr(pred), B, S, border, pi, _lib.ptr(loss),
                                                          _lib.ptr(arg), _lib.ptr(mean), _lib.current_stream()), "probav_shift_l1edge_forward")
        ctx.save_for_backward(pred, hr, m, arg)
        ctx.border, ctx.pi = border, pi
        ctx.per_sample = loss
        return mean[0].clone()

    @staticmethod
    def backward(ctx, g):
        pred, hr, m, arg = ctx.saved_tensors
        g = g.contiguous().float().reshape(1)
        dpred = torch.empty_like(pred)
        _lib.check(_lib.lib().probav_shift_l1edge_backward(_lib.ptr(hr), _lib.ptr(m), _lib.ptr(pred), _lib.ptr(arg), pred.shape[0], pred.shape[1],
                                                           ctx.border, ctx.pi, _lib.ptr(g), _lib.ptr(dpred), _lib.current_stream()),
                   "probav_shift_l1edge_backward")
        return dpred, None, None, None, None


class _RevSSIM(torch.autograd.Function):
    """cfg loss = l1msssim (models/loss.py:99-124, 189-212): moments per (shift, sample, scale), one scalar per shift for the whole
    batch, minimum over the shifts; the backward differentiates the arg-min shift."""

    @staticmethod
    def forward(ctx, pred, hr, m, border, bit_depth, eta):
        B, S = pred.shape[0], pred.shape[1]
        nbytes = _lib.lib().probav_revssim_scratch_bytes(B, border)
        scratch = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device=pred.device)
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        arg = torch.empty(1, dtype=torch.int32, device=pred.device)
        _lib.check(_lib.lib().probav_revssim_forward(_lib.ptr(hr), _lib.ptr(m), _lib.ptr(pred), B, S, border, bit_depth, eta, _lib.ptr(scratch),
                                                     nbytes, _lib.ptr(loss), _lib.ptr(arg), _lib.current_stream()), "probav_revssim_forward")
        ctx.save_for_backward(pred, hr, m, arg, scratch)
        ctx.border, ctx.bit_depth, ctx.eta = border, bit_depth, eta
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        pred, hr, m, arg, scratch = ctx.saved_tensors
        g = g.contiguous().float().reshape(1)
        dpred = torch.empty_like(pred)
        _lib.check(_lib.lib().probav_revssim_backward(_lib.ptr(hr), _lib.ptr(m), _lib.ptr(pred), _lib.ptr(arg), _lib.ptr(scratch), pred.shape[0],
                                                      pred.shape[1], ctx.border, ctx.bit_depth, ctx.eta, _lib.ptr(g), _lib.ptr(dpred),
                                                      _lib.current_stream()), "probav_revssim_backward")
        return dpred, None, None, None, None, None


class Losses:
    """models/loss.py:8-35: all losses / metrics in one object; constants follow the reference."""

    def __init__(self, targetShape=(96, 96, 1), cropBorder=3, bitDepth=16):
        self.targetShapeHeight, self.targetShapeWidth, self.targetShapeChannels = targetShape
        self.cropBorder = cropBorder
        self.maxPixelShift = 2 * cropBorder
        self.bitDepth = bitDepth
        self.numBytes = 2 ** bitDepth - 1
        self.cropSizeHeight = self.targetShapeHeight - self.maxPixelShift
        self.cropSizeWidth = self.targetShapeWidth - self.maxPixelShift
        self.pi = 0.7                                     # SobelL1Mix weight (models/loss.py:21)
        self.eta = 0.25                                   # SSIM share of the l1msssim mixture (models/loss.py:35)

    def _check(self, pred):
        if pred.shape[1] != self.targetShapeHeight or pred.shape[2] != self.targetShapeWidth:
            raise ValueError("Losses(targetShape=%r) got a %dx%d prediction"
                             % ((self.targetShapeHeight, self.targetShapeWidth, self.targetShapeChannels),
                                pred.shape[1], pred.shape[2]))

    def shiftCompensatedL1Loss(self, patchHR, maskHR, predPatchHR):
        """models/loss.py:73-84 -> scalar: mean over the batch of the minimum masked, bias-corrected L1."""
        hr, m, pred = _prep(patchHR, maskHR, predPatchHR)
        self._check(pred)
        return torch.ops.probav.shift_loss(pred, hr, m, self.cropBorder, self.bitDepth, 1)[0]

    def shiftCompensatedL2Loss(self, patchHR, maskHR, predPatchHR):
        """models/loss.py:55-71."""
        hr, m, pred = _prep(patchHR, maskHR, predPatchHR)
        self._check(pred)
        return torch.ops.probav.shift_loss(pred, hr, m, self.cropBorder, self.bitDepth, 2)[0]

    def shiftCompensatedcPSNR(self, patchHR, maskHR, predPatchHR):
        """models/loss.py:37-53 -> [B]: maximum cPSNR over the shifts (no gradient, as in trainStep)."""
        hr, m, pred = _prep(patchHR, maskHR, predPatchHR)
        self._check(pred)
        with torch.no_grad():
            f, _, _ = _launch_forward(hr, m, pred.detach(), self.cropBorder, self.bitDepth)
        return f[2]

    def evaluate_all(self, patchHR, maskHR, predPatchHR):
        """One launch, everything it computes: dict(l1[B], l2[B], cpsnr[B], arg_l1[B], arg_l2[B], mean_l1, mean_l2)."""
        hr, m, pred = _prep(patchHR, maskHR, predPatchHR)
        with torch.no_grad():
            f, arg, means = _launch_forward(hr, m, pred.detach(), self.cropBorder, self.bitDepth)
        return {"l1": f[0], "l2": f[1], "cpsnr": f[2], "arg_l1": arg[0], "arg_l2": arg[1],
                "mean_l1": means[0], "mean_l2": means[1]}

    def shiftCompensatedL1EdgeLoss(self, patchHR, maskHR, predPatchHR):
        """models/loss.py:86-97 (cfg loss = sobel_l1_mix): pi * L1 + (1 - pi) * Sobel-edge L1, minimum over the shifts, batch mean."""
        hr, m, pred = _prep(patchHR, maskHR, predPatchHR)
        self._check(pred)
        return _ShiftL1Edge.apply(pred, hr, m, self.cropBorder, float(self.pi))

    def shiftCompensatedRevSSIM(self, patchHR, maskHR, predPatchHR):
        """models/loss.py:99-110 (cfg loss = l1msssim): the batch-level multi-scale SSIM / weighted-L1 mixture, minimum over the shifts."""
        hr, m, pred = _prep(patchHR, maskHR, predPatchHR)
        self._check(pred)
        return _RevSSIM.apply(pred, hr, m, self.cropBorder, self.bitDepth, float(self.eta))
