"""H3 arithmetic (impl 4, the default kernel family) under INTRA-TENSOR dynamic range (lead item).

H3 evaluates an fp32 product as three products of fp16 piece pairs and therefore scales its operands by powers of two.  A
scale may vary along an operand's free index, never along the contracted one; the kernels scale per SAMPLE (activations and
gradients: a patch never meets another patch in a forward / backward-data contraction) and per OUTPUT COLUMN (filters).  These
tests put the dynamic range where a per-tensor scale would lose it and judge every output slice -- one sample, one output
channel -- against ITS OWN maximum, holding impl 4 to the bound the native fp32-MFMA kernels (impl 2) meet on the same inputs:

  * a near-zero sample (2^-30) next to a full-scale one, and a 2^20 outlier voxel inside one sample;
  * filter columns with log-uniform gains over 2^-30 .. 1;
  * channel gains on the CONTRACTED index (input channels): the small channels' contributions are small in every output, so the
    per-slice error stays at fp32 level although those inputs are represented coarsely;
  * the backward-FILTER products contract over the voxels of all samples: the slices are (cin, cout) pairs and the operands take one
    scale per tensor.  Rounds 2 - 4 documented a limit here (a channel 2^-24 below its tensor mates got a coarse gradient slice);
    since the end of round 5 the second pieces are stored lifted by 2^11 (conv3_wgrad_w4_kernel) and every slice is held to the bar
    (test_wgrad_resolves_every_channel_slice, test_reducer_wgrad_resolves_every_channel_slice); a layer that kernel has no instance
    for runs the unscaled x6 arithmetic (test_wgrad_without_an_instance_of_the_lifted_kernel_is_not_scaled).

The forward result of a sample does not depend on its batch mates, bit for bit (the reference's model(x) has no cross-sample
term, models/modelsTF.py:15-43): test_forward_is_bitwise_independent_of_the_batch."""
import ctypes
import zlib

import numpy as np
import pytest
import torch

from oracle import wdsr_numpy as on
from probav_amd import synth

pytestmark = pytest.mark.gpu


def _L():
    from probav_amd import _lib as L
    return L


def _geom(N, Hi, Wi, Ti, Cin, Ho, Wo, To, Cout, k, pad, reflect=0, relu=0):
    return (ctypes.c_int32 * 17)(N, Hi, Wi, Ti, Cin, Ho, Wo, To, Cout, k[0], k[1], k[2], pad[0], pad[1], pad[2], reflect, relu)


def _t(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(dev)


def _gains(rng, n, lo=-30):
    g = 2.0 ** rng.uniform(lo, 0, size=n)
    g[rng.integers(n)] = 1.0
    g[rng.integers(n)] = 2.0 ** lo
    return g.astype(np.float32)


def _slice_err(got, ref, axes):
    """max over slices of  max|got - ref| / max|ref|  with the maxima taken over `axes` (the axes INSIDE a slice)."""
    num = np.abs(got.astype(np.float64) - ref).max(axis=axes)
    den = np.abs(ref).max(axis=axes)
    return float((num / np.maximum(den, 1e-300)).max())


def _conv(dev, impl, g, x, w, bias=None, skip=None, gate=None):
    L = _L()
    N, ho, Cout = g[0], (g[5], g[6], g[7]), g[8]
    y = torch.full((N,) + ho + (Cout,), float("nan"), device=dev)
    args = [_t(a, dev) if a is not None else None for a in (x, gate, w, bias, skip)]
    L.check(L.lib().probav_conv3d_forward(ctypes.byref(g), *[L.ptr(a) for a in args], L.ptr(y), impl, L.current_stream()), "probav_conv3d_forward")
    return y.cpu().numpy()


def _oracle_conv(x, w, bias, skip, pad):
    xp = np.pad(np.asarray(x, np.float64), [(0, 0), (pad[0],) * 2, (pad[1],) * 2, (pad[2],) * 2, (0, 0)])
    y = on.conv_valid(xp, w)
    if bias is not None:
        y = y + np.asarray(bias, np.float64)
    if skip is not None:
        y = y + np.asarray(skip, np.float64)
    return y


# normConv forward (pstrip<25>), its backward-data (pstrip<32>), a reducer (row-tile kernel)
CASES = [("normConv same 25->32 + skip", 4, (22, 22, 9), 25, 32, (1, 1, 1), True),
         ("bwd-data of normConv: same 32->25", 4, (22, 22, 9), 32, 25, (1, 1, 1), False),
         ("convReducer valid 32->32", 4, (22, 22, 7), 32, 32, (0, 0, 0), False)]
BAR = 2e-6          # per-slice bound the native fp32-MFMA kernels meet on every case below (asserted for impl 2 as well)


@pytest.mark.parametrize("impl", [2, 4])
@pytest.mark.parametrize("scenario", ["dead_sample", "outlier_voxel", "filter_column_gains", "input_channel_gains", "everything"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_forward_slices_under_dynamic_range(dev, case, scenario, impl):
    name, N, hwt, Cin, Cout, pad, use_skip = case
    rng = np.random.default_rng(zlib.crc32((name + scenario).encode()))
    ho = tuple(hwt[i] + 2 * pad[i] - 2 for i in range(3))
    x = rng.normal(size=(N,) + hwt + (Cin,)).astype(np.float32)
    w = (rng.normal(size=(3, 3, 3, Cin, Cout)) / np.sqrt(27 * Cin)).astype(np.float32)
    skip = None
    if scenario in ("dead_sample", "everything"):
        x[1] *= np.float32(2.0 ** -30)                            # a near-zero patch between full-scale ones
        x[2] *= np.float32(2.0 ** 11)
    if scenario in ("outlier_voxel", "everything"):
        x[0, 7, 9, 2, :] *= np.float32(2.0 ** 20)                 # one hot voxel: everything else of that sample sits 2^-20 below the maximum
    if scenario in ("filter_column_gains", "everything"):
        w *= _gains(rng, Cout)                                    # output channels 2^-30 .. 1
    if scenario in ("input_channel_gains", "everything"):
        x *= _gains(rng, Cin, -24)                                # CONTRACTED channels 2^-24 .. 1
    if use_skip:                                                   # a skip tile of the output's own magnitude, per sample and channel
        ref0 = _oracle_conv(x, w, None, None, pad)
        skip = (rng.normal(size=ref0.shape) * np.abs(ref0).max(axis=(1, 2, 3), keepdims=True)).astype(np.float32)
    g = _geom(N, hwt[0], hwt[1], hwt[2], Cin, ho[0], ho[1], ho[2], Cout, (3, 3, 3), pad)
    try:
        y = _conv(dev, impl, g, x, w, None, skip)
    except ValueError:
        pytest.skip("geometry not covered by this kernel family's strip kernel")
    ref = _oracle_conv(x, w, None, skip, pad)
    e_slice = _slice_err(y, ref, (1, 2, 3))                       # slice = (sample, output channel)
    print("impl %d %s / %s: worst (sample, channel) slice error %.3g" % (impl, name, scenario, e_slice))
    assert e_slice < BAR, (name, scenario, impl, e_slice)


@pytest.mark.parametrize("impl", [2, 4])
@pytest.mark.parametrize("scenario", ["dead_sample", "outlier_voxel", "w2_column_gains", "input_channel_gains"])
def test_fused_pointwise_slices_under_dynamic_range(dev, scenario, impl):
    L = _L()
    D, vps, ns = 25, 22 * 22 * 3, 3
    nvox = vps * ns
    rng = np.random.default_rng(zlib.crc32(scenario.encode()))
    x = rng.normal(size=(ns, vps, 32)).astype(np.float32)
    w1 = (rng.normal(size=(32, 256)) / np.sqrt(32)).astype(np.float32)
    b1 = rng.normal(scale=0.3, size=256).astype(np.float32)
    w2 = (rng.normal(size=(256, D)) / 16).astype(np.float32)
    b2 = np.zeros(D, np.float32)
    ddec = rng.normal(size=(ns, vps, D)).astype(np.float32)
    dskip = np.zeros((ns, vps, 32), np.float32)
    if scenario == "dead_sample":
        x[1] *= np.float32(2.0 ** -30); b1[:] = 0                 # (with a bias the hidden units of a dead sample ARE the bias: nothing to resolve)
        ddec[2] *= np.float32(2.0 ** -30)
    if scenario == "outlier_voxel":
        x[0, 100] *= np.float32(2.0 ** 20); ddec[1, 7] *= np.float32(2.0 ** 20)
    if scenario == "w2_column_gains":
        w2 *= _gains(rng, D)
    gx = np.ones(32, np.float32)
    if scenario == "input_channel_gains":
        gx = _gains(rng, 32, -24)
        x *= gx
    X, W1, W2 = x.reshape(nvox, 32).astype(np.float64), w1.astype(np.float64), w2.astype(np.float64)
    Hpre = X @ W1 + b1
    Hh = np.maximum(Hpre, 0)
    ref = (Hh @ W2 + b2).reshape(ns, vps, D)
    xd, w1d, b1d, w2d, b2d, ddd, dsd = (_t(a, dev) for a in (x, w1, b1, w2, b2, ddec, dskip))
    dec = torch.full((nvox, D), float("nan"), device=dev)
    L.check(L.lib().probav_pw_forward(L.ptr(xd), L.ptr(w1d), L.ptr(b1d), L.ptr(w2d), L.ptr(b2d), L.ptr(dec), nvox, vps, D, impl, L.current_stream()))
    e_fwd = _slice_err(dec.cpu().numpy().reshape(ns, vps, D), ref, (1,))
    nbytes = L.lib().probav_pw_backward_scratch_bytes(D)
    scratch = torch.empty(nbytes // 4 + 1, device=dev)
    dH = (ddec.reshape(nvox, D).astype(np.float64) @ W2.T) * (Hpre > 0)
    rdx = (dH @ W1.T).reshape(ns, vps, 32)
    dx, dw1, db1 = torch.full((nvox, 32), float("nan"), device=dev), torch.full((32, 256), float("nan"), device=dev), torch.full((256,), float("nan"), device=dev)
    dw2, db2 = torch.full((256, D), float("nan"), device=dev), torch.full((D,), float("nan"), device=dev)
    L.check(L.lib().probav_pw_backward(L.ptr(xd), L.ptr(ddd), L.ptr(dsd), L.ptr(w1d), L.ptr(b1d), L.ptr(w2d), L.ptr(dx), L.ptr(dw1),
                                       L.ptr(db1), L.ptr(dw2), L.ptr(db2), L.ptr(scratch), nbytes, nvox, vps, D, impl, L.current_stream()))
    e_dx = _slice_err(dx.cpu().numpy().reshape(ns, vps, 32), rdx, (1,))
    # the weight gradients sum over the samples (a dead sample contributes nothing, by construction); slices: a ROW of dW1 (one input channel) and a COLUMN of dW2 (one
    # output channel), each against its own maximum (a whole-tensor norm hid the quiet channels' rows).  A gate the oracle itself calls undecidable in fp32
    # may go either way: its voxel's term is the only slack (see test_pointwise_filter_gradients_slice_by_slice)
    r1, r2 = X.T @ dH, Hh.T @ ddec.reshape(nvox, D)
    amb = np.abs(Hpre) < 4e-6 * (np.abs(X) @ np.abs(W1) + np.abs(b1))
    slack1 = np.abs(X).T @ (np.abs(ddec.reshape(nvox, D).astype(np.float64) @ W2.T) * amb)
    e1 = np.maximum(np.abs(dw1.cpu().double().numpy() - r1) - slack1, 0).max(axis=1) / np.maximum(np.abs(r1).max(axis=1), 1e-300)
    e2 = np.abs(dw2.cpu().double().numpy() - r2).max(axis=0) / np.maximum(np.abs(r2).max(axis=0), 1e-300)
    loud = gx >= 2.0 ** -18                                          # input channels whose two fp16 pieces are both normal at the sample's scale
    e_dw1, e_dw2 = float(e1[loud].max()), float(e2.max())
    e_dw1_quiet = float(e1[~loud].max()) if (~loud).any() else 0.0
    print("impl %d pointwise / %s: forward slice %.3g, dX slice %.3g, dW1 rows %.3g (rows of channels below 2^-18: %.3g), dW2 columns %.3g"
          % (impl, scenario, e_fwd, e_dx, e_dw1, e_dw1_quiet, e_dw2))
    assert e_fwd < 5e-6 and e_dx < 5e-6 and e_dw1 < 5e-6 and e_dw2 < 5e-6, (scenario, impl, e_fwd, e_dx, e_dw1, e_dw2)
    # the declared floor of the fused kernel's dW1 / dW2 (include/probav_hip.h, probav_pw_backward): impl 4 only; the other families hold every row to the bar
    assert e_dw1_quiet < (PW_DW_BAR_H3_FLOOR if impl == 4 else 5e-6), (scenario, impl, e_dw1_quiet)


# ---- the fused pointwise backward's FILTER gradients, slice by slice ----
# dW1[cin, hidden] = sum_voxels x[v, cin] dH'[v, hidden] and dW2[hidden, out] = sum_voxels H'[v, hidden] d_dec[v, out] contract over the voxels: the input channel of
# x and the output channel of d_dec are FREE indices there, but the kernel cuts x / d_dec ONCE per tile with the per-SAMPLE scale that products (a) / (b) -- which contract
# over those very channels -- need, and it has no register for a second accumulator set (conv3_wgrad_w4_kernel's cure: lifted second pieces).  So a channel that sits
# 2^-k below its sample's maximum keeps both fp16 pieces normal only down to k = 18 .. 19; below that the second piece loses one bit per binade.  The bound is DECLARED
# (include/probav_hip.h, probav_pw_backward) and asserted here per slice instead of hiding behind a whole-tensor norm:
PW_DW_BAR_FP32 = 5e-6        # every slice whose channel gain is >= 2^-18: the bar test_fused_pointwise_slices_under_dynamic_range holds all three families to (impl 2 meets it on EVERY slice)
PW_DW_BAR_H3_FLOOR = 1e-4    # impl 4, slices whose channel gain is below 2^-18 (down to 2^-24): the declared fp16 floor


def _pw_backward(dev, impl, x, w1, b1, w2, ddec, vps):
    L = _L()
    nvox, D = x.shape[0], w2.shape[1]
    xd, w1d, b1d, w2d, ddd = (_t(a, dev) for a in (x, w1, b1, w2, ddec))
    dsd = torch.zeros((nvox, 32), device=dev)
    nbytes = L.lib().probav_pw_backward_scratch_bytes(D)
    scratch = torch.empty(nbytes // 4 + 1, device=dev)
    dx, dw1, db1 = torch.full((nvox, 32), float("nan"), device=dev), torch.full((32, 256), float("nan"), device=dev), torch.full((256,), float("nan"), device=dev)
    dw2, db2 = torch.full((256, D), float("nan"), device=dev), torch.full((D,), float("nan"), device=dev)
    L.check(L.lib().probav_pw_backward(L.ptr(xd), L.ptr(ddd), L.ptr(dsd), L.ptr(w1d), L.ptr(b1d), L.ptr(w2d), L.ptr(dx), L.ptr(dw1),
                                       L.ptr(db1), L.ptr(dw2), L.ptr(db2), L.ptr(scratch), nbytes, nvox, vps, D, impl, L.current_stream()))
    return [t.cpu().double().numpy() for t in (dx, dw1, db1, dw2, db2)]


@pytest.mark.parametrize("which", ["x", "d_dec", "both"])
def test_pointwise_filter_gradients_slice_by_slice(dev, which):
    """Channel gains of 2^-24 .. 1 on the block's input channels (rows of dW1) and on d_dec's channels (columns of dW2); every row / column judged against ITS OWN
    maximum, against the fp64 oracle.  A ReLU gate is a step: a hidden value within rounding of zero may be gated either way by an fp32 device, and one such voxel is
    1 / sqrt(17 424) of a random-walk sum -- the oracle's tolerance for an element therefore carries the terms of the voxels whose pre-activation the oracle itself calls
    undecidable in fp32 (|pre| below 4e-6 of sum |x||w1| + |b1|), and nothing else."""
    D, vps, ns = 25, 22 * 22 * 9, 4
    nvox = vps * ns
    rng = np.random.default_rng(3)
    x = rng.normal(size=(nvox, 32)).astype(np.float32)
    w1 = (rng.normal(size=(32, 256)) / np.sqrt(32)).astype(np.float32)
    b1 = rng.normal(scale=0.3, size=256).astype(np.float32)
    w2 = (rng.normal(size=(256, D)) / 16).astype(np.float32)
    ddec = rng.normal(size=(nvox, D)).astype(np.float32)
    gx, gd = _gains(rng, 32, -24), _gains(rng, D, -24)
    if which in ("x", "both"):
        x *= gx
    else:
        gx = np.ones(32, np.float32)
    if which in ("d_dec", "both"):
        ddec *= gd
    else:
        gd = np.ones(D, np.float32)
    X, W1, W2, DD = x.astype(np.float64), w1.astype(np.float64), w2.astype(np.float64), ddec.astype(np.float64)
    pre = X @ W1 + b1
    Hh = np.maximum(pre, 0)
    draw = DD @ W2.T                                               # dH before the gate
    dH = draw * (pre > 0)
    r1, r2 = X.T @ dH, Hh.T @ DD
    amb = np.abs(pre) < 4e-6 * (np.abs(X) @ np.abs(W1) + np.abs(b1))      # gates fp32 cannot be asked to decide
    slack1 = np.abs(X).T @ (np.abs(draw) * amb)                  # what those voxels may move in dW1 [cin, hidden]
    print("pointwise backward, gains on %s: %d of %d gates undecidable in fp32" % (which, int(amb.sum()), amb.size))
    for impl in (4, 2):
        _, dw1, _, dw2, _ = _pw_backward(dev, impl, x, w1, b1, w2, ddec, vps)
        m1, m2 = np.abs(r1).max(axis=1), np.abs(r2).max(axis=0)   # a row of dW1 (one input channel), a column of dW2 (one output channel)
        e1 = (np.maximum(np.abs(dw1 - r1) - slack1, 0)).max(axis=1) / m1
        e2 = np.abs(dw2 - r2).max(axis=0) / m2
        hi1, hi2 = gx >= 2.0 ** -18, gd >= 2.0 ** -18
        def worst(e, sel):
            return float(e[sel].max()) if sel.any() else 0.0
        print("impl %d  dW1 rows: gain >= 2^-18 worst %.2e, below worst %.2e | dW2 columns: gain >= 2^-18 worst %.2e, below worst %.2e"
              % (impl, worst(e1, hi1), worst(e1, ~hi1), worst(e2, hi2), worst(e2, ~hi2)))
        assert worst(e1, hi1) < PW_DW_BAR_FP32 and worst(e2, hi2) < PW_DW_BAR_FP32, (which, impl, e1, e2)
        bar_lo = PW_DW_BAR_H3_FLOOR if impl == 4 else PW_DW_BAR_FP32
        assert worst(e1, ~hi1) < bar_lo and worst(e2, ~hi2) < bar_lo, (which, impl, e1, e2)


def _wgrad(dev, impl, g, x, dy, gate=None):
    L = _L()
    nbytes = L.lib().probav_conv3d_wgrad_scratch_bytes(ctypes.byref(g), impl)
    scratch = torch.empty(nbytes // 4 + 1, device=dev)
    dw, db = torch.empty((3, 3, 3, g[4], g[8]), device=dev), torch.empty((g[8],), device=dev)
    xd, dd, gd = _t(x, dev), _t(dy, dev), (_t(gate, dev) if gate is not None else None)
    L.check(L.lib().probav_conv3d_wgrad(ctypes.byref(g), L.ptr(xd), L.ptr(dd), L.ptr(gd), L.ptr(dw), L.ptr(db), L.ptr(scratch), nbytes, impl, L.current_stream()))
    return dw.cpu().double().numpy()


def _oracle_wgrad(x, dy, pad):
    xp = np.pad(x.astype(np.float64), [(0, 0), (pad,) * 2, (pad,) * 2, (pad,) * 2, (0, 0)])
    N, H, W, T, _ = dy.shape
    out = np.zeros((3, 3, 3, x.shape[-1], dy.shape[-1]))
    d = dy.astype(np.float64)
    for a in range(3):
        for b in range(3):
            for c in range(3):
                out[a, b, c] = np.einsum("nhwti,nhwto->io", xp[:, a:a + H, b:b + W, c:c + T, :], d)
    return out


@pytest.mark.parametrize("impl", [1, 4])
@pytest.mark.parametrize("scenario", ["dead_sample", "outlier_voxel"])
def test_wgrad_under_sample_and_voxel_range(dev, scenario, impl):
    """Backward-filter: a dead sample and a hot voxel do not cost the filter gradient anything (judged per (cin, cout) slice = per element)."""
    rng = np.random.default_rng(zlib.crc32(("wg" + scenario).encode()))
    N, hwt, Cin, Cout = 3, (22, 22, 9), 25, 32
    x = rng.normal(size=(N,) + hwt + (Cin,)).astype(np.float32)
    dy = rng.normal(size=(N,) + hwt + (Cout,)).astype(np.float32)
    if scenario == "dead_sample":
        x[1] *= np.float32(2.0 ** -30); dy[1] *= np.float32(2.0 ** -30)
    else:
        x[0, 5, 5, 4] *= np.float32(2.0 ** 20)
    g = _geom(N, 22, 22, 9, Cin, 22, 22, 9, Cout, (3, 3, 3), (1, 1, 1))
    ref = _oracle_wgrad(x, dy, 1)
    got = _wgrad(dev, impl, g, x, dy)
    e = _slice_err(got, ref, (0, 1, 2))                           # slice = (cin, cout): the 27 taps of one filter plane
    print("impl %d wgrad / %s: worst (cin, cout) slice error %.3g" % (impl, scenario, e))
    assert e < 1e-5, (scenario, impl, e)


@pytest.mark.parametrize("T", [9, 13])
@pytest.mark.parametrize("which", ["x", "dy", "both"])
def test_wgrad_resolves_every_channel_slice(dev, which, T):
    """VERDICT r4 #3.  The backward-filter product contracts over the voxels of ALL samples, so its operands take ONE scale per tensor; with plain second pieces
    a channel 2^-24 below its tensor mates got a 1e-2 gradient slice (rounds 2 - 4: the documented limit of H3).  conv3_wgrad_w4_kernel stores the second pieces LIFTED by
    2^11 (both pieces of a value normal 29 binades below the tensor's maximum) and sums the cross products in an accumulator of their own: channel gains of 2^-24 .. 1 on the
    input channels, on the gradient's channels and on both -- every (cin, cout) slice of dW held to the bar the native fp32-MFMA kernel meets (measured 3.3e-7 / 1.2e-7)."""
    rng = np.random.default_rng(7)
    N, hwt, Cin, Cout = 2, (22, 22, T), 25, 32                     # (T = 13: a workgroup takes half the columns of a row)
    x = rng.normal(size=(N,) + hwt + (Cin,)).astype(np.float32)
    dy = rng.normal(size=(N,) + hwt + (Cout,)).astype(np.float32)
    gx, gd = _gains(rng, Cin, -24), _gains(rng, Cout, -24)
    if which in ("x", "both"):
        x *= gx
    if which in ("dy", "both"):
        dy *= gd
    g = _geom(N, 22, 22, T, Cin, 22, 22, T, Cout, (3, 3, 3), (1, 1, 1))
    ref = _oracle_wgrad(x, dy, 1)
    for impl in (4, 3, 2):
        got = _wgrad(dev, impl, g, x, dy)
        e = _slice_err(got, ref, (0, 1, 2))                       # slice = (cin, cout): the 27 taps of one filter plane
        print("impl %d wgrad / channel gains on %s: worst (cin, cout) slice error %.3g" % (impl, which, e))
        assert e < BAR, (which, impl, e)


@pytest.mark.parametrize("layer", ["mirrored pads 22x22x9 -> 7", "mirrored pads 22x22x13 -> 11", "unpadded 22x22x7 -> 20x20x5", "unpadded 20x20x5 -> 18x18x3"])
def test_reducer_wgrad_resolves_every_channel_slice(dev, layer):
    """The same for the reducers' layers (32 -> 32 channels, dY masked by the layer's own output; models/modelsTF.py:123-150): the mirrored-pad layer and the two unpadded
    ones behind it -- conv3_wgrad_w4_kernel's second and third mode -- with channel gains of 2^-24 .. 1 on both operands."""
    rng = np.random.default_rng(zlib.crc32(layer.encode()))
    N = 3
    if layer.startswith("mirrored"):
        ti = 13 if "x13" in layer else 9
        hwt, ho, reflect, pad = (22, 22, ti), (22, 22, ti - 2), 1, (1, 1, 0)
    elif "22x22x7" in layer:
        hwt, ho, reflect, pad = (22, 22, 7), (20, 20, 5), 0, (0, 0, 0)
    else:
        hwt, ho, reflect, pad = (20, 20, 5), (18, 18, 3), 0, (0, 0, 0)
    x = (rng.normal(size=(N,) + hwt + (32,)) * _gains(rng, 32, -24)).astype(np.float32)
    dy = (rng.normal(size=(N,) + ho + (32,)) * _gains(rng, 32, -24)).astype(np.float32)
    gate = rng.normal(size=dy.shape).astype(np.float32)
    g = _geom(N, hwt[0], hwt[1], hwt[2], 32, ho[0], ho[1], ho[2], 32, (3, 3, 3), pad, reflect, 1)
    xp = np.pad(x.astype(np.float64), [(0, 0), (1, 1), (1, 1), (0, 0), (0, 0)], mode="reflect") if reflect else x
    ref = _oracle_wgrad(xp, dy * (gate > 0), 0)
    for impl in (4, 3):
        got = _wgrad(dev, impl, g, x, dy, gate)
        e = _slice_err(got, ref, (0, 1, 2))
        print("impl %d reducer wgrad / %s: worst (cin, cout) slice error %.3g" % (impl, layer, e))
        assert e < BAR, (layer, impl, e)


@pytest.mark.parametrize("hwt", [(22, 14, 9), (22, 22, 19), (9, 30, 5)], ids=["22x14x9", "22x22x19", "9x30x5"])
def test_wgrad_without_an_instance_of_the_lifted_kernel_is_not_scaled(dev, hwt):
    """A layer conv3_wgrad_w4_kernel has no instance for (depth 19, rows that are not 22 columns) runs the general backward-filter kernel with the x6 arithmetic also in the
    default family: H3 there would be H3 with plain second pieces and one scale per tensor -- a channel 2^-24 below its mates resolved to 7e-5 (rounds 2 - 4: the documented limit;
    profiles/r05_wgrad_lift_ab.txt) -- while bf16 pieces carry fp32's exponent and need no scale.  Channel gains on both operands, every (cin, cout) slice held to BAR."""
    rng = np.random.default_rng(zlib.crc32(repr(hwt).encode()))
    N, Cin, Cout = 2, 25, 32
    x = (rng.normal(size=(N,) + hwt + (Cin,)) * _gains(rng, Cin, -24)).astype(np.float32)
    dy = (rng.normal(size=(N,) + hwt + (Cout,)) * _gains(rng, Cout, -24)).astype(np.float32)
    g = _geom(N, hwt[0], hwt[1], hwt[2], Cin, hwt[0], hwt[1], hwt[2], Cout, (3, 3, 3), (1, 1, 1))
    if _L().lib().probav_conv3d_wgrad_scratch_bytes(ctypes.byref(g), 4) == 0:
        pytest.skip("geometry not covered by the MFMA backward-filter kernels (the engine falls back to the VALU kernel)")
    ref = _oracle_wgrad(x, dy, 1)
    e = _slice_err(_wgrad(dev, 4, g, x, dy), ref, (0, 1, 2))
    print("impl 4 wgrad on %s (no instance of the lifted kernel): worst (cin, cout) slice error %.3g" % (hwt, e))
    assert e < BAR, (hwt, e)


def _model(dev, params, T=9):
    from probav_amd.modelsTF import WDSRConv3D
    m = WDSRConv3D("t", "NIR", synth.NIR_MEAN, synth.NIR_STD, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, T, 16, True, seed=0)
    m.load_variables(params)
    return m.to(dev)


@pytest.mark.parametrize("impl", [4, 3, 2])
def test_forward_is_bitwise_independent_of_the_batch(dev, impl):
    """model(x) of the reference has no cross-sample term (models/modelsTF.py:15-43): a patch's prediction must not depend on what
    else is in the batch or on the micro-batch size -- bit for bit, for the default H3 family too (per-sample operand scales)."""
    m = _model(dev, synth.synth_params(seed=41, perturb=True))
    m.set_impl(impl)
    x = synth.synth_batch(48, seed=42)[0]
    x[5] *= np.float32(1e-3)                                      # a dim patch and a saturated one among ordinary ones
    x[6] = np.clip(x[6] * 3.0, 0, 65535)
    xd = torch.as_tensor(x).to(dev)
    with torch.no_grad():
        full = m(xd, training=False).clone()
        for lo, hi in ((0, 16), (16, 32), (5, 7), (6, 7), (40, 48)):
            part = m(xd[lo:hi].contiguous(), training=False)
            assert torch.equal(part, full[lo:hi]), (impl, lo, hi)
        perm = torch.randperm(48, generator=torch.Generator().manual_seed(0)).to(dev)
        shuffled = m(xd[perm].contiguous(), training=False)
        assert torch.equal(shuffled, full[perm]), impl
    # the training forward (saves activations) gives the same bits as the inference forward
    assert torch.equal(m(xd[:16].contiguous(), training=True).detach(), full[:16])


def test_end_to_end_with_a_dead_and_a_bright_patch(dev):
    """Whole network, default family: a patch at 2^-8 of the usual radiometry and one at 4x next to ordinary ones -- every sample's
    prediction within 1e-5 of the fp64 oracle relative to ITS OWN range (north_star: 1e-3)."""
    from oracle import wdsr_torch as ot
    params = synth.synth_params(seed=51, perturb=True)
    m = _model(dev, params)
    x = synth.synth_batch(4, seed=52)[0]
    x[1] = (x[1] - synth.NIR_MEAN) * np.float32(2.0 ** -8) + np.float32(synth.NIR_MEAN)      # a nearly flat patch (tiny normalised values)
    x[2] = np.clip(x[2] * 4.0, 0, 65535)
    with torch.no_grad():
        y = m(torch.as_tensor(x).to(dev), training=False).cpu().double().numpy()
    ref = ot.wdsr_forward(torch.tensor(x, dtype=torch.float64), ot.to_torch_params(params, requires_grad=False), synth.NIR_MEAN, synth.NIR_STD).numpy()
    for n in range(4):
        dev_n = np.abs(ref[n] - synth.NIR_MEAN).max()             # the sample's own signal range around the band mean
        e = np.abs(y[n] - ref[n]).max() / max(dev_n, 1.0)
        print("sample %d: |y - ref| max / own range = %.3g" % (n, e))
        assert e < 1e-5, (n, e)
