"""Parity tests proper: the HIP path, called through the C ABI (ctypes), against the CPU oracle on the same
seeded inputs, against the committed golden fixtures, and -- at the benchmark's full size (batch 128) --
through size-independent properties (batch independence, bitwise determinism, gradient linearity).

Tolerances (north_star): network output within 1e-3 relative fp32 (we hold 2e-5), loss within 1e-5."""
import ctypes
import os
import zlib

import numpy as np
import pytest
import torch

from oracle import wdsr_numpy as on
from oracle import wdsr_torch as ot
from probav_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _keep_workspaces(monkeypatch):
    """The gate tests of THIS module read a pass's workspace after its backward has run (modelsTF.WDSRModel.forward keeps only a weak
    reference by default).  Per test, so that every other GPU module runs with the product's default."""
    monkeypatch.setenv("PROBAV_KEEP_WS", "1")

GOLD = os.path.join(os.path.dirname(__file__), "golden")
# Un-gated golden gradients, relative L2 per tensor: ONE fixed bar for every depth and kernel family, 5e-3 -- not a number fitted to what a build happens to measure
# (ADVICE r4).  Where it comes from: the gradient of this network is discontinuous in its ReLU gates; an fp32 and an fp64 evaluation set the gates of pre-activations at ~0
# differently, and each flipped gate moves a filter-gradient entry by ~1/sqrt(#voxels) of its norm.  With B = 2 (8 712 voxels per block) that is the 1e-3 ... 3e-3 every
# correct fp32 evaluation shows, whatever its summation order (measured on MI355X, rounds 4 / 5: T = 9 2.7e-4 ... 3.1e-3 by family, T = 13 6.5e-4, T = 7 2.7e-3, T = 19 2.4e-3).
# What the bar cannot see -- an arithmetic error below it -- is what the gate-masked tests further down are for: the same gradients, the device's gates imposed on the
# oracle, held to 1e-5.
def _grad_l2_tol(T, impl):
    return 5e-3


IMPLS = [0, 1, 2, 3, 4]   # 4 = H3 kernels (three products of scaled fp16 piece pairs), same tolerances; 3 = x6 kernels (fp32 products as six bf16-piece MFMA products), held to the SAME tolerances


def _lib():
    from probav_amd import _lib as L
    return L


def _geom(N, Hi, Wi, Ti, Cin, Ho, Wo, To, Cout, k, pad, reflect=0, relu=0):
    return (ctypes.c_int32 * 17)(N, Hi, Wi, Ti, Cin, Ho, Wo, To, Cout, k[0], k[1], k[2], pad[0], pad[1], pad[2], reflect, relu)


def _t(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(dev)


def _oracle_conv(x, gate, w, bias, skip, pad, reflect, relu, out_hw_t):
    x = np.asarray(x, np.float64)
    if gate is not None:
        x = x * (np.asarray(gate) > 0)
    ph, pw, pt = pad
    if reflect:
        x = np.pad(x, [(0, 0), (ph, ph), (pw, pw), (0, 0), (0, 0)], mode="reflect")
        x = np.pad(x, [(0, 0), (0, 0), (0, 0), (pt, pt), (0, 0)])
    else:
        x = np.pad(x, [(0, 0), (ph, ph), (pw, pw), (pt, pt), (0, 0)])
    y = on.conv_valid(x, w)[:, :out_hw_t[0], :out_hw_t[1], :out_hw_t[2]]
    if bias is not None:
        y = y + np.asarray(bias, np.float64)
    if relu:
        y = np.maximum(y, 0)
    if skip is not None:
        y = y + np.asarray(skip, np.float64)
    return y


# name, N, (Hi,Wi,Ti), Cin, Cout, k, pad, reflect, relu, use_gate, use_skip
CONV_CASES = [
    ("mainConv1 same 1->32 relu", 2, (22, 22, 9), 1, 32, (3, 3, 3), (1, 1, 1), 0, 1, 0, 0),
    ("expConv 1x1x1 32->256 relu", 1, (6, 5, 9), 32, 256, (1, 1, 1), (0, 0, 0), 0, 1, 0, 0),
    ("decConv 1x1x1 256->25", 1, (6, 5, 9), 256, 25, (1, 1, 1), (0, 0, 0), 0, 0, 0, 0),
    ("normConv same 25->32 + skip", 2, (22, 22, 9), 25, 32, (3, 3, 3), (1, 1, 1), 0, 0, 0, 1),
    ("normConv small ragged", 3, (7, 5, 3), 25, 32, (3, 3, 3), (1, 1, 1), 0, 0, 0, 1),
    ("convReducer_1 reflect+valid", 2, (22, 22, 9), 32, 32, (3, 3, 3), (1, 1, 0), 1, 1, 0, 0),
    ("convReducer_2 valid relu", 2, (22, 22, 7), 32, 32, (3, 3, 3), (0, 0, 0), 0, 1, 0, 0),
    ("upscaleConv1 valid 32->9", 2, (18, 18, 3), 32, 9, (3, 3, 3), (0, 0, 0), 0, 0, 0, 0),
    ("residConv1 2-D 1->9 relu", 2, (22, 22, 1), 1, 9, (3, 3, 1), (0, 0, 0), 0, 1, 0, 0),
    ("residConv2 2-D 9->9", 2, (20, 20, 1), 9, 9, (3, 3, 1), (0, 0, 0), 0, 0, 0, 0),
    ("bwd-data of normConv: same 32->25", 2, (22, 22, 9), 32, 25, (3, 3, 3), (1, 1, 1), 0, 0, 0, 0),
    ("bwd-data of reducer: full 32->32 gated", 2, (20, 20, 5), 32, 32, (3, 3, 3), (2, 2, 2), 0, 0, 1, 0),
    ("bwd-data of expConv: 256->32 gated + skip", 1, (6, 5, 9), 256, 32, (1, 1, 1), (0, 0, 0), 0, 0, 1, 1),
    ("normConv strip: 3 patches, relu + skip", 3, (22, 22, 9), 25, 32, (3, 3, 3), (1, 1, 1), 0, 1, 0, 1),
    ("bwd-data of normConv gated (two channel passes)", 2, (22, 22, 9), 32, 25, (3, 3, 3), (1, 1, 1), 0, 0, 1, 0),
    ("bwd-data of convReducer_1: full 32->32 gated, 24x24x9 out", 2, (22, 22, 7), 32, 32, (3, 3, 3), (2, 2, 2), 0, 0, 1, 0),
    ("T=13 normConv same 25->32", 1, (22, 22, 13), 25, 32, (3, 3, 3), (1, 1, 1), 0, 0, 0, 1),
    ("bwd-data of the T=13 normConv: 32->25 gated (two column ranges in the piece-ring strip kernel)", 2, (22, 22, 13), 32, 25, (3, 3, 3), (1, 1, 1), 0, 0, 1, 0),
    # the alternating-halves strip kernel: one patch cut into five row strips (the last one short), the reducers with rows shorter
    # than 128 voxels, seven frames
    ("normConv one patch, five strips, relu + skip", 1, (22, 22, 9), 25, 32, (3, 3, 3), (1, 1, 1), 0, 1, 0, 1),
    ("bwd-data of normConv, one patch: same 32->25", 1, (22, 22, 9), 32, 25, (3, 3, 3), (1, 1, 1), 0, 0, 0, 0),
    ("convReducer_3 valid relu", 2, (20, 20, 5), 32, 32, (3, 3, 3), (0, 0, 0), 0, 1, 0, 0),
    ("bwd-data of convReducer_3: full 32->32 gated", 2, (18, 18, 3), 32, 32, (3, 3, 3), (2, 2, 2), 0, 0, 1, 0),
    ("T=7 normConv same 25->32 + skip", 3, (22, 22, 7), 25, 32, (3, 3, 3), (1, 1, 1), 0, 0, 0, 1),
    ("bwd-data of upscaleConv1: full 9->32, depth 1 -> 3", 3, (16, 16, 1), 9, 32, (3, 3, 3), (2, 2, 2), 0, 0, 0, 0),
    # the mirrored-pad layer at the other depths conv3_wgrad_w4_kernel is instantiated for (rows of 7 and 5 k-blocks; the T = 13 network's third reducer is the 9 -> 7 case above)
    ("mirrored-pad reducer, depth 7 -> 5", 3, (22, 22, 7), 32, 32, (3, 3, 3), (1, 1, 0), 1, 1, 0, 0),
    ("mirrored-pad reducer, depth 5 -> 3", 2, (22, 22, 5), 32, 32, (3, 3, 3), (1, 1, 0), 1, 1, 0, 0),
    # mirrored pads at other extents (rows that are one column range and rows that are cut), odd extents through the one-wave-per-SIMD kernels (or past them, where their plans decline)
    ("mirrored-pad 16x16x9 -> 7", 2, (16, 16, 9), 32, 32, (3, 3, 3), (1, 1, 0), 1, 1, 0, 0),
    ("mirrored-pad 10x22x5 -> 3", 3, (10, 22, 5), 32, 32, (3, 3, 3), (1, 1, 0), 1, 1, 0, 0),
    ("mirrored-pad 22x22x13 -> 11", 1, (22, 22, 13), 32, 32, (3, 3, 3), (1, 1, 0), 1, 1, 0, 0),
    ("normConv 5x10x9, three patches + skip", 3, (5, 10, 9), 25, 32, (3, 3, 3), (1, 1, 1), 0, 0, 0, 1),
    ("normConv 13x22x7 relu + skip", 2, (13, 22, 7), 25, 32, (3, 3, 3), (1, 1, 1), 0, 1, 0, 1),
    ("bwd-data of normConv 9x12x7: same 32->25", 2, (9, 12, 7), 32, 25, (3, 3, 3), (1, 1, 1), 0, 0, 0, 0),
    ("same 32->32 on 11x14x9 + skip", 2, (11, 14, 9), 32, 32, (3, 3, 3), (1, 1, 1), 0, 0, 0, 1),
    ("bwd-data of the mirrored-pad reducer 7 -> 5: full 32->32 gated, 24x24x7 out", 2, (22, 22, 5), 32, 32, (3, 3, 3), (2, 2, 2), 0, 0, 1, 0),
    ("bwd-data of the mirrored-pad reducer 5 -> 3: full 32->32 gated, 24x24x5 out", 2, (22, 22, 3), 32, 32, (3, 3, 3), (2, 2, 2), 0, 0, 1, 0),
    # the unpadded reducers through conv3_wgrad_w4_kernel's third mode (rows of 20 x 5 and 18 x 3 voxels: 7 and 4 k-blocks), other row counts (uneven strips, one-row strips)
    ("unpadded reducer 15x22x7 -> 13x20x5", 3, (15, 22, 7), 32, 32, (3, 3, 3), (0, 0, 0), 0, 1, 0, 0),
    ("unpadded reducer 9x20x5 -> 7x18x3", 5, (9, 20, 5), 32, 32, (3, 3, 3), (0, 0, 0), 0, 1, 0, 0),
    ("unpadded reducer 3x20x5 -> 1x18x3", 2, (3, 20, 5), 32, 32, (3, 3, 3), (0, 0, 0), 0, 1, 0, 0),
    # depth 13 through conv3_wgrad_w4_kernel: a workgroup takes half the columns of a row (the neighbour half's edge column beside its own)
    ("T=13 normConv, three patches + skip", 3, (22, 22, 13), 25, 32, (3, 3, 3), (1, 1, 1), 0, 0, 0, 1),
    ("T=13 normConv 5 rows", 2, (5, 22, 13), 25, 32, (3, 3, 3), (1, 1, 1), 0, 0, 0, 1),
    ("mirrored-pad reducer, depth 13 -> 11, three patches", 3, (22, 22, 13), 32, 32, (3, 3, 3), (1, 1, 0), 1, 1, 0, 0),
    ("mirrored-pad reducer, depth 11 -> 9", 2, (22, 22, 11), 32, 32, (3, 3, 3), (1, 1, 0), 1, 1, 0, 0),
]


def _out_dims(hwt, k, pad, reflect):
    return tuple(hwt[i] + 2 * pad[i] - k[i] + 1 for i in range(3))


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv3d_forward_matches_oracle(dev, case, impl):
    name, N, hwt, Cin, Cout, k, pad, reflect, relu, use_gate, use_skip = case
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    ho = _out_dims(hwt, k, pad, reflect)
    x = rng.normal(size=(N,) + hwt + (Cin,)).astype(np.float32)
    w =(rng.normal(size=k + (Cin, Cout)) / np.sqrt(np.prod(k) * Cin)).astype(np.float32)
    bias = rng.normal(size=Cout).astype(np.float32)
    gate = rng.normal(size=x.shape).astype(np.float32) if use_gate else None
    skip = rng.normal(size=(N,) + ho + (Cout,)).astype(np.float32) if use_skip else None
    g = _geom(N, hwt[0], hwt[1], hwt[2], Cin, ho[0], ho[1], ho[2], Cout, k, pad, reflect, relu)
    L = _lib()
    y = torch.full((N,) + ho + (Cout,), float("nan"), device=dev)
    args = [_t(a, dev) if a is not None else None for a in (x, gate, w, bias, skip)]
    rc = L.lib().probav_conv3d_forward(ctypes.byref(g), *[L.ptr(a) for a in args], L.ptr(y), impl, L.current_stream())
    if impl >= 1 and rc == L.PROBAV_EINVAL:
        pytest.skip("geometry not covered by this MFMA kernel (the engine falls back)")
    L.check(rc, "probav_conv3d_forward")
    ref = _oracle_conv(x, gate, w, bias, skip, pad, reflect, relu, ho)
    err = np.abs(y.cpu().double().numpy() - ref).max() / np.abs(ref).max()
    print("conv3d_forward impl %d %s: max err / max |ref| = %.3g" % (impl, name, err))
    assert err < 2e-6, "%s: rel err %.3e" % (name, err)


WGRAD_CASES = [c for c in CONV_CASES if not c[0].startswith("bwd-data")]


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("case", WGRAD_CASES, ids=[c[0] for c in WGRAD_CASES])
def test_conv3d_wgrad_matches_autograd(dev, case, impl):
    if impl == 2:
        pytest.skip("impl 2 only changes the forward / backward-data kernels")
    name, N, hwt, Cin, Cout, k, pad, reflect, relu, _, _ = case
    rng = np.random.default_rng(zlib.crc32(name.encode()) + 1)
    ho = _out_dims(hwt, k, pad, reflect)
    x = rng.normal(size=(N,) + hwt + (Cin,)).astype(np.float32)
    dy = rng.normal(size=(N,) + ho + (Cout,)).astype(np.float32)
    gate = rng.normal(size=dy.shape).astype(np.float32) if relu else None      # the layer's own output (ReLU mask)
    # oracle: d/dw of sum(conv(x, w) * dy_gated) via torch autograd in fp64
    xt = torch.tensor(x, dtype=torch.float64)
    if reflect:
        xp = torch.tensor(np.pad(x.astype(np.float64), [(0, 0), (pad[0],) * 2, (pad[1],) * 2, (0, 0), (0, 0)], mode="reflect"))
        xp = torch.nn.functional.pad(xp, (0, 0, pad[2], pad[2]))
    else:
        xp = torch.nn.functional.pad(xt, (0, 0, pad[2], pad[2], pad[1], pad[1], pad[0], pad[0]))
    wt = torch.zeros(k + (Cin, Cout), dtype=torch.float64, requires_grad=True)
    yt = torch.nn.functional.conv3d(xp.permute(0, 4, 1, 2, 3), wt.permute(4, 3, 0, 1, 2)).permute(0, 2, 3, 4, 1)
    dyg = torch.tensor(dy, dtype=torch.float64) * (torch.tensor(gate) > 0 if gate is not None else 1.0)
    (yt * dyg).sum().backward()
    L = _lib()
    g = _geom(N, hwt[0], hwt[1], hwt[2], Cin, ho[0], ho[1], ho[2], Cout, k, pad, reflect, relu)
    nbytes = L.lib().probav_conv3d_wgrad_scratch_bytes(ctypes.byref(g), impl)
    if impl in (1, 3, 4) and nbytes == 0:
        pytest.skip("geometry not covered by this MFMA backward-filter kernel (the engine falls back)")
    scratch = torch.empty(nbytes // 4 + 1, device=dev)
    dw = torch.full(k + (Cin, Cout), float("nan"), device=dev)
    db = torch.full((Cout,), float("nan"), device=dev)
    xd, dyd, gd = _t(x, dev), _t(dy, dev), (_t(gate, dev) if gate is not None else None)
    L.check(L.lib().probav_conv3d_wgrad(ctypes.byref(g), L.ptr(xd), L.ptr(dyd), L.ptr(gd), L.ptr(dw), L.ptr(db),
                                        L.ptr(scratch), nbytes, impl, L.current_stream()))
    ref_w, ref_b = wt.grad.numpy(), dyg.sum(dim=(0, 1, 2, 3)).numpy()
    print("conv3d_wgrad impl %d %s: dw err %.3g db err %.3g" % (impl, name, np.abs(dw.cpu().double().numpy() - ref_w).max() / np.abs(ref_w).max(),
                                                                np.abs(db.cpu().double().numpy() - ref_b).max() / np.abs(ref_b).max()))
    assert np.abs(dw.cpu().double().numpy() - ref_w).max() / np.abs(ref_w).max() < 1e-5
    assert np.abs(db.cpu().double().numpy() - ref_b).max() / np.abs(ref_b).max() < 1e-5
    # bitwise reproducible (fixed-order partial sums, no float atomics)
    dw2 = torch.empty_like(dw)
    L.check(L.lib().probav_conv3d_wgrad(ctypes.byref(g), L.ptr(xd), L.ptr(dyd), L.ptr(gd), L.ptr(dw2), L.ptr(db),
                                        L.ptr(scratch), nbytes, impl, L.current_stream()))
    assert torch.equal(dw, dw2)


def _model(dev, T=9, params=None, seed=0, gray=True):
    from probav_amd.modelsTF import WDSRConv3D
    m = WDSRConv3D("t", "NIR", synth.NIR_MEAN, synth.NIR_STD, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, T, 16, gray, seed=seed)
    if params is not None:
        m.load_variables(params)
    return m.to(dev)


def test_weight_norm_forward_backward(dev):
    params = synth.synth_params(seed=11, perturb=True)
    m = _model(dev, params=params)
    L = _lib()
    h = m._handle()
    assert [(n, tuple(s)) for n, _, _, _, s in m.native_layer_table()][:2] == [("mainConv1", (3, 3, 3, 1, 32)), ("expConv_0", (1, 1, 1, 32, 256))]
    for (n, g, v, b, _), Lh in zip(m.native_layer_table(), m.layers):
        assert (n, g, v, b) == (Lh.name, Lh.g_off, Lh.v_off, Lh.b_off)
    nw, nc = L.lib().probav_weff_count(h), L.lib().probav_cout_total(h)
    weff, weffT, invn = torch.empty(nw, device=dev), torch.empty(nw, device=dev), torch.empty(nc, device=dev)
    L.check(L.lib().probav_wn_forward(h, L.ptr(m.flat), L.ptr(weff), L.ptr(weffT), L.ptr(invn), L.current_stream()))
    rng = np.random.default_rng(0)
    dweff = torch.as_tensor(rng.normal(size=nw).astype(np.float32)).to(dev)
    grads = torch.zeros_like(m.flat)
    L.check(L.lib().probav_wn_backward(h, L.ptr(m.flat), L.ptr(dweff), L.ptr(invn), L.ptr(grads), L.current_stream()))
    weff, weffT, grads, dweff = weff.cpu().double().numpy(), weffT.cpu().double().numpy(), grads.cpu().double().numpy(), dweff.cpu().double().numpy()
    off = 0
    for Lh in m.layers:
        p = params[Lh.name]
        w = on.weight_norm(p["v"], p["g"])
        n = w.size
        got = weff[off:off + n].reshape(w.shape)
        assert np.abs(got - w).max() < 2e-6 * np.abs(w).max(), Lh.name
        taps = n // (w.shape[-1] * w.shape[-2])
        wT = w.reshape(taps, w.shape[-2], w.shape[-1])[::-1].transpose(0, 2, 1)      # flipped taps, [tap][co][ci]
        assert np.abs(weffT[off:off + n].reshape(wT.shape) - wT).max() < 2e-6 * np.abs(w).max(), Lh.name
        vt = torch.tensor(p["v"], dtype=torch.float64, requires_grad=True)
        gt = torch.tensor(p["g"], dtype=torch.float64, requires_grad=True)
        (ot.weight_norm(vt, gt) * torch.tensor(dweff[off:off + n].reshape(w.shape))).sum().backward()
        assert np.abs(grads[Lh.g_off:Lh.v_off] - gt.grad.numpy()).max() < 1e-5 * np.abs(gt.grad.numpy()).max(), Lh.name
        assert np.abs(grads[Lh.v_off:Lh.b_off] - vt.grad.numpy().reshape(-1)).max() < 1e-5 * np.abs(vt.grad.numpy()).max(), Lh.name
        off += n


@pytest.mark.parametrize("B,S,border", [(1, 48, 3), (5, 48, 3), (2, 96, 3), (3, 30, 2)])
def test_shift_losses_match_oracle(dev, B, S, border):
    from probav_amd.loss import Losses
    rng = np.random.default_rng(B * 1000 + S)
    hr = np.clip(rng.normal(synth.NIR_MEAN, synth.NIR_STD, (B, S, S, 1)), 0, 16383).astype(np.float32)
    pred = (hr + rng.normal(0, 150, hr.shape)).astype(np.float32)
    pred = np.roll(pred, (1, -2), axis=(1, 2))                       # true registration is not the centre shift
    mask = rng.random((B, S, S, 1)) < 0.9
    mask[0, : S // 3] = False                                        # a heavily clouded sample
    lo = Losses(targetShape=(S, S, 1), cropBorder=border)
    hd, md = torch.as_tensor(hr).to(dev), torch.as_tensor(mask).to(dev)
    pd = torch.as_tensor(pred).to(dev).requires_grad_(True)
    l1 = lo.shiftCompensatedL1Loss(hd, md, pd)
    l1.backward()
    l2 = lo.shiftCompensatedL2Loss(hd, md, pd.detach())
    ps = lo.shiftCompensatedcPSNR(hd, md, pd.detach())
    r1, r2 = on.shift_l1_loss(hr, mask, pred, border), on.shift_l2_loss(hr, mask, pred, border)
    assert abs(float(l1) - r1) < 1e-6 * r1                           # bar: 1e-5 relative
    assert abs(float(l2) - r2) < 1e-6 * r2
    np.testing.assert_allclose(ps.cpu().numpy(), on.shift_cpsnr(hr, mask, pred, border), rtol=1e-6)
    gref = on.shift_l1_grad(hr, mask, pred, border)
    assert np.abs(pd.grad.cpu().double().numpy() - gref).max() < 1e-6 * np.abs(gref).max()
    # L2 gradient against autograd, float mask accepted, upstream scale honoured
    pd2 = torch.as_tensor(pred).to(dev).requires_grad_(True)
    (3.0 * lo.shiftCompensatedL2Loss(hd, md.float(), pd2)).backward()
    pt = torch.tensor(pred, dtype=torch.float64, requires_grad=True)
    (3.0 * ot.shift_l2_loss(torch.tensor(hr), torch.tensor(mask), pt, border)).backward()
    assert np.abs(pd2.grad.cpu().double().numpy() - pt.grad.numpy()).max() < 1e-5 * np.abs(pt.grad.numpy()).max()
    with pytest.raises(ValueError):
        lo.shiftCompensatedL1Loss(hd[:, :-1], md, pd)


def test_sobel_l1_mix_loss_and_gradient(dev):
    """cfg loss = sobel_l1_mix: forward value and the gradient through the arg-min shift against the torch fp64 restatement."""
    from oracle import wdsr_torch as ot_
    from probav_amd.loss import Losses
    rng = np.random.default_rng(7)
    _, hr, mask = synth.synth_batch(5, seed=8)
    pred = (hr + rng.normal(0, 300, hr.shape)).astype(np.float32)
    lo = Losses(targetShape=(48, 48, 1))
    pd = torch.tensor(pred, device=dev, requires_grad=True)
    loss = lo.shiftCompensatedL1EdgeLoss(torch.tensor(hr).to(dev), torch.tensor(mask).to(dev), pd)
    loss.backward()
    pt = torch.tensor(pred, dtype=torch.float64, requires_grad=True)
    ref = ot_.shift_l1edge_loss(torch.tensor(hr), torch.tensor(mask), pt, border=3, pi=0.7)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref))
    g, gr = pd.grad.cpu().double().numpy(), pt.grad.numpy()
    assert np.abs(g - gr).max() < 1e-5 * np.abs(gr).max()


def test_l1msssim_loss_and_gradient(dev):
    """cfg loss = l1msssim: batch-level value and the gradient through the arg-min shift against the torch fp64 restatement."""
    from oracle import wdsr_torch as ot_
    from probav_amd.loss import Losses
    rng = np.random.default_rng(11)
    _, hr, mask = synth.synth_batch(4, seed=12)
    pred = (hr + rng.normal(0, 300, hr.shape)).astype(np.float32)
    lo = Losses(targetShape=(48, 48, 1))
    pd = torch.tensor(pred, device=dev, requires_grad=True)
    loss = lo.shiftCompensatedRevSSIM(torch.tensor(hr).to(dev), torch.tensor(mask).to(dev), pd)
    loss.backward()
    pt = torch.tensor(pred, dtype=torch.float64, requires_grad=True)
    ref = ot_.shift_revssim_loss(torch.tensor(hr), torch.tensor(mask), pt)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref)), (float(loss), float(ref))
    g, gr = pd.grad.cpu().double().numpy(), pt.grad.numpy()
    assert np.abs(g - gr).max() < 1e-4 * np.abs(gr).max(), (np.abs(g - gr).max(), np.abs(gr).max())


def test_fused_nadam_matches_keras_restatement(dev):
    """HIP Nadam (probav_nadam_step) against the fp64 restatement of the Keras rule, incl. checkpoint round trip."""
    from oracle.nadam_numpy import Nadam
    from probav_amd.trainClass import HipNadam
    rng = np.random.default_rng(1)
    theta = rng.normal(size=535267).astype(np.float32)
    p = torch.nn.Parameter(torch.as_tensor(theta).to(dev))
    opt, ref = HipNadam([p], lr=5e-4), Nadam(lr=5e-4)
    th = theta.astype(np.float64)
    for k in range(5):
        g = (rng.normal(size=theta.shape) * (10.0 ** rng.integers(-3, 2))).astype(np.float32)
        p.grad = torch.as_tensor(g).to(dev)
        opt.step()
        th = ref.step(th, g)
        if k == 2:                                      # resume from a state_dict, like ModelTrainer.restore does
            sd = opt.state_dict()
            opt = HipNadam([p], lr=1.0)
            opt.load_state_dict(sd)
    assert np.abs(p.detach().cpu().double().numpy() - th).max() < 2e-6 * np.abs(th).max()


@pytest.mark.parametrize("name", ["adam", "sgd"])
def test_fused_adam_sgd_match_keras_restatement(dev, name):
    """train.py:77-83's other optimizers through the same fused launch, against the fp64 restatements of the Keras rules."""
    from oracle import nadam_numpy as on_
    from probav_amd.trainClass import HipAdam, HipSGD
    rng = np.random.default_rng(2)
    theta = rng.normal(size=100003).astype(np.float32)
    p = torch.nn.Parameter(torch.as_tensor(theta).to(dev))
    opt, ref = (HipAdam([p], lr=5e-4), on_.Adam(lr=5e-4)) if name == "adam" else (HipSGD([p], lr=1e-2), on_.SGD(lr=1e-2))
    th = theta.astype(np.float64)
    for _ in range(4):
        g = (rng.normal(size=theta.shape) * (10.0 ** rng.integers(-3, 2))).astype(np.float32)
        p.grad = torch.as_tensor(g).to(dev)
        opt.step()
        th = ref.step(th, g)
    assert np.abs(p.detach().cpu().double().numpy() - th).max() < 2e-6 * np.abs(th).max()


def test_clip_round_half_to_even(dev):
    L = _lib()
    x = torch.tensor([-3.2, 0.5, 1.5, 2.5, 65535.5, 65536.4, 70000.0, 123.49], device=dev)
    y = torch.empty_like(x)
    L.check(L.lib().probav_clip_round(L.ptr(x), L.ptr(y), x.numel(), 0.0, 65536.0, L.current_stream()))
    assert y.tolist() == [0.0, 0.0, 2.0, 2.0, 65536.0, 65536.0, 65536.0, 123.0]      # tf.round: half to even


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("T", [9, 13, 7, 19])
def test_end_to_end_against_golden(dev, T, impl):
    """forward, loss, metric and all 132 (138 / 129 / 153) gradients against the committed fp64 fixtures."""
    from probav_amd.loss import Losses
    z = np.load(os.path.join(GOLD, "wdsr_t%d_b2.npz" % T))
    seeds = {9: (101, 102), 13: (131, 132), 7: (71, 72), 19: (191, 192)}[T]
    params = synth.synth_params(seed=seeds[0], perturb=True, numImgLR=T)
    m = _model(dev, T, params)
    m.set_impl(impl)
    lo = Losses(targetShape=(48, 48, 1))
    x, hr, mask = (torch.as_tensor(z[k]).to(dev) for k in ("x", "hr", "mask"))
    pred = m(x, training=True)
    loss = lo.shiftCompensatedL1Loss(hr, mask, pred)
    loss.backward()
    e = np.abs(pred.detach().cpu().double().numpy() - z["pred"]).max() / np.abs(z["pred"]).max()
    assert e < 2e-5, "output rel err %.3e (bar 1e-3)" % e
    assert abs(float(loss) - float(z["loss_l1"])) < 1e-5 * float(z["loss_l1"])
    np.testing.assert_allclose(lo.shiftCompensatedcPSNR(hr, mask, pred.detach()).cpu().numpy(), z["cpsnr"], rtol=1e-5)
    assert abs(float(lo.shiftCompensatedL2Loss(hr, mask, pred.detach())) - float(z["loss_l2"])) < 1e-4 * float(z["loss_l2"])
    grads = [g.cpu().double().numpy() for g in m.variable_gradients()]
    names = m.variable_names
    assert len(grads) == z["grad_norms"].shape[0]
    if T == 9:
        # ALL 132 gradient tensors element-wise against the committed fp64 gradient (tests/golden/wdsr_t9_b2_grads.npz): a norm cannot
        # see a wrong direction.  Metric here: relative L2 per tensor (ReLU gates at ~0 flip between an fp32 and an fp64 evaluation;
        # test_gradients_match_oracle_with_the_devices_relu_masks removes exactly that effect and then holds 1e-3 in the max norm).
        gref = np.load(os.path.join(GOLD, "wdsr_t9_b2_grads.npz"))["grad_flat"].astype(np.float64)
        gdev = m.flat.grad.detach().cpu().double().numpy()
        worst = {"g": (0.0, None), "v": (0.0, None), "bias": (0.0, None)}
        for L_ in m.layers:
            for key, lo, hi in (("g", L_.g_off, L_.v_off), ("v", L_.v_off, L_.b_off), ("bias", L_.b_off, L_.b_off + L_.cout)):
                tol = _grad_l2_tol(9, impl)
                err = np.sqrt(((gdev[lo:hi] - gref[lo:hi]) ** 2).sum()) / (np.sqrt((gref[lo:hi] ** 2).sum()) + 1e-30)
                if err > worst[key][0]:
                    worst[key] = (err, L_.name)
                assert err < tol, (L_.name, key, err)
        print("impl %d: worst un-gated relative-L2 gradient error per tensor kind: %s" % (impl, worst))
    worst_sub = {"g": 0.0, "v": 0.0}
    for k, (n, g) in enumerate(zip(names, grads)):
        ref_norm, ref_max = z["grad_norms"][k]
        # Metric: relative L2 error per tensor.  Element-wise maxima are not meaningful for these gradients: one
        # ReLU gate whose pre-activation is ~0 can flip between an fp32 and an fp64 evaluation and moves the
        # affected filter-gradient entries by ~1/sqrt(#voxels) (each entry is a sum over all voxels with heavy
        # cancellation); the gradient of a weight-norm gain `g` is the ill-conditioned dot product <dw, v>/||v||.
        tol = _grad_l2_tol(T, impl)
        assert abs(np.sqrt((g ** 2).sum()) - ref_norm) < tol * ref_norm + 1e-12, n
        key = "grad/" + n
        if key in z.files:
            e2 = np.sqrt(((g - z[key]) ** 2).sum()) / (ref_norm + 1e-30)
            wk = "g" if n.endswith("/g") else "v"
            worst_sub[wk] = max(worst_sub[wk], e2)
            assert e2 < tol, (n, e2)
    print("impl %d T %d: worst un-gated relative-L2 error over the stored gradient tensors: %s" % (impl, T, worst_sub))
    # inference mode (ping-pong workspace) gives the same prediction bit for bit
    with torch.no_grad():
        assert torch.equal(m(x, training=False), pred.detach())


@pytest.mark.parametrize("nvox,vps", [(32, 0), (1000, 0), (4356 * 3, 0), (5 * 1640, 1640), (600 * 64, 64), (7 * 33, 33), (1030 * 40, 40)],
                         ids=["32", "1000", "13068", "5 samples of 1640 (partial last tiles, sample boundaries inside the runs)",
                              "600 samples of 64 (runs longer than a sample: the general kernel)", "7 samples of 33 (one full and one one-voxel tile each)",
                              "1030 samples of 40 (more samples than waves)"])
# (No case of the benchmark's size here: the comparison is un-gated, and among 130 x 4 356 voxels x 256 hidden channels a handful of pre-activations sit within fp32 rounding of
#  zero -- every impl, 2 included, then differs from the fp64 reference by a whole term in three or four voxels of dX.  Full-size runs are compared at the device's own
#  gates: test_full_size_batch128_properties, test_forward_and_reverse_pass_decide_the_same_relu_gates.)
def test_fused_pointwise_forward_backward(dev, nvox, vps):
    """expConv + ReLU + decConv fused in accumulators (and its fused reverse pass) against fp64 numpy.  vps = voxels per sample (0: one
    sample): H3 scales, tiles and the reverse kernel's runs follow the samples."""
    L = _lib()
    D = 25
    rng = np.random.default_rng(nvox)
    x = rng.normal(size=(nvox, 32)).astype(np.float32)
    w1 = (rng.normal(size=(32, 256)) / np.sqrt(32)).astype(np.float32)
    b1 = rng.normal(scale=0.3, size=256).astype(np.float32)
    w2 = (rng.normal(size=(256, D)) / 16).astype(np.float32)
    b2 = rng.normal(scale=0.3, size=D).astype(np.float32)
    ddec = rng.normal(size=(nvox, D)).astype(np.float32)
    dskip = rng.normal(size=(nvox, 32)).astype(np.float32)
    xd, w1d, b1d, w2d, b2d, ddd, dsd = (_t(a, dev) for a in (x, w1, b1, w2, b2, ddec, dskip))
    X, W1, W2 = x.astype(np.float64), w1.astype(np.float64), w2.astype(np.float64)
    Hpre = X @ W1 + b1
    Hh = np.maximum(Hpre, 0)
    ref = Hh @ W2 + b2
    for impl in (2, 3, 4):                    # native fp32 MFMA, the six-product bf16 split and the three-product fp16 split: one tolerance
        dec = torch.full((nvox, D), float("nan"), device=dev)
        L.check(L.lib().probav_pw_forward(L.ptr(xd), L.ptr(w1d), L.ptr(b1d), L.ptr(w2d), L.ptr(b2d), L.ptr(dec), nvox, vps, D, impl,
                                          L.current_stream()))
        err = np.abs(dec.cpu().double().numpy() - ref).max() / np.abs(ref).max()
        print("pw_forward impl %d nvox %d: max err / max |ref| = %.3g" % (impl, nvox, err))
        assert err < 2e-6, (impl, err)
    nbytes = L.lib().probav_pw_backward_scratch_bytes(D)
    scratch = torch.empty(nbytes // 4 + 1, device=dev)
    dH = (ddec.astype(np.float64) @ W2.T) * (Hpre > 0)
    refs = {"dx": dskip + dH @ W1.T, "dw1": X.T @ dH, "db1": dH.sum(0), "dw2": Hh.T @ ddec, "db2": ddec.astype(np.float64).sum(0)}
    for impl in (2, 3, 4):
        dx, dw1, db1 = torch.full((nvox, 32), float("nan"), device=dev), torch.full((32, 256), float("nan"), device=dev), torch.full((256,), float("nan"), device=dev)
        dw2, db2 = torch.full((256, D), float("nan"), device=dev), torch.full((D,), float("nan"), device=dev)
        L.check(L.lib().probav_pw_backward(L.ptr(xd), L.ptr(ddd), L.ptr(dsd), L.ptr(w1d), L.ptr(b1d), L.ptr(w2d), L.ptr(dx), L.ptr(dw1),
                                           L.ptr(db1), L.ptr(dw2), L.ptr(db2), L.ptr(scratch), nbytes, nvox, vps, D, impl, L.current_stream()))
        for name, got in (("dx", dx), ("dw1", dw1), ("db1", db1), ("dw2", dw2), ("db2", db2)):
            r = refs[name]
            err = np.abs(got.cpu().double().numpy() - r).max() / np.abs(r).max()
            print("pw_backward impl %d nvox %d %s: max err / max |ref| = %.3g" % (impl, nvox, name, err))
            assert err < 5e-6, "impl %d %s rel err %.3e" % (impl, name, err)


@pytest.mark.parametrize("batch", [5, 1, 37])
def test_mfma_engine_matches_direct_engine(dev, batch):
    """The kernel families of the engine (generic direct kernels / fp32-MFMA kernels / x6 kernels) agree on ragged batches."""
    from probav_amd.loss import Losses
    params = synth.synth_params(seed=31, perturb=True)
    m = _model(dev, params=params)
    lo = Losses(targetShape=(48, 48, 1))
    x, hr, mask = (torch.as_tensor(a).to(dev) for a in synth.synth_batch(batch, seed=32))
    res = []
    for impl in (0, 2, 3, 4):
        m.set_impl(impl)
        m.flat.grad = None
        p = m(x, training=True)
        lo.shiftCompensatedL1Loss(hr, mask, p).backward()
        res.append((p.detach().clone(), [g.clone() for g in m.variable_gradients()]))
    m.set_impl(1)
    with torch.no_grad():
        p1 = m(x, training=False)
    assert float((res[0][0] - p1).abs().max()) < 1e-5 * float(res[0][0].abs().max())
    for other in (1, 2, 3):
        assert float((res[0][0] - res[other][0]).abs().max()) < 1e-5 * float(res[0][0].abs().max())
        # two fp32 summation orders can flip a few of the ~10^8 ReLU gates whose pre-activation is ~0, which moves single
        # filter-gradient entries by ~1/sqrt(#voxels): compare in relative L2 per tensor (the sharp per-kernel checks are
        # the single-operator tests above); tiny batches have few voxels per gate, hence the looser bound there
        for n, g0, g1 in zip(m.variable_names, res[0][1], res[other][1]):
            tol = (2e-2 if n.endswith("/g") else 5e-3) * (1.0 if batch >= 5 else 4.0)
            assert float((g0 - g1).norm()) <= tol * float(g0.norm()) + 1e-12, (n, other)


@pytest.mark.parametrize("P,T,B", [(24, 9, 3), (32, 9, 2), (16, 7, 4), (20, 13, 2)], ids=["p24-t9", "p32-t9", "p16-t7", "p20-t13"])
def test_other_patch_sizes_agree_across_engines(dev, P, T, B):
    """patchSizeLR is a cfg value (`train.py`, `[Patches]`): the one-wave-per-SIMD kernels are instantiated for the reference's 16 (rows of 22 columns) and must hand other
    sizes to the general forms -- the H3 engine (impl 4) against the generic direct kernels (impl 0) on 30-, 38- and 26-wide inputs: predictions to 1e-5 of their max-norm,
    the flat gradient in relative L2 (two summation orders flip a few ReLU gates at zero: the sharp per-kernel bars are the single-operator tests)."""
    from probav_amd.loss import Losses
    from probav_amd.modelsTF import WDSRConv3D
    m = WDSRConv3D("t", "NIR", synth.NIR_MEAN, synth.NIR_STD, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, T, P, True, seed=3).to(dev)
    rng = np.random.default_rng(P * 100 + T)
    x = torch.as_tensor((rng.uniform(size=(B, P + 6, P + 6, T, 1)) * 8000).astype(np.float32)).to(dev)
    hr = torch.as_tensor((rng.uniform(size=(B, 3 * P, 3 * P, 1)) * 8000).astype(np.float32)).to(dev)
    mask = torch.as_tensor((rng.uniform(size=(B, 3 * P, 3 * P, 1)) > 0.1).astype(np.float32)).to(dev)
    lo = Losses(targetShape=(3 * P, 3 * P, 1))
    res = []
    for impl in (0, 4):
        m.set_impl(impl)
        m.flat.grad = None
        p = m(x, training=True)
        lo.shiftCompensatedL1Loss(hr, mask, p).backward()
        res.append((p.detach().clone(), m.flat.grad.detach().clone()))
    assert torch.isfinite(res[1][0]).all() and torch.isfinite(res[1][1]).all()
    assert float((res[0][0] - res[1][0]).abs().max()) < 1e-5 * float(res[0][0].abs().max())
    assert float((res[0][1] - res[1][1]).norm()) < 5e-3 * float(res[0][1].norm())


@pytest.mark.parametrize("T", [9, 13])
def test_full_size_batch128_properties(dev, T):
    """BASELINE.json config 2 (T = 9) and config 3 (T = 13, the reference's longer-T network) at batch 128: properties that need no
    oracle run at that size, plus the oracle on two samples of the batch (a sample's forward result does not depend on its batch)."""
    from probav_amd.loss import Losses
    params = synth.synth_params(seed=21, perturb=True, numImgLR=T)
    m = _model(dev, T, params=params)
    lo = Losses(targetShape=(48, 48, 1))
    x, hr, mask = (torch.as_tensor(a).to(dev) for a in synth.synth_batch(128, seed=22, numImgLR=T))

    def step(xs, hs, ms):
        m.flat.grad = None
        p = m(xs, training=True)
        l = lo.shiftCompensatedL1Loss(hs, ms, p)
        l.backward()
        return p.detach().clone(), float(l), m.flat.grad.detach().clone()

    p_full, l_full, g_full = step(x, hr, mask)
    assert torch.isfinite(p_full).all() and torch.isfinite(g_full).all()
    p_again, l_again, g_again = step(x, hr, mask)
    assert torch.equal(p_full, p_again) and l_full == l_again and torch.equal(g_full, g_again)      # deterministic
    # batch independence: samples are processed independently (no cross-sample term in the forward)
    with torch.no_grad():
        p_small = m(x[5:8].contiguous(), training=False)
    assert torch.equal(p_small, p_full[5:8])
    # ... so two samples of the full batch can be judged by the fp64 oracle at oracle-sized cost (north_star: 1e-3 relative fp32)
    with torch.no_grad():
        pick = [77, 127]
        po = ot.wdsr_forward(torch.tensor(x[pick].cpu().numpy(), dtype=torch.float64), ot.to_torch_params(params, requires_grad=False),
                             synth.NIR_MEAN, synth.NIR_STD, numImgLR=T).numpy()
    e = np.abs(p_full[pick].cpu().double().numpy() - po).max() / np.abs(po).max()
    assert e < 2e-5, e
    # linearity of the gradient in the batch: grad(mean over 128) = mean of the two half-batch gradients
    _, l_a, g_a = step(x[:64].contiguous(), hr[:64].contiguous(), mask[:64].contiguous())
    _, l_b, g_b = step(x[64:].contiguous(), hr[64:].contiguous(), mask[64:].contiguous())
    assert abs(0.5 * (l_a + l_b) - l_full) < 1e-5 * l_full
    gm = 0.5 * (g_a + g_b)
    assert float((gm - g_full).abs().max()) < 2e-4 * float(g_full.abs().max())
    # a zeroed pixel-shuffle head leaves exactly the denormalised zero: y = (0 + 0) * std + mean
    with torch.no_grad():
        keep = m.flat.detach().clone()
        for L in m.layers:
            if L.name in ("upscaleConv1", "residConv3"):
                m.flat[L.g_off:L.v_off] = 0
                m.flat[L.b_off:L.b_off + L.cout] = 0
        y0 = m(x[:4].contiguous(), training=False)
        assert float((y0 - synth.NIR_MEAN).abs().max()) < 1e-3
        m.flat.copy_(keep)


def test_trainer_and_inference_on_device(dev, tmp_path):
    from probav_amd.loss import Losses
    from probav_amd.trainClass import ModelTrainer, make_optimizer
    from probav_amd import testClass
    m = _model(dev, seed=3)
    lo = Losses(targetShape=(48, 48, 1))
    from probav_amd.trainClass import HipNadam
    opt = make_optimizer("nadam", m, 5e-4)
    assert isinstance(opt, HipNadam)                   # on the device the optimizer update is the fused HIP kernel
    tr = ModelTrainer(m, lo.shiftCompensatedL1Loss, lo.shiftCompensatedcPSNR, opt,
                      str(tmp_path / "ck"), str(tmp_path / "lg"), multiGPU=False)
    x, _, mask = synth.synth_batch(8, seed=4)
    rng = np.random.default_rng(4)
    hr = np.repeat(np.repeat(x.mean(axis=3)[:, 3:19, 3:19], 3, axis=1), 3, axis=2) + rng.normal(0, 20, (8, 48, 48, 1)).astype(np.float32)
    xs, hs, ms = torch.as_tensor(x).to(dev), torch.as_tensor(hr.astype(np.float32)).to(dev), torch.as_tensor(mask).to(dev)
    losses = []
    for _ in range(12):
        tr.trainLoss.reset_states()
        tr.trainStep(xs, hs, ms)
        losses.append(tr.trainLoss.result())
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
    tr.testStep(xs, hs, ms)
    assert tr.testPSNR.result() > 0
    # inference path of test.py: 64 patches -> micro-batches of 16 -> clip/round -> 8x8 stitch
    patches = synth.synth_batch(64, seed=9)[0]
    sr = testClass.resolveByBatch(m, patches, batch_size=16)
    assert sr.shape == (64, 48, 48, 1) and np.all(sr == np.round(sr)) and sr.min() >= 0 and sr.max() <= 65536
    one = testClass.resolve(m, patches[20:21])
    np.testing.assert_array_equal(one[0], sr[20])
    img = testClass.reconstruct_from_patches(sr)
    assert img.shape == (384, 384, 1)
    np.testing.assert_array_equal(img[48:96, 96:144], sr[8 + 2])          # row-major block order (test.py:149-160)
    imgs = testClass.evaluate(m, patches[None], batch_size=16)
    np.testing.assert_array_equal(imgs[0], img)
    np.testing.assert_array_equal(testClass.evaluate_device(m, patches[None], micro_batch=16)[0], img)      # the device pipeline test.py uses
    np.testing.assert_array_equal(testClass.evaluate_device(m, patches[None], micro_batch=16, launch_batch=16)[0], img)      # ... one launch set per micro-batch
    looped = np.concatenate([testClass.resolve(m, patches[16 * i:16 * i + 16]) for i in range(4)])       # the reference's loop as written (test.py:125-134)
    np.testing.assert_array_equal(looped, sr)
    # test.py:137-146, resolveBySampleAveraging: mean of 20 clipped + rounded predictions over compounding frame permutations
    avg = testClass.resolveBySampleAveraging(m, patches[:4], rng=np.random.default_rng(3)).cpu().numpy()
    rng3, xp, acc = np.random.default_rng(3), patches[:4], 0.0
    for _ in range(20):
        xp = xp[:, :, :, rng3.permutation(9), :]
        acc = acc + testClass.resolve(m, np.ascontiguousarray(xp)).astype(np.float64)
    np.testing.assert_allclose(avg, (acc / 20.0).astype(np.float32), rtol=0, atol=1e-3)
    # device-side pipeline for whole images: unfold of the reflect-padded frame, any micro-batch, on-device stitch
    frames = np.clip(np.random.default_rng(2).normal(synth.NIR_MEAN, synth.NIR_STD, (2, 9, 128, 128)), 0, 16383).astype(np.float32)
    pt = testClass.unfold_frames(torch.as_tensor(frames).to(dev))
    assert tuple(pt.shape) == (2, 64, 22, 22, 9, 1)
    padded = np.pad(frames, [(0, 0), (0, 0), (3, 3), (3, 3)], mode="reflect")
    np.testing.assert_array_equal(pt[1, 8 * 3 + 5, :, :, 4, 0].cpu().numpy(), padded[1, 4, 48:70, 80:102])      # patch (3,5): rows 48.., cols 80..
    big = testClass.resolve_images(m, pt, micro_batch=128)
    ref_imgs = testClass.evaluate(m, pt.cpu().numpy(), batch_size=16)
    # Samples are independent in every kernel family (the H3 kernels scale their operands per SAMPLE, not per batch): the reference's
    # micro-batches of 16 and one batch of 128 give the same pixels, bit for bit.
    np.testing.assert_array_equal(big.cpu().numpy(), np.stack(ref_imgs)[..., 0])
    m.set_impl(3)
    big3 = testClass.resolve_images(m, pt, micro_batch=128)
    ref3 = testClass.evaluate(m, pt.cpu().numpy(), batch_size=16)
    np.testing.assert_array_equal(big3.cpu().numpy(), np.stack(ref3)[..., 0])
    assert np.abs(big3.cpu().numpy() - big.cpu().numpy()).max() <= 1.0
    m.set_impl(4)
    # every forward call owns its workspace (an output of torch.ops.probav.wdsr_forward): a second forward before the first one's
    # backward clobbers nothing, and both reverse passes give the same gradient
    p1 = m(xs, training=True)
    p2 = m(xs[:4].contiguous(), training=True)
    m.flat.grad = None
    p1.sum().backward()
    g1 = m.flat.grad.clone()
    m.flat.grad = None
    m(xs, training=True).sum().backward()
    assert torch.equal(g1, m.flat.grad)
    m.flat.grad = None
    p2.sum().backward()
    assert torch.isfinite(m.flat.grad).all()
    with torch.no_grad():
        y = m(xs, training=True)                              # no graph: inference workspace, nothing to differentiate
    assert not y.requires_grad


@pytest.mark.parametrize("ea,eb", [(-37, 21), (30, -5), (0, 0)])
def test_h3_kernels_do_not_care_about_operand_magnitudes(dev, ea, eb):
    """H3 arithmetic scales every operand tensor by a power of two chosen from its largest magnitude (x6_device.h), so multiplying
    an operand by a power of two must change the result by exactly that factor -- bit for bit -- wherever fp32 itself allows it."""
    L = _lib()
    rng = np.random.default_rng(99)
    N, hwt, Cin, Cout = 2, (22, 22, 9), 25, 32
    x = rng.normal(size=(N,) + hwt + (Cin,)).astype(np.float32)
    w = (rng.normal(size=(3, 3, 3, Cin, Cout)) / 26).astype(np.float32)
    dy = rng.normal(size=(N,) + hwt + (Cout,)).astype(np.float32)
    g = _geom(N, 22, 22, 9, Cin, 22, 22, 9, Cout, (3, 3, 3), (1, 1, 1), 0, 0)
    sa, sb = np.float32(2.0 ** ea), np.float32(2.0 ** eb)

    def fwd(xx, ww):
        y = torch.full((N,) + hwt + (Cout,), float("nan"), device=dev)
        xd, wd = _t(xx, dev), _t(ww, dev)                   # (named: the operands must outlive the call)
        L.check(L.lib().probav_conv3d_forward(ctypes.byref(g), L.ptr(xd), None, L.ptr(wd), None, None, L.ptr(y), 4, L.current_stream()))
        return y.cpu().numpy()

    def wgrad(xx, dd):
        nbytes = L.lib().probav_conv3d_wgrad_scratch_bytes(ctypes.byref(g), 4)
        scratch = torch.empty(nbytes // 4 + 1, device=dev)
        dw, db = torch.empty((3, 3, 3, Cin, Cout), device=dev), torch.empty((Cout,), device=dev)
        xd, dd_ = _t(xx, dev), _t(dd, dev)
        L.check(L.lib().probav_conv3d_wgrad(ctypes.byref(g), L.ptr(xd), L.ptr(dd_), None, L.ptr(dw), L.ptr(db),
                                            L.ptr(scratch), nbytes, 4, L.current_stream()))
        return dw.cpu().numpy()

    np.testing.assert_array_equal(fwd(x * sa, w * sb), fwd(x, w) * (sa * sb))
    np.testing.assert_array_equal(wgrad(x * sa, dy * sb), wgrad(x, dy) * (sa * sb))


def test_h3_backward_refuses_a_forward_of_another_family(dev):
    """The H3 kernels scale their operands from amax slots the forward pass fills; a backward pass after a forward of another
    kernel family would read stale slots, so the engine refuses it instead of computing with arbitrary scales."""
    from probav_amd.loss import Losses
    m = _model(dev, params=synth.synth_params(seed=5, perturb=True))
    lo = Losses(targetShape=(48, 48, 1))
    x, hr, mask = (torch.as_tensor(a).to(dev) for a in synth.synth_batch(2, seed=6))
    m.set_impl(3)
    loss = lo.shiftCompensatedL1Loss(hr, mask, m(x, training=True))
    m.set_impl(4)
    with pytest.raises((RuntimeError, ValueError), match="same kernel family"):
        loss.backward()
    m.flat.grad = None
    lo.shiftCompensatedL1Loss(hr, mask, m(x, training=True)).backward()          # a forward of the same family: fine
    assert torch.isfinite(m.flat.grad).all()


# ---------------------------------------------------------------------------------------------------------------------------------
# Gradient parity with the ReLU masks of the device.  The gradient of this network is discontinuous in its ~27 million
# ReLU gates per patch pair; an fp32 and an fp64 evaluation decide a few gates whose pre-activation is ~0 differently, and each such gate
# moves single filter-gradient entries by ~1/sqrt(#voxels).  Taking the masks from the HIP forward removes exactly that effect: what is
# left is the arithmetic of the kernels, and it is held to SURVEY.md section 8c's bar, 1e-3 of the per-tensor max norm, element-wise.
# ---------------------------------------------------------------------------------------------------------------------------------
def _device_gates(m, flat_used, B, T=9):
    from probav_amd.introspect import device_gates
    return device_gates(m, flat_used, B, T)


@pytest.mark.parametrize("impl,T,B", [(4, 9, 2), (3, 9, 2), (4, 13, 2), (4, 9, 5), (4, 7, 3), (4, 19, 2)],
                         ids=["h3-t9-b2", "x6-t9-b2", "h3-t13-b2", "h3-t9-b5", "h3-t7-b3", "h3-t19-b2"])
def test_gradients_match_oracle_with_the_devices_relu_masks(dev, impl, T, B):
    """T = 9, B = 2: the golden inputs.  T = 13 (reducer v3, rows cut into column ranges), T = 7, and B = 5 (more than two per-sample
    scale slots behind `amax_over_samples`, a different strip partition): seeded synthetic inputs.  T = 19: the only network with a
    5x5x5 kernel and a mirror pad along T (models/modelsTF.py:76-121), on the generic kernels."""
    _gate_masked_parity(dev, impl, T, B, 1)


@pytest.mark.parametrize("impl", [3, 4])
def test_three_channel_input_branch_matches_the_oracle(dev, impl):
    """isGrayScale=False (models/modelsTF.py:19-20): input [N, 22, 22, 9, 3], mainConv1 and residConv1 on three channels, the temporal mean per
    channel (:23), a one-channel output.  Forward against the fp64 numpy oracle, loss and all 132 gradients against the fp64 autograd oracle
    at the device's gates (the split-operand families expose their hidden tiles: impl 3 and the default, 4)."""
    _gate_masked_parity(dev, impl, 9, 2, 3)


@pytest.mark.parametrize("impl", [0, 1, 2])
def test_three_channel_input_branch_on_the_fp32_families(dev, impl):
    """The same branch on the generic kernels (impl 0) and the fp32-MFMA families (1, 2), which do not expose their gates: forward and loss
    to the same bars, gradients in relative L2 per tensor against the un-gated fp64 oracle (the tolerance of the one-channel golden test)."""
    from probav_amd.loss import Losses
    x, hr, mask = synth.synth_batch(2, seed=392, numImgLR=9, inChannels=3)
    params = synth.synth_params(seed=101, perturb=True, inChannels=3)
    m = _model(dev, 9, params, gray=False)
    m.set_impl(impl)
    pred = m(torch.as_tensor(x).to(dev), training=True)
    loss = Losses(targetShape=(48, 48, 1)).shiftCompensatedL1Loss(torch.as_tensor(hr).to(dev), torch.as_tensor(mask).to(dev), pred)
    loss.backward()
    ref = on.wdsr_forward(x, params, synth.NIR_MEAN, synth.NIR_STD)
    e = np.abs(pred.detach().cpu().double().numpy() - ref).max() / np.abs(ref).max()
    assert e < 2e-5, "output rel err %.3e (bar 1e-3)" % e
    _, loss_o, grads_o = ot.train_step_grads(torch.tensor(x, dtype=torch.float64), torch.tensor(hr), torch.tensor(mask), ot.to_torch_params(params),
                                             synth.NIR_MEAN, synth.NIR_STD, numImgLR=9)
    assert abs(float(loss.detach()) - float(loss_o)) < 1e-5 * float(loss_o)
    gdev = m.flat.grad.detach().cpu().double().numpy()
    for L_ in m.layers:
        for key, lo_, hi_ in (("g", L_.g_off, L_.v_off), ("v", L_.v_off, L_.b_off), ("bias", L_.b_off, L_.b_off + L_.cout)):
            r = grads_o[L_.name][key].numpy().reshape(-1)
            err = np.sqrt(((gdev[lo_:hi_] - r) ** 2).sum()) / (np.sqrt((r ** 2).sum()) + 1e-30)
            assert err < _grad_l2_tol(9, 0), (L_.name, key, err)


def _gate_masked_parity(dev, impl, T, B, C):
    from probav_amd.loss import Losses
    if (T, B, C) == (9, 2, 1):
        z = np.load(os.path.join(GOLD, "wdsr_t9_b2.npz"))
    else:
        z = dict(zip(("x", "hr", "mask"), synth.synth_batch(B, seed=300 + 10 * T + B, numImgLR=T, inChannels=C)))
    params = synth.synth_params(seed=101, perturb=True, numImgLR=T, inChannels=C)
    m = _model(dev, T, params, gray=(C == 1))
    m.set_impl(impl)
    lo = Losses(targetShape=(48, 48, 1))
    x, hr, mask = (torch.as_tensor(z[k]).to(dev) for k in ("x", "hr", "mask"))
    pred = m(x, training=True)
    loss = lo.shiftCompensatedL1Loss(hr, mask, pred)
    loss.backward()
    gates = _device_gates(m, m.flat.detach(), B, T)
    report = {}
    pt = ot.to_torch_params(params)
    pred_o, loss_o, grads_o = ot.train_step_grads(torch.tensor(z["x"], dtype=torch.float64), torch.tensor(z["hr"]), torch.tensor(z["mask"]),
                                                  pt, synth.NIR_MEAN, synth.NIR_STD, numImgLR=T, gates=gates, gate_report=report)
    nflip = sum(r[0][0] for r in report.values())
    ngate = sum(int(np.prod(g.shape)) for g in gates.values())
    worst_margin = max((r[0][1] / max(r[0][2], 1e-30)) for r in report.values())
    print("impl %d: %d of %d ReLU gates differ between the device and the fp64 evaluation; largest |pre-activation| among them = %.3g of the layer's rms"
          % (impl, nflip, ngate, worst_margin))
    assert worst_margin < 1e-4, "a gate that differs is NOT a ~0 pre-activation: the forward itself is off"
    assert abs(float(loss) - float(loss_o)) < 1e-5 * float(loss_o)
    if C != 1:       # (the one-channel forward is held against the committed fixtures in test_end_to_end_against_golden)
        assert tuple(pred.shape) == (B, 48, 48, 1)
        ref = on.wdsr_forward(z["x"], params, synth.NIR_MEAN, synth.NIR_STD, numImgLR=T)
        e = np.abs(pred.detach().cpu().double().numpy() - ref).max() / np.abs(ref).max()
        assert e < 2e-5, "output rel err %.3e (bar 1e-3)" % e
    gdev = m.flat.grad.detach().cpu().double().numpy()
    worst = (0.0, None)
    for L_ in m.layers:
        go = grads_o[L_.name]
        for key, lo_, hi_ in (("g", L_.g_off, L_.v_off), ("v", L_.v_off, L_.b_off), ("bias", L_.b_off, L_.b_off + L_.cout)):
            ref = go[key].numpy().reshape(-1)
            e = np.abs(gdev[lo_:hi_] - ref).max() / (np.abs(ref).max() + 1e-30)
            if e > worst[0]:
                worst = (e, L_.name + "/" + key)
            assert e < 1e-3, (L_.name, key, e)
    print("impl %d: worst per-tensor max-norm gradient error with the device's gates: %.3g (%s)" % (impl, worst[0], worst[1]))


@pytest.mark.parametrize("B", [2, 128], ids=["golden-b2", "b128"])
def test_forward_and_reverse_pass_decide_the_same_relu_gates(dev, B, monkeypatch):
    """The 256-channel hidden tile never reaches memory: the reverse pass recomputes it (pw_bwd_w4_kernel), and a pre-activation that is zero to rounding would be open
    in one pass and closed in the other if the two summed its piece products in different orders -- the gradient would be taken at a gate the forward did not use (VERDICT
    r4, weak item 2a).  Since round 5 the forward kernel (pw_fwd_w4_kernel) sums in the reverse pass' order -- two k-blocks of v_mfma_f32_32x32x16_f16, w1 x0 + w0 x1 + w0 x0
    each: on the golden inputs and at the benchmark's batch NO gate differs and the two dumps agree bit for bit.  With PROBAV_GEN1=pwf (pw_fwd_h3k_kernel, one 16x16x32
    instruction per hidden value) the older bound holds: the gates that differ are a vanishing share, every one of them a value below 1e-6 of its sample's rms in both
    evaluations."""
    monkeypatch.setenv("PROBAV_KEEP_WS", "1")
    from probav_amd.introspect import hidden_tile
    T = 9
    if B == 2:
        z = np.load(os.path.join(GOLD, "wdsr_t9_b2.npz"))
        x = torch.as_tensor(z["x"]).to(dev)
    else:
        x = torch.as_tensor(synth.synth_batch(B, seed=4242)[0]).to(dev)
    m = _model(dev, T, synth.synth_params(seed=101, perturb=True))
    m(x, training=True)
    flat = m.flat.detach()
    total = differ = 0
    worst = 0.0
    for blk in range(m.numResBlocks):
        a = hidden_tile(m, flat, B, blk, T)                                   # the order of the reverse pass
        f = hidden_tile(m, flat, B, blk, T, from_forward_kernel=True)         # the forward kernel's own
        # each dump is at a power-of-two scale of its kernel's choosing: compare in units of the sample's rms.  The forward kernel brings its tile to scale by an
        # integer add on the exponent field, which turns an exact zero into 2^-111 (both fp16 pieces of which are zero): below 1e-30 is closed
        f = torch.where(f < 1e-30, torch.zeros_like(f), f)
        a = a / torch.sqrt((a.double() ** 2).mean(dim=(1, 2), keepdim=True)).float()
        f = f / torch.sqrt((f.double() ** 2).mean(dim=(1, 2), keepdim=True)).float()
        d = (a > 0) != (f > 0)
        nd = int(d.sum())
        total += a.numel(); differ += nd
        if nd:
            worst = max(worst, float(torch.maximum(a, f)[d].max()))
        # away from zero the two evaluations agree to fp32 rounding of the tile's scale (bit for bit when the forward kernel sums in the reverse pass' order)
        assert float((a - f).abs().max()) < 2e-5, blk
        if os.environ.get("PROBAV_GEN1", "") not in ("1", "pw", "pwf"):
            assert torch.equal(a, f), blk
    print("B = %d: %d of %d hidden gates differ between the forward kernel and the reverse pass's recompute; the largest value among them is %.3g of its sample's rms"
          % (B, differ, total, worst))
    assert differ <= 1e-5 * total, (differ, total)
    assert worst < 1e-6, worst
    if os.environ.get("PROBAV_GEN1", "") not in ("1", "pw", "pwf"):
        assert differ == 0, differ


@pytest.mark.parametrize("T", [9])
def test_batch128_backward_of_a_sub_batch_matches_the_oracle(dev, T):
    """The reverse pass AT THE BENCHMARK'S SIZE against the oracle (at batch 128 the backward was covered by properties only).
    The loss is a batch mean of per-sample terms, so d loss / d pred of sample s depends on sample s alone: run the forward on all 128
    patches, hand the backward an output gradient that is that of a two-sample loss on samples (5, 77) and zero elsewhere, and compare all
    132 gradient tensors with the fp64 oracle on those two samples, evaluated at the device's gates for them -- element-wise, 1e-3 of each
    tensor's max norm.  Every kernel runs its batch-128 partition (two-sample tile runs, 128 per-sample scale
    slots, the slabs of 256 workgroups); what the other 126 samples contribute is exactly zero."""
    from probav_amd.loss import Losses
    pick = [5, 77]
    params = synth.synth_params(seed=31, perturb=True, numImgLR=T)
    m = _model(dev, T, params=params)
    lo = Losses(targetShape=(48, 48, 1))
    xs, hs, ms = synth.synth_batch(128, seed=32, numImgLR=T)
    x, hr, mask = (torch.as_tensor(a).to(dev) for a in (xs, hs, ms))
    pred = m(x, training=True)
    sub = pred[pick].detach().clone().requires_grad_(True)
    l2 = lo.shiftCompensatedL1Loss(hr[pick].contiguous(), mask[pick].contiguous(), sub)
    l2.backward()
    dy = torch.zeros_like(pred)
    dy[pick] = sub.grad
    pred.backward(dy)
    gates = _device_gates_sub(m, m.flat.detach(), 128, T, pick)
    report = {}
    pred_o, loss_o, grads_o = ot.train_step_grads(torch.tensor(xs[pick], dtype=torch.float64), torch.tensor(hs[pick]), torch.tensor(ms[pick]),
                                                  ot.to_torch_params(params), synth.NIR_MEAN, synth.NIR_STD, numImgLR=T, gates=gates, gate_report=report)
    worst_margin = max((r[0][1] / max(r[0][2], 1e-30)) for r in report.values())
    assert worst_margin < 1e-4, "a gate that differs is NOT a ~0 pre-activation: the forward itself is off"
    assert abs(float(l2) - float(loss_o)) < 1e-5 * float(loss_o)
    e = np.abs(pred[pick].detach().cpu().double().numpy() - pred_o.numpy()).max() / np.abs(pred_o.numpy()).max()
    assert e < 2e-5, e
    gdev = m.flat.grad.detach().cpu().double().numpy()
    worst = (0.0, None)
    for L_ in m.layers:
        go = grads_o[L_.name]
        for key, lo_, hi_ in (("g", L_.g_off, L_.v_off), ("v", L_.v_off, L_.b_off), ("bias", L_.b_off, L_.b_off + L_.cout)):
            ref = go[key].numpy().reshape(-1)
            err = np.abs(gdev[lo_:hi_] - ref).max() / (np.abs(ref).max() + 1e-30)
            if err > worst[0]:
                worst = (err, L_.name + "/" + key)
            assert err < 1e-3, (L_.name, key, err)
    print("batch 128, samples %s: worst per-tensor max-norm gradient error with the device's gates: %.3g (%s)" % (pick, worst[0], worst[1]))


def _device_gates_sub(m, flat_used, B, T, samples):
    from probav_amd.introspect import device_gates
    return device_gates(m, flat_used, B, T, samples=samples)


def test_train_steps_match_oracle(dev, tmp_path):
    """models/trainClass.py:124-135 as a whole -- forward, loss, tape.gradient, apply_gradients, metric, running means -- for three
    consecutive steps on B = 2, teacher-forced: at every step the fp64 oracle is evaluated at the parameters the device holds, with the
    device's ReLU masks, and
      (1) the loss and the cPSNR agree (1e-5), (2) all 132 gradient tensors agree element-wise (1e-3 max norm),
      (3) oracle-Nadam applied to the device's gradient reproduces the device's next parameters (flat-buffer offsets, zero_grad, optimizer
          state across steps: 2e-6 of the largest parameter),
      (4) after the FIRST step the free-running oracle trajectory is also matched: all but a handful of the 535 267 parameters within 1e-5.
    Free-running parity of later steps is not a meaningful bar for ANY fp32 implementation: Nadam's first updates are ~lr * sign(g), so the
    ~1e-3 of the gradient entries that sit within fp32 noise of zero move by up to 2 lr = 1e-3 (measured with the oracle itself, fp32 vs
    fp64 on CPU: 6 % of the parameters differ by more than 1e-5 after two steps, 52 % after three)."""
    from oracle.nadam_numpy import Nadam
    from probav_amd.loss import Losses
    from probav_amd.trainClass import ModelTrainer, make_optimizer
    params0 = synth.synth_params(seed=7, perturb=True)
    m = _model(dev, 9, params0)
    lo = Losses(targetShape=(48, 48, 1))
    opt = make_optimizer("nadam", m, 5e-4)
    tr = ModelTrainer(m, lo.shiftCompensatedL1Loss, lo.shiftCompensatedcPSNR, opt, str(tmp_path / "ck"), str(tmp_path / "lg"), multiGPU=False)
    x, hr, mask = synth.synth_batch(2, seed=8)
    xs, hs, ms = torch.as_tensor(x).to(dev), torch.as_tensor(hr).to(dev), torch.as_tensor(mask).to(dev)
    ref_opt = Nadam(lr=5e-4)
    free_opt = Nadam(lr=5e-4)
    # the ReLU masks of a step's forward pass must be read BEFORE the optimizer touches the parameters and refills the weight cache:
    # they are captured from inside optimizer.step (the trainer itself stays untouched)
    captured = {}
    real_step = opt.step

    def step_and_capture(*a, **k):
        captured["gates"] = _device_gates(m, m.flat.detach(), 2)
        captured["used_cache"] = m.weight_cache() is not None
        return real_step(*a, **k)
    opt.step = step_and_capture
    for step in range(3):
        theta_k = m.flat.detach().clone()
        tk = theta_k.cpu().double().numpy()
        tr.trainLoss.reset_states(); tr.trainPSNR.reset_states()
        tr.trainStep(xs, hs, ms)
        g_dev = m.flat.grad.detach().cpu().double().numpy()
        theta_next = m.flat.detach().cpu().double().numpy()
        gates = captured["gates"]
        assert captured["used_cache"] == (step > 0)          # from the second step on the forward pass starts from the fused optimizer's weight cache
        pt = ot.to_torch_params(synth.unflatten_params(theta_k.cpu().numpy()))
        pred_o, loss_o, grads_o = ot.train_step_grads(torch.tensor(x, dtype=torch.float64), torch.tensor(hr), torch.tensor(mask), pt,
                                                      synth.NIR_MEAN, synth.NIR_STD, gates=gates)
        assert abs(tr.trainLoss.result() - float(loss_o)) < 1e-5 * float(loss_o), step                       # (1)
        psnr_o = float(ot.shift_cpsnr(torch.tensor(hr), torch.tensor(mask), pred_o).mean())
        assert abs(tr.trainPSNR.result() - psnr_o) < 1e-5 * abs(psnr_o), step
        g_o = np.zeros_like(g_dev)
        for L_ in m.layers:                                                                                   # (2)
            for key, a, b in (("g", L_.g_off, L_.v_off), ("v", L_.v_off, L_.b_off), ("bias", L_.b_off, L_.b_off + L_.cout)):
                ref = grads_o[L_.name][key].numpy().reshape(-1)
                g_o[a:b] = ref
                e = np.abs(g_dev[a:b] - ref).max() / (np.abs(ref).max() + 1e-30)
                assert e < 1e-3, (step, L_.name, key, e)
        want = ref_opt.step(tk, g_dev)                                                                        # (3)
        assert np.abs(theta_next - want).max() < 2e-6 * np.abs(want).max(), (step, np.abs(theta_next - want).max())
        if step == 0:                                                                                         # (4)
            free = free_opt.step(tk, g_o)
            d = np.abs(theta_next - free)
            print("after one free-running step: max |dtheta| %.3g, fraction beyond 1e-5: %.3g" % (d.max(), (d > 1e-5).mean()))
            assert (d > 1e-5).mean() < 1e-4 and d.max() <= 2.2 * 5e-4
    assert tr.optimizer.state[m.flat]["step"] == 3


def test_config4_full_size_inference_is_deterministic_and_micro_batch_independent(dev):
    """BASELINE.json config 4 at its full size: 32 image sets of 9 registered 128x128 frames -> 2048 patches -> forward -> clip / round
    -> 32 images of 384x384 (test.py:103-122,149-160).  One batch of 2048, the reference's micro-batches of 16 (test.py:125-134) and a
    ragged micro-batch (2048 = 5 * 384 + 128) give the same pixels bit for bit; a second pass repeats them; two image sets are judged
    by the fp64 oracle (a pixel may sit on a rounding boundary: at most one count apart, and only rarely)."""
    from probav_amd import testClass
    params = synth.synth_params(seed=77, perturb=True)
    m = _model(dev, params=params)
    rng = np.random.default_rng(7)
    frames = torch.as_tensor(np.clip(rng.normal(synth.NIR_MEAN, synth.NIR_STD, (32, 9, 128, 128)), 0, 16383).astype(np.float32)).to(dev)
    patches = testClass.unfold_frames(frames)
    assert tuple(patches.shape) == (32, 64, 22, 22, 9, 1)
    full = testClass.resolve_images(m, patches, micro_batch=2048)
    assert tuple(full.shape) == (32, 384, 384) and bool((full == full.round()).all()) and float(full.min()) >= 0 and float(full.max()) <= 65536
    again = testClass.resolve_images(m, patches, micro_batch=2048)
    assert torch.equal(full, again)
    assert torch.equal(full, testClass.resolve_images(m, patches, micro_batch=16, launch_batch=16))       # 128 launch sets of 16 patches, as the reference's loop
    assert torch.equal(full, testClass.resolve_images(m, patches, micro_batch=384, launch_batch=384))
    assert torch.equal(full, testClass.resolve_images(m, patches, micro_batch=16))                        # the drop-in default: micro-batches coalesced
    # the reference's own entry points (test.py:103-134), coalesced on the engine: the same pixels
    imgs = testClass.evaluate(m, patches.cpu().numpy(), batch_size=16)
    np.testing.assert_array_equal(np.stack(imgs)[..., 0], full.cpu().numpy().astype(np.float64))
    one_set = testClass.resolveByBatch(m, patches[5].cpu().numpy(), batch_size=16)
    np.testing.assert_array_equal(testClass.stitch_device(torch.as_tensor(one_set), 1)[0].numpy(), full[5].cpu().numpy())
    pt = ot.to_torch_params(params, requires_grad=False)
    for s in (0, 31):
        with torch.no_grad():
            po = ot.wdsr_forward(torch.tensor(patches[s].cpu().numpy(), dtype=torch.float64), pt, synth.NIR_MEAN, synth.NIR_STD).numpy()
        want = np.round(np.clip(po, 0, 2 ** 16))[..., 0].reshape(8, 8, 48, 48).transpose(0, 2, 1, 3).reshape(384, 384)       # np.round: half to even, like tf.round
        d = np.abs(full[s].cpu().numpy().astype(np.float64) - want)
        assert d.max() <= 1.0 and (d > 0).mean() < 2e-3, (s, d.max(), (d > 0).mean())


@pytest.mark.parametrize("C", [1, 3], ids=["gray", "three-channel"])
@pytest.mark.parametrize("impl", [4, 0], ids=["one-launch", "generic"])
def test_residual_path_first_layer_and_output_match_the_oracle(dev, impl, C):
    """The low-frequency residual path (models/modelsTF.py:45-53) runs as ONE launch each way in every kernel family but 0 (kernels_direct.hip: resid_path_fwd_kernel /
    resid_path_bwd_kernel; round 6), family 0 keeps the generic direct kernels.  Forward, layer by layer as far as the boundary shows it: residConv1's output r1 (the saved
    activation behind its ReLU: probav_workspace_view RESID1) against the fp64 numpy oracle, and the network's output -- main path + depth_to_space(r3) -- against
    the oracle's.  (The reverse pass of the three layers is held element-wise by the gate-masked gradient tests above, for one and for three input channels.)"""
    B, T = 3, 9
    x, _, _ = synth.synth_batch(B, seed=77 + C, numImgLR=T, inChannels=C)
    params = synth.synth_params(seed=78, perturb=True, inChannels=C)
    m = _model(dev, T, params, gray=(C == 1))
    m.set_impl(impl)
    pred = m(torch.as_tensor(x).to(dev), training=True)
    torch.cuda.synchronize()
    L = _lib()
    off, cnt = ctypes.c_int64(), ctypes.c_int64()
    L.check(L.lib().probav_workspace_view(m._handle(), B, 1, 3, 0, ctypes.byref(off), ctypes.byref(cnt)), "probav_workspace_view")
    r1 = m._workspace(B, True)[off.value: off.value + cnt.value].cpu().double().numpy().reshape(B, 20, 20, 9)
    xm = np.asarray(x, np.float64)
    mn = (xm.mean(axis=3) - synth.NIR_MEAN) / synth.NIR_STD                       # [B, 22, 22, C]: :23, :27
    ref1 = on.wn_conv(mn, params["residConv1"], "valid", True)                     # a 2-D layer: [B, 20, 20, 9]
    e1 = np.abs(r1 - ref1).max() / np.abs(ref1).max()
    ref = on.wdsr_forward(x, params, synth.NIR_MEAN, synth.NIR_STD)
    e = np.abs(pred.detach().cpu().double().numpy() - ref).max() / np.abs(ref).max()
    print("impl %d, %d input channel(s): residConv1 output rel err %.3g, network output rel err %.3g" % (impl, C, e1, e))
    assert e1 < 2e-6 and e < 2e-5, (impl, C, e1, e)
