"""SURVEY.md section 8b's minimum op set as torch custom ops of their own (probav_amd/ops.py): probav::conv3d_k3_{fwd,bwd_weight},
probav::pw_expand_relu_decay_{fwd,bwd}, probav::wn_weight_{fwd,bwd}.  A WDSR-B residual block (models/modelsTF.py:168-185) and a reducer
stage composed from them -- without the engine's whole-network op -- match an fp64 torch restatement, forward and through autograd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from probav_amd import synth

pytestmark = pytest.mark.gpu


def _conv_ref(x, w, b, pad):
    xp = F.pad(x, (0, 0, pad[2], pad[2], pad[1], pad[1], pad[0], pad[0]))
    return F.conv3d(xp.permute(0, 4, 1, 2, 3), w.permute(4, 3, 0, 1, 2)).permute(0, 2, 3, 4, 1) + b


def _rel(a, b):
    return float((a.double().cpu() - b).abs().max() / b.abs().max())


def test_residual_block_composed_from_layer_ops(dev):
    import probav_amd.ops  # noqa: F401
    rng = np.random.default_rng(5)
    N, hwt, D = 2, (10, 9, 5), 25
    nvps = hwt[0] * hwt[1] * hwt[2]
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    x = t(rng.normal(size=(N,) + hwt + (32,)))
    w1, b1 = t(rng.normal(size=(32, 256)) / 6), t(rng.normal(size=256) * 0.1)
    w2, b2 = t(rng.normal(size=(256, D)) / 16), t(rng.normal(size=D) * 0.1)
    w3, b3 = t(rng.normal(size=(3, 3, 3, D, 32)) / 26), t(rng.normal(size=32) * 0.1)
    w4, b4 = t(rng.normal(size=(3, 3, 3, 32, 32)) / 30), t(rng.normal(size=32) * 0.1)
    r = t(rng.normal(size=(N, hwt[0] - 2, hwt[1] - 2, hwt[2] - 2, 32)))
    leaves = [p.clone().requires_grad_(True) for p in (x, w1, b1, w2, b2, w3, b3, w4, b4)]
    xo, w1o, b1o, w2o, b2o, w3o, b3o, w4o, b4o = leaves
    blk = xo + _conv_ref(torch.relu(xo @ w1o + b1o) @ w2o + b2o, w3o, b3o, (1, 1, 1))          # ResConv3D: exp -> ReLU -> dec -> 3x3x3 same, + skip
    red = torch.relu(_conv_ref(blk, w4o, b4o, (0, 0, 0)))                                            # a reducer stage: 3x3x3 valid + ReLU
    (red * r).sum().backward()

    dl = [p.detach().float().to(dev).requires_grad_(True) for p in (x, w1, b1, w2, b2, w3, b3, w4, b4)]
    xd, w1d, b1d, w2d, b2d, w3d, b3d, w4d, b4d = dl
    for impl in (4, 2):
        for p in dl:
            p.grad = None
        dec = torch.ops.probav.pw_expand_relu_decay_fwd(xd, w1d, b1d, w2d, b2d, nvps, impl)
        blk_d = torch.ops.probav.conv3d_k3_fwd(dec, w3d, b3d, [1, 1, 1], False, False, xd, None, impl)
        red_d = torch.ops.probav.conv3d_k3_fwd(blk_d, w4d, b4d, [0, 0, 0], False, True, None, None, impl)
        assert _rel(blk_d.detach(), blk.detach()) < 2e-6 and _rel(red_d.detach(), red.detach()) < 2e-6
        (red_d * r.float().to(dev)).sum().backward()
        for name, got, want in zip("x w1 b1 w2 b2 w3 b3 w4 b4".split(), dl, leaves):
            e = _rel(got.grad, want.grad)
            assert e < 2e-5, (impl, name, e)


def test_weight_norm_ops_match_the_restated_reparameterisation(dev):
    import probav_amd.ops  # noqa: F401
    from oracle import wdsr_numpy as on
    from probav_amd.modelsTF import WDSRConv3D
    m = WDSRConv3D("t", "NIR", synth.NIR_MEAN, synth.NIR_STD, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True, seed=0)
    params = synth.synth_params(seed=91, perturb=True)
    m.load_variables(params)
    m = m.to(dev)
    eng = int(m._handle().value)
    flat = m.flat.detach().clone().requires_grad_(True)
    weff, weffT, inv = torch.ops.probav.wn_weight_fwd(flat, eng)
    off, gw = 0, np.random.default_rng(3).normal(size=weff.numel())
    want_grad = np.zeros(flat.numel())
    for L in m.layers:
        w = on.weight_norm(params[L.name]["v"], params[L.name]["g"])
        got = weff[off:off + w.size].detach().cpu().double().numpy().reshape(w.shape)
        assert np.abs(got - w).max() < 2e-6 * np.abs(w).max(), L.name
        # gradient of sum(weff * gw) by torch autograd on the restated formula (fp64)
        v = torch.tensor(params[L.name]["v"], dtype=torch.float64, requires_grad=True)
        g = torch.tensor(params[L.name]["g"], dtype=torch.float64, requires_grad=True)
        wt = v * (g / v.reshape(-1, v.shape[-1]).pow(2).sum(0).sqrt())
        (wt * torch.tensor(gw[off:off + w.size].reshape(w.shape))).sum().backward()
        want_grad[L.g_off:L.v_off] = g.grad.numpy()
        want_grad[L.v_off:L.b_off] = v.grad.numpy().reshape(-1)
        off += w.size
    (weff * torch.tensor(gw, dtype=torch.float32, device=dev)).sum().backward()
    got = flat.grad.cpu().double().numpy()
    for L in m.layers:
        for lo, hi in ((L.g_off, L.v_off), (L.v_off, L.b_off)):
            assert np.abs(got[lo:hi] - want_grad[lo:hi]).max() < 2e-5 * np.abs(want_grad[lo:hi]).max(), L.name


def test_opcheck_layer_ops(dev):
    import probav_amd.ops  # noqa: F401
    rng = np.random.default_rng(1)
    t = lambda *s: torch.tensor(rng.normal(size=s), dtype=torch.float32, device=dev)
    x, w, b = t(1, 6, 5, 4, 25), t(3, 3, 3, 25, 32) / 20, t(32)
    tests = ("test_schema", "test_autograd_registration", "test_faketensor")
    torch.library.opcheck(torch.ops.probav.conv3d_k3_fwd.default, (x.requires_grad_(True), w.requires_grad_(True), b, [1, 1, 1], False, False), test_utils=tests)
    torch.library.opcheck(torch.ops.probav.conv3d_k3_bwd_weight.default, (x.detach(), t(1, 6, 5, 4, 32), [3, 3, 3], [1, 1, 1], False), test_utils=tests)
    xp, w1, b1, w2, b2 = t(40, 32), t(32, 256) / 6, t(256), t(256, 25) / 16, t(25)
    torch.library.opcheck(torch.ops.probav.pw_expand_relu_decay_fwd.default, (xp.requires_grad_(True), w1.requires_grad_(True), b1, w2, b2), test_utils=tests)
    torch.library.opcheck(torch.ops.probav.pw_expand_relu_decay_bwd.default, (xp.detach(), t(40, 25), t(40, 32), w1.detach(), b1, w2), test_utils=tests)
