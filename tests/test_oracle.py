"""The CPU oracle against itself and against the committed golden fixtures (no GPU).

Parity is unpinned by the reference (no TF here, no reference tests); what is checked is that the two
independent restatements agree, that the analytic gradients match finite differences, and that the
golden vectors in tests/golden/ are reproduced."""
import os

import numpy as np
import pytest
import torch

from oracle import wdsr_numpy as on
from oracle import wdsr_torch as ot
from probav_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_two_restatements_agree_fp64():
    p = synth.synth_params(seed=3, perturb=True)
    x, hr, m = synth.synth_batch(1, seed=4)
    y = on.wdsr_forward(x, p, synth.NIR_MEAN, synth.NIR_STD)
    yt = ot.wdsr_forward(torch.tensor(x, dtype=torch.float64), ot.to_torch_params(p, requires_grad=False),
                         synth.NIR_MEAN, synth.NIR_STD)
    assert y.shape == (1, 48, 48, 1)
    assert np.abs(y - yt.numpy()).max() <= 1e-9 * np.abs(y).max()
    assert abs(on.shift_l1_loss(hr, m, y) - float(ot.shift_l1_loss(torch.tensor(hr), torch.tensor(m), yt))) < 1e-9
    assert abs(on.shift_l2_loss(hr, m, y) - float(ot.shift_l2_loss(torch.tensor(hr), torch.tensor(m), yt))) < 1e-6
    np.testing.assert_allclose(on.shift_cpsnr(hr, m, y), ot.shift_cpsnr(torch.tensor(hr), torch.tensor(m), yt).numpy(), rtol=1e-12)


def test_two_restatements_agree_on_the_three_channel_branch():
    """isGrayScale=False (models/modelsTF.py:19-20): [N, 22, 22, 9, 3] in, the temporal mean taken per channel (:23), mainConv1 and residConv1
    on three channels, one channel out."""
    assert dict(on.layer_specs(inChannels=3))["mainConv1"] == (3, 3, 3, 3, 32) and dict(on.layer_specs(inChannels=3))["residConv1"] == (3, 3, 3, 9)
    p = synth.synth_params(seed=3, perturb=True, inChannels=3)
    x, hr, m = synth.synth_batch(1, seed=4, inChannels=3)
    assert x.shape == (1, 22, 22, 9, 3)
    y = on.wdsr_forward(x, p, synth.NIR_MEAN, synth.NIR_STD)
    yt = ot.wdsr_forward(torch.tensor(x, dtype=torch.float64), ot.to_torch_params(p, requires_grad=False), synth.NIR_MEAN, synth.NIR_STD)
    assert y.shape == (1, 48, 48, 1)
    assert np.abs(y - yt.numpy()).max() <= 1e-9 * np.abs(y).max()
    # the three channels are really used: another value in channel 2 of one frame moves the output
    x2 = x.copy(); x2[0, 11, 11, 4, 2] += 500.0
    assert np.abs(on.wdsr_forward(x2, p, synth.NIR_MEAN, synth.NIR_STD) - y).max() > 1e-3


def test_loss_gradient_matches_autograd_and_finite_differences():
    rng = np.random.default_rng(0)
    _, hr, m = synth.synth_batch(3, seed=5)
    pred = (hr + rng.normal(0, 200, hr.shape)).astype(np.float64)
    g = on.shift_l1_grad(hr, m, pred)
    pt = torch.tensor(pred, requires_grad=True)
    ot.shift_l1_loss(torch.tensor(hr), torch.tensor(m), pt).backward()
    np.testing.assert_allclose(g, pt.grad.numpy(), atol=1e-15)
    # directional finite difference (the loss is piecewise linear in pred; a small step stays in one piece)
    d = rng.normal(0, 1, pred.shape)
    eps = 1e-4
    fd = (on.shift_l1_loss(hr, m, pred + eps * d) - on.shift_l1_loss(hr, m, pred - eps * d)) / (2 * eps)
    assert abs(fd - (g * d).sum()) < 1e-6 * max(1.0, abs(fd))
    assert np.all(g[:, :3] == 0) and np.all(g[:, :, :3] == 0)          # the 3-pixel border never gets gradient


def test_loss_reference_quirk_hr_not_masked():
    """models/loss.py:146,151: HR is left un-masked, so masked-out pixels contribute |HR|."""
    hr = np.full((1, 48, 48, 1), 100.0)
    pred = np.full((1, 48, 48, 1), 100.0)
    mask = np.ones((1, 48, 48, 1), bool)
    assert on.shift_l1_loss(hr, mask, pred) == 0.0
    mask[0, 10, 10, 0] = False          # one cloudy pixel inside every crop window
    l = on.shift_l1_loss(hr, mask, pred)
    n = 42 * 42 - 1
    b = 100.0 / n                        # bias absorbs the un-masked HR pixel
    assert abs(l - (n * b + 100.0) / n) < 1e-9


def test_weight_norm_and_its_clamp():
    v = np.random.default_rng(1).normal(size=(3, 3, 3, 4, 5))
    g = np.arange(1, 6, dtype=np.float64)
    w = on.weight_norm(v, g)
    np.testing.assert_allclose(np.sqrt((w ** 2).reshape(-1, 5).sum(0)), g, rtol=1e-12)
    tiny = np.zeros((1, 1, 1, 2, 1))
    tiny[..., 0, 0] = 1e-9
    np.testing.assert_allclose(on.weight_norm(tiny, np.ones(1))[0, 0, 0, 0, 0], 1e-9 / 1e-6)   # eps=1e-12 inside rsqrt


def test_depth_to_space_and_reflect_pad_conventions():
    x = np.arange(2 * 2 * 9, dtype=np.float64).reshape(1, 2, 2, 9)
    y = on.depth_to_space(x, 3)
    for h in range(2):
        for w in range(2):
            for i in range(3):
                for j in range(3):
                    assert y[0, 3 * h + i, 3 * w + j, 0] == x[0, h, w, 3 * i + j]
    np.testing.assert_array_equal(y, ot.depth_to_space(torch.tensor(x), 3).numpy())
    a = np.arange(16, dtype=np.float64).reshape(1, 4, 4, 1, 1)
    p = on.reflect_pad_hw(a)
    assert p[0, 0, 0, 0, 0] == a[0, 1, 1, 0, 0] and p[0, 5, 2, 0, 0] == a[0, 2, 1, 0, 0]    # mirror without the edge


@pytest.mark.parametrize("T", [9, 13, 7, 19])
def test_golden_fixture_reproduced(T):
    z = np.load(os.path.join(GOLD, "wdsr_t%d_b2.npz" % T))
    seeds = {9: (101, 102), 13: (131, 132), 7: (71, 72), 19: (191, 192)}[T]
    params = synth.synth_params(seed=seeds[0], perturb=True, numImgLR=T)
    x, hr, mask = synth.synth_batch(2, seed=seeds[1], numImgLR=T)
    flat = synth.flatten_params(params, numImgLR=T).astype(np.float64)
    np.testing.assert_allclose([flat.sum(), (flat ** 2).sum()], z["param_checksum"], rtol=1e-12,
                               err_msg="numpy's generator stream changed: regenerate tests/golden with make_golden.py")
    np.testing.assert_array_equal(x, z["x"])
    np.testing.assert_array_equal(mask, z["mask"])
    pt = ot.to_torch_params(params)
    pred, loss, grads = ot.train_step_grads(torch.tensor(x, dtype=torch.float64), torch.tensor(hr), torch.tensor(mask),
                                            pt, synth.NIR_MEAN, synth.NIR_STD, numImgLR=T)
    np.testing.assert_allclose(pred.numpy(), z["pred"], rtol=1e-10)
    assert abs(float(loss) - float(z["loss_l1"])) < 1e-9 * float(z["loss_l1"])
    np.testing.assert_allclose(grads["mainConv1"]["v"].numpy(), z["grad/mainConv1/v"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(on.shift_cpsnr(hr, mask, z["pred"]), z["cpsnr"], rtol=1e-12)


def test_nadam_restatement_equals_torch_nadam():
    """Keras Nadam restated (oracle/nadam_numpy.py) == torch.optim.NAdam(eps=1e-7, momentum_decay=0.004)."""
    from oracle.nadam_numpy import Nadam
    rng = np.random.default_rng(0)
    theta = rng.normal(size=1000)
    p = torch.nn.Parameter(torch.tensor(theta))
    opt_t = torch.optim.NAdam([p], lr=5e-4, betas=(0.9, 0.999), eps=1e-7, momentum_decay=0.004)
    opt_n = Nadam(lr=5e-4)
    for _ in range(6):
        g = rng.normal(size=1000) * 10
        p.grad = torch.tensor(g)
        opt_t.step()
        theta = opt_n.step(theta, g)
    np.testing.assert_allclose(p.detach().numpy(), theta, rtol=1e-7, atol=1e-10)     # same rule, different fp64 op order


def test_sobel_l1_mix_restatement_conventions():
    """oracle/wdsr_torch.py::shift_l1edge_loss: Sobel of a constant is 0 (reflect padding), of a unit ramp 8 in the interior;
    with pred == hr and a full mask the loss is 0; pi = 1 reduces it to the shift-compensated L1."""
    hr = np.tile(np.arange(48, dtype=np.float64).reshape(1, 1, 48, 1), (1, 48, 1, 1)) * 10.0     # ramp along W
    mask = np.ones((1, 48, 48, 1), bool)
    z = ot.shift_l1edge_loss(torch.tensor(hr), torch.tensor(mask), torch.tensor(hr))
    assert float(z) == 0.0
    rng = np.random.default_rng(3)
    _, hr2, m2 = synth.synth_batch(2, seed=9)
    pred = hr2 + rng.normal(0, 100, hr2.shape)
    a = ot.shift_l1edge_loss(torch.tensor(hr2), torch.tensor(m2), torch.tensor(pred), pi=1.0)
    b = ot.shift_l1_loss(torch.tensor(hr2), torch.tensor(m2), torch.tensor(pred))
    assert abs(float(a) - float(b)) < 1e-9 * float(b)
    # a flat +c offset is absorbed by the brightness bias: loss unchanged
    c = ot.shift_l1edge_loss(torch.tensor(hr2), torch.tensor(m2), torch.tensor(pred + 37.0))
    d = ot.shift_l1edge_loss(torch.tensor(hr2), torch.tensor(m2), torch.tensor(pred))
    assert abs(float(c) - float(d)) < 1e-9 * float(d)


# ---- third implementations: library routines that were written by neither the reference's authors nor this repository's ------------
def test_depth_to_space_equals_torch_pixel_shuffle():
    """tf.nn.depth_to_space(x, 3) with ONE output channel (models/modelsTF.py:52,73) is torch's pixel_shuffle on the NCHW view."""
    import torch.nn.functional as F
    x = np.random.default_rng(2).normal(size=(3, 16, 16, 9))
    want = F.pixel_shuffle(torch.tensor(x).permute(0, 3, 1, 2), 3).permute(0, 2, 3, 1).numpy()
    np.testing.assert_array_equal(on.depth_to_space(x, 3), want)
    np.testing.assert_array_equal(ot.depth_to_space(torch.tensor(x), 3).numpy(), want)


def test_weight_norm_equals_torch_parametrization_away_from_the_clamp():
    """TFA WeightNormalization (g * v / ||v||, norm over all kernel axes but the output one) == torch's weight_norm
    parametrization with dim = the output axis; the two differ only at ||v||^2 < 1e-12 (TFA's l2_normalize epsilon)."""
    from torch.nn.utils.parametrizations import weight_norm
    rng = np.random.default_rng(5)
    v = rng.normal(size=(3, 3, 3, 25, 32))
    g = rng.uniform(0.5, 2.0, size=32)
    conv = torch.nn.Conv3d(25, 32, 3, bias=False).double()            # torch layout [Cout, Cin, kh, kw, kt]
    conv = weight_norm(conv, name="weight", dim=0)
    with torch.no_grad():
        conv.parametrizations.weight.original1.copy_(torch.tensor(v).permute(4, 3, 0, 1, 2))
        conv.parametrizations.weight.original0.copy_(torch.tensor(g).reshape(32, 1, 1, 1, 1))
    want = conv.weight.detach().permute(2, 3, 4, 1, 0).numpy()
    np.testing.assert_allclose(on.weight_norm(v, g), want, rtol=1e-13)
    # and its gradient (dg, dv) against autograd through the parametrization
    dw = rng.normal(size=v.shape)
    (conv.weight * torch.tensor(dw).permute(4, 3, 0, 1, 2)).sum().backward()
    vt = torch.tensor(v, requires_grad=True)
    gt = torch.tensor(g, requires_grad=True)
    (ot.weight_norm(vt, gt) * torch.tensor(dw)).sum().backward()
    np.testing.assert_allclose(vt.grad.numpy(), conv.parametrizations.weight.original1.grad.permute(2, 3, 4, 1, 0).numpy(), rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(gt.grad.numpy(), conv.parametrizations.weight.original0.grad.reshape(-1).numpy(), rtol=1e-10)


def test_conv_conventions_equal_torch_conv3d():
    """Keras Conv3D 'same' / 'valid', stride 1, cross-correlation, kernel [kh,kw,kt,Cin,Cout] on [N,H,W,T,C] ==
    torch.nn.functional.conv3d on the permuted tensors; tf.pad(REFLECT) == torch 'reflect' padding."""
    import torch.nn.functional as F
    rng = np.random.default_rng(6)
    x = rng.normal(size=(2, 6, 6, 5, 4))
    w = rng.normal(size=(3, 3, 3, 4, 7))
    b = rng.normal(size=7)
    for pad in (1, 0):
        want = F.conv3d(torch.tensor(x).permute(0, 4, 1, 2, 3), torch.tensor(w).permute(4, 3, 0, 1, 2), torch.tensor(b), padding=pad).permute(0, 2, 3, 4, 1).numpy()
        got = on.conv_valid(on.pad_zero_same(x, w) if pad else x, w) + b          # the numpy tap loops never call into torch
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
    N, H, W, T, C = x.shape
    xt = torch.tensor(x).permute(0, 3, 4, 1, 2).reshape(N, T * C, H, W)                       # 2-D reflect pad over (H, W)
    want = F.pad(xt, (1, 1, 1, 1), mode="reflect").reshape(N, T, C, H + 2, W + 2).permute(0, 3, 4, 1, 2).numpy()
    np.testing.assert_array_equal(on.reflect_pad_hw(x), want)
