"""torch (CPU) restatement of the reference's WDSR-B Conv3D network and shift-compensated losses.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- PARITY UNPINNED.

Second, independent formulation of oracle/wdsr_numpy.py: convolutions go through
``torch.nn.functional.conv3d/conv2d`` on permuted tensors and gradients come from autograd
(the stand-in for ``tf.GradientTape`` at models/trainClass.py:126-131).  It runs in fp64 as the
gradient oracle and in fp32 on all host cores as the timed CPU baseline of bench.py.
"""
import torch
import torch.nn.functional as F


def reducer_plan(numImgLR):
    """models/modelsTF.py:62-69, :123-175 (see oracle/wdsr_numpy.reducer_plan)."""
    a, b = (3, 1, 0), (3, 0, 0)
    return {9: [a, b, b], 13: [a, a, a, b, b], 7: [b, b],
            19: [(5, 2, 2), (3, 2, 1), (3, 2, 0), (3, 2, 0), a, b, b, b, b, b]}[numImgLR]


def weight_norm(v, g):
    """tf.nn.l2_normalize(v, all-but-last) * g  (models/modelsTF.py:191-197)."""
    ss = (v * v).reshape(-1, v.shape[-1]).sum(dim=0)
    return v * torch.rsqrt(torch.clamp(ss, min=1e-12)) * g


def _conv(x, w, b, same):
    """x [N,H,W,T,C] / [N,H,W,C] channels-last as in the reference; w Keras layout."""
    if w.dim() == 5:
        y = F.conv3d(x.permute(0, 4, 1, 2, 3), w.permute(4, 3, 0, 1, 2), b,
                     padding=tuple((k - 1) // 2 for k in w.shape[:3]) if same else 0)
        return y.permute(0, 2, 3, 4, 1)
    y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), b,
                 padding=tuple((k - 1) // 2 for k in w.shape[:2]) if same else 0)
    return y.permute(0, 2, 3, 1)


def wn_conv(x, p, padding, relu, gate=None, report=None):
    """gate (optional, ReLU layers only): a boolean tensor used INSTEAD of the layer's own sign test, y * gate -- the ReLU masks of
    another evaluation of the same network (tests: the masks the HIP forward produced).  report (optional list) then receives
    (#gates that differ from this evaluation's own, largest |pre-activation| among them, rms of the pre-activations)."""
    y = _conv(x, weight_norm(p["v"], p["g"]), p["bias"], padding == "same")
    if relu and gate is not None:
        g = torch.as_tensor(gate, dtype=torch.bool).reshape(y.shape)
        if report is not None:
            flip = g != (y > 0)
            yd = y.detach()
            report.append((int(flip.sum()), float(yd[flip].abs().max()) if bool(flip.any()) else 0.0, float(yd.pow(2).mean().sqrt())))
        return y * g
    return torch.relu(y) if relu else y


def depth_to_space(x, s):
    """tf.nn.depth_to_space NHWC == pixel_shuffle on NCHW with Co = C/s^2 (channel = (i*s+j)*Co+c;
    for Co = 1 the two orderings coincide)."""
    N, H, W, C = x.shape
    co = C // (s * s)
    return x.reshape(N, H, W, s, s, co).permute(0, 1, 3, 2, 4, 5).reshape(N, H * s, W * s, co)


def wdsr_forward(x, params, mean, std, numResBlocks=12, numImgLR=9, scale=3, gates=None, gate_report=None):
    """WDSRConv3D.build graph (models/modelsTF.py:15-43).
    gates (optional): {layer name: bool array} -- ReLU masks taken from another evaluation (see wn_conv); gate_report: dict
    receiving, per gated layer, (#differing gates, max |pre-activation| among them, rms pre-activation)."""
    def G(name):
        return None if gates is None else gates.get(name)

    def R(name):
        if gate_report is None or gates is None or name not in gates:
            return None
        return gate_report.setdefault(name, [])
    meanLR = x.mean(dim=3)                                     # :23
    xn = (x - mean) / std                                      # :26
    mn = (meanLR - mean) / std                                 # :27
    h = wn_conv(xn, params["mainConv1"], "same", True, G("mainConv1"), R("mainConv1"))         # :58
    for i in range(numResBlocks):                              # :177-189
        e = wn_conv(h, params["expConv_%d" % i], "same", True, G("expConv_%d" % i), R("expConv_%d" % i))
        d = wn_conv(e, params["decConv_%d" % i], "same", False)
        h = wn_conv(d, params["normConv_%d" % i], "same", False) + h
    for i, (_, pad, pad_t) in enumerate(reducer_plan(numImgLR)):   # :152-164, :76-121
        if pad or pad_t:                                       # tf.pad(mode='reflect') on H, W (and T)
            h = F.pad(h.permute(0, 4, 1, 2, 3), (pad_t, pad_t, pad, pad, pad, pad), mode="reflect").permute(0, 2, 3, 4, 1)
        name = "convReducer_%d" % (i + 1)
        h = wn_conv(h, params[name], "valid", True, G(name), R(name))
    h = wn_conv(h, params["upscaleConv1"], "valid", False)     # :162-163
    main = depth_to_space(h[:, :, :, 0, :], scale)             # :71-73
    r = wn_conv(mn, params["residConv1"], "valid", True, G("residConv1"), R("residConv1"))       # :45-53
    r = wn_conv(r, params["residConv2"], "valid", False)
    r = wn_conv(r, params["residConv3"], "valid", False)
    return (main + depth_to_space(r, scale)) * std + mean      # :38, :41


def _shift_terms(hr, mask, pred, cropBorder=3):
    """models/loss.py:140-180 scaffolding -> l1[49,B], mse[49,B]."""
    dt = pred.dtype
    hr = hr.to(dt)[..., 0]
    m = mask.to(dt)[..., 0]
    p = pred[..., 0]
    S = hr.shape[1]
    c = cropBorder
    L = S - 2 * c
    P = p[:, c:c + L, c:c + L]
    l1, l2 = [], []
    for i in range(2 * c + 1):
        for j in range(2 * c + 1):
            H = hr[:, i:i + L, j:j + L]
            M = m[:, i:i + L, j:j + L]
            n = M.sum(dim=(1, 2))
            b = (1.0 / n) * (H - P * M).sum(dim=(1, 2))
            C = (P + b[:, None, None]) * M
            l1.append((1.0 / n) * (H - C).abs().sum(dim=(1, 2)))
            l2.append((1.0 / n) * (H - C).square().sum(dim=(1, 2)))
    return torch.stack(l1), torch.stack(l2)


def shift_l1_loss(hr, mask, pred, cropBorder=3):
    """models/loss.py:73-84.  torch.amin splits the gradient equally among ties like tf.reduce_min."""
    return _shift_terms(hr, mask, pred, cropBorder)[0].amin(dim=0).mean()


def shift_l2_loss(hr, mask, pred, cropBorder=3):
    """models/loss.py:55-71."""
    return _shift_terms(hr, mask, pred, cropBorder)[1].amin(dim=0).mean()


def shift_cpsnr(hr, mask, pred, cropBorder=3, bitDepth=16):
    """models/loss.py:37-53, 234-238."""
    l2 = _shift_terms(hr, mask, pred, cropBorder)[1]
    nb = float(2 ** bitDepth - 1)
    ten = torch.log(torch.tensor(10.0, dtype=l2.dtype))
    return (10.0 * (torch.log(nb * nb / l2) / ten)).amax(dim=0)


def to_torch_params(params_np, dtype=torch.float64, requires_grad=True):
    out = {}
    for name, p in params_np.items():
        out[name] = {k: torch.tensor(v, dtype=dtype, requires_grad=requires_grad) for k, v in p.items()}
    return out


def train_step_grads(x, hr, mask, params, mean, std, **kw):
    """One `trainStep` up to the gradients (models/trainClass.py:124-131): forward, L1 loss,
    d loss / d every trainable.  Returns (pred, loss, grads{name:{v,g,bias}})."""
    pred = wdsr_forward(x, params, mean, std, **kw)
    loss = shift_l1_loss(hr, mask, pred)
    leaves = [t for p in params.values() for t in (p["g"], p["v"], p["bias"])]
    gl = torch.autograd.grad(loss, leaves)
    grads, k = {}, 0
    for name in params:
        grads[name] = {"g": gl[k], "v": gl[k + 1], "bias": gl[k + 2]}
        k += 3
    return pred.detach(), loss.detach(), grads


def shift_l1edge_loss(hr, mask, pred, border=3, pi=0.7):
    """cfg loss = sobel_l1_mix (models/loss.py:86-97, 126-137, 214-219): min over the shifts of
    pi * L1 + (1 - pi) * sum |tf.image.sobel_edges(HR) - sobel_edges(corrected SR)| / n, batch mean.
    tf.image.sobel_edges = depthwise 3x3 cross-correlation of the REFLECT-padded image with
    [[-1,-2,-1],[0,0,0],[1,2,1]] (dy) and its transpose (dx)."""
    import torch.nn.functional as F
    hr, pred = hr.to(torch.float64), pred.to(torch.float64)
    m = mask.to(torch.float64)
    S = pred.shape[1]
    L = S - 2 * border
    ky = torch.tensor([[-1., -2., -1.], [0., 0., 0.], [1., 2., 1.]], dtype=torch.float64)
    k = torch.stack([ky, ky.t()]).unsqueeze(1)                      # [2,1,3,3]

    def sobel(img):                                                  # [B,L,L,1] -> [B,2,L,L]
        return F.conv2d(F.pad(img.permute(0, 3, 1, 2), (1, 1, 1, 1), mode="reflect"), k)

    cp = pred[:, border:border + L, border:border + L]
    cands = []
    for i in range(2 * border + 1):
        for j in range(2 * border + 1):
            h, mm = hr[:, i:i + L, j:j + L], m[:, i:i + L, j:j + L]
            n = mm.sum(dim=(1, 2, 3))
            b = ((h - cp * mm).sum(dim=(1, 2, 3)) / n).reshape(-1, 1, 1, 1)
            c = (cp + b) * mm
            l1 = (h - c).abs().sum(dim=(1, 2, 3)) / n
            sob = (sobel(h) - sobel(c)).abs().sum(dim=(1, 2, 3)) / n
            cands.append(pi * l1 + (1 - pi) * sob)
    return torch.stack(cands).min(dim=0).values.mean()


def shift_revssim_loss(hr, mask, pred, border=3, bit_depth=16, eta=0.25):
    """cfg loss = l1msssim (models/loss.py:99-124, 189-212), with the reference's quirks: exponential (not Gaussian) windows
    exp(-x / (2 sigma^2)) over x = linspace(-L/2, L/2, L); C1 in the contrast term; variances (not standard deviations) called sigma;
    one scalar per shift for the WHOLE batch, minimum over the shifts."""
    hr, pred = hr.to(torch.float64), pred.to(torch.float64)
    m = mask.to(torch.float64)
    B, S = pred.shape[0], pred.shape[1]
    L = S - 2 * border
    nb = 2.0 ** bit_depth - 1
    C1, C3 = (0.01 * nb) ** 2, (0.03 * nb) ** 2 / 2
    x = torch.linspace(-L / 2, L / 2, L, dtype=torch.float64)
    cp = pred[:, border:border + L, border:border + L]
    cands = []
    for i in range(2 * border + 1):
        for j in range(2 * border + 1):
            h, mm = hr[:, i:i + L, j:j + L], m[:, i:i + L, j:j + L]
            n = mm.sum(dim=(1, 2, 3))
            b = ((h - cp * mm).sum(dim=(1, 2, 3)) / n).reshape(-1, 1, 1, 1)
            c = (cp + b) * mm
            ws = []
            for sig in (0.5, 1.0, 2.0, 4.0, 8.0):
                w1 = torch.exp(-x / (2 * sig ** 2))
                w = torch.outer(w1, w1).reshape(1, L, L, 1) * mm
                ws.append(w / w.sum(dim=(1, 2, 3), keepdim=True))
            w = torch.stack(ws)                                          # [5,B,L,L,1]
            mu_h = (w * h).sum(dim=(2, 3), keepdim=True)
            mu_s = (w * c).sum(dim=(2, 3), keepdim=True)
            s_h = (w * h ** 2).sum(dim=(2, 3), keepdim=True) - mu_h ** 2
            s_s = (w * c ** 2).sum(dim=(2, 3), keepdim=True) - mu_s ** 2
            cov = (w * h * c).sum(dim=(2, 3), keepdim=True) - mu_s * mu_h
            lum = (2 * mu_h * mu_s + C1) / (mu_h ** 2 + mu_s ** 2 + C1)
            con = (2 * s_h * s_s + C1) / (s_h ** 2 + s_s ** 2 + C1)
            stc = (2 * cov + C3) / (s_h * s_s + C3)
            pcs = (con * stc).prod(dim=0)
            loss = 1 - (lum * pcs).sum() / B
            l1w = ((h - c).abs() * w).sum() / B
            cands.append(eta * loss + (1 - eta) * l1w / nb)
    return torch.stack(cands).min()
