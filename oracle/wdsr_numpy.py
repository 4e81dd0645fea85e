"""fp64 numpy restatement of the reference's WDSR-B Conv3D network and shift-compensated losses.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- PARITY UNPINNED: TensorFlow / TFA cannot be
run here, so this follows the reference source line by line instead of its outputs.

Every function cites the reference lines it restates (paths relative to /root/reference).
Tensor layout is the reference's: activations [N, H, W, T, C], kernels [kh, kw, kt, Cin, Cout]
(2-D: [kh, kw, Cin, Cout]); the three Conv3D "spatial" axes are (H, W, T).

Written with explicit tap loops + einsum so that it shares no code path with the torch
restatement (oracle/wdsr_torch.py) it is cross-checked against.
"""
import numpy as np

F64 = np.float64


# ----------------------------------------------------------------------------------------------
# architecture tables
# ----------------------------------------------------------------------------------------------
def reducer_plan(numImgLR):
    """(k, pad_hw, pad_t) of the valid k*k*k `convReducer_i` layers, per temporal depth; every pad is a
    tf.pad(mode='reflect').

    models/modelsTF.py:62-69 selects the reducer by numImgLR:
      9  -> ConvReduceAndUpscale   (:152-164)  numImgLR//scale = 3 reducers, reflect pad H,W only before the first
      13 -> ConvReduceAndUpscalev3 (:123-150)  5 reducers, reflect pad before the first three
      7  -> ConvReduceAndUpscalev2 (:166-175)  2 reducers, no pad
      19 -> ConvReduceAndUpscaleEx (:76-121, EXPERIMENTAL there)  10 reducers; the first is 5x5x5 on a reflect pad of 2 on
            H, W and T, the second pads (2,2,1), the third and fourth (2,2,0), the fifth (1,1,0), the rest none
    """
    a, b = (3, 1, 0), (3, 0, 0)
    if numImgLR == 9:
        return [a, b, b]
    if numImgLR == 13:
        return [a, a, a, b, b]
    if numImgLR == 7:
        return [b, b]
    if numImgLR == 19:
        return [(5, 2, 2), (3, 2, 1), (3, 2, 0), (3, 2, 0), a, b, b, b, b, b]
    raise ValueError("reference defines reducers only for numImgLR in {7, 9, 13, 19}; got %r" % numImgLR)


def layer_specs(numFilters=32, numResBlocks=12, expRate=8, decayRate=0.8, numImgLR=9, scale=3, inChannels=1):
    """[(keras_name, v_shape)] in Keras topological order = checkpoint order
    `model/layer_with_weights-K` (modelInfo/ckpt_p16t9c85r12/NIR/ckpt-124.index).
    Per layer the variables are g [Cout], v (= kernel), bias [Cout]  (TFA WeightNormalization)."""
    f = numFilters
    dec = int(f * decayRate)                                   # models/modelsTF.py:182
    specs = [("mainConv1", (3, 3, 3, inChannels, f))]           # :58; Input(..., 1) or (..., 3): :19-20
    for i in range(numResBlocks):                               # :59-60, :177-189
        specs.append(("expConv_%d" % i, (1, 1, 1, f, f * expRate)))
        specs.append(("decConv_%d" % i, (1, 1, 1, f * expRate, dec)))
        specs.append(("normConv_%d" % i, (3, 3, 3, dec, f)))
    for i, (k, _, _) in enumerate(reducer_plan(numImgLR)):      # :159-160, :80
        specs.append(("convReducer_%d" % (i + 1), (k, k, k, f, f)))
    s2 = scale * scale
    specs.append(("residConv1", (3, 3, inChannels, s2)))        # :45-50 (depth-interleaved with main path)
    specs.append(("upscaleConv1", (3, 3, 3, f, s2)))            # :162-163
    specs.append(("residConv2", (3, 3, s2, s2)))
    specs.append(("residConv3", (3, 3, s2, s2)))
    return specs


# ----------------------------------------------------------------------------------------------
# primitive ops
# ----------------------------------------------------------------------------------------------
def weight_norm(v, g):
    """TFA WeightNormalization kernel:  tf.nn.l2_normalize(v, axis=all-but-last) * g
    = v * rsqrt(max(sum v^2, 1e-12)) * g      (models/modelsTF.py:191-197)."""
    v = np.asarray(v, F64)
    ss = (v * v).reshape(-1, v.shape[-1]).sum(axis=0)
    return v * (1.0 / np.sqrt(np.maximum(ss, 1e-12))) * np.asarray(g, F64)


def conv_valid(x, w):
    """Cross-correlation, stride 1, no padding.  x [N,H,W,T,Cin] (or [N,H,W,Cin]),
    w [kh,kw,kt,Cin,Cout] (or [kh,kw,Cin,Cout]).  Keras Conv3D/Conv2D never flips the kernel."""
    x = np.asarray(x, F64)
    w = np.asarray(w, F64)
    if w.ndim == 4:                                             # 2-D conv: add a unit T axis
        return conv_valid(x[:, :, :, None, :], w[:, :, None, :, :])[:, :, :, 0, :]
    kh, kw, kt = w.shape[:3]
    N, H, W, T, _ = x.shape
    Ho, Wo, To = H - kh + 1, W - kw + 1, T - kt + 1
    y = np.zeros((N, Ho, Wo, To, w.shape[-1]), F64)
    for a in range(kh):
        for b in range(kw):
            for c in range(kt):
                y += np.einsum("nhwti,io->nhwto", x[:, a:a + Ho, b:b + Wo, c:c + To, :], w[a, b, c])
    return y


def pad_zero_same(x, w):
    """Keras padding='same' for stride 1: (k-1)//2 zeros before, k//2 after, on every kernel axis
    (H, W and T for Conv3D)."""
    ks = w.shape[:-2]
    pads = [(0, 0)] + [((k - 1) // 2, k // 2) for k in ks] + [(0, 0)]
    return np.pad(x, pads, mode="constant")


def wn_conv(x, p, padding, relu):
    """WeightNormalization(Conv(outChannels, k, padding, activation)) (models/modelsTF.py:191-197)."""
    w = weight_norm(p["v"], p["g"])
    if padding == "same":
        x = pad_zero_same(x, w)
    y = conv_valid(x, w) + np.asarray(p["bias"], F64)
    return np.maximum(y, 0.0) if relu else y


def reflect_pad_hw(x, n=1, nt=0):
    """tf.pad(x, [[0,0],[n,n],[n,n],[nt,nt],[0,0]], mode='reflect') (models/modelsTF.py:157-158, :78-79)."""
    return np.pad(x, [(0, 0), (n, n), (n, n), (nt, nt), (0, 0)], mode="reflect")


def depth_to_space(x, s):
    """tf.nn.depth_to_space NHWC: out[n, s*h+i, s*w+j, c] = x[n, h, w, (i*s+j)*Co + c]."""
    N, H, W, C = x.shape
    co = C // (s * s)
    y = x.reshape(N, H, W, s, s, co).transpose(0, 1, 3, 2, 4, 5)
    return y.reshape(N, H * s, W * s, co)


# ----------------------------------------------------------------------------------------------
# network
# ----------------------------------------------------------------------------------------------
def wdsr_forward(x, params, mean, std, numResBlocks=12, numImgLR=9, scale=3, taps=None):
    """WDSRConv3D.build graph (models/modelsTF.py:15-43), x [N, P+6, P+6, T, 1] -> [N, 3P, 3P, 1].
    `params[name] = {"v","g","bias"}`.  `taps` (optional dict) collects intermediates."""
    x = np.asarray(x, F64)
    meanLR = x.mean(axis=3)                                     # :23  reduce_mean over T -> [N,H,W,1]
    xn = (x - mean) / std                                       # :26, :199-200
    mn = (meanLR - mean) / std                                  # :27

    # main / high-frequency path (:55-74)
    h = wn_conv(xn, params["mainConv1"], "same", True)          # :58
    if taps is not None:
        taps["mainConv1"] = h
    for i in range(numResBlocks):                               # :177-189
        e = wn_conv(h, params["expConv_%d" % i], "same", True)
        d = wn_conv(e, params["decConv_%d" % i], "same", False)
        n = wn_conv(d, params["normConv_%d" % i], "same", False)
        h = n + h
        if taps is not None:
            taps["dec_%d" % i] = d
            taps["block_%d" % i] = h
    for i, (_, pad, pad_t) in enumerate(reducer_plan(numImgLR)):  # :152-164 / :123-150 / :166-175 / :76-121
        if pad or pad_t:
            h = reflect_pad_hw(h, pad, pad_t)
        h = wn_conv(h, params["convReducer_%d" % (i + 1)], "valid", True)
        if taps is not None:
            taps["reducer_%d" % (i + 1)] = h
    h = wn_conv(h, params["upscaleConv1"], "valid", False)      # :162-163  -> [N,P,P,1,s*s]
    assert h.shape[3] == 1, "temporal axis must collapse to 1 before Reshape (models/modelsTF.py:71)"
    main = depth_to_space(h[:, :, :, 0, :], scale)              # :71-73

    # low-frequency residual path on the T-mean image (:45-53)
    r = wn_conv(mn, params["residConv1"], "valid", True)
    r = wn_conv(r, params["residConv2"], "valid", False)
    r = wn_conv(r, params["residConv3"], "valid", False)
    resid = depth_to_space(r, scale)
    if taps is not None:
        taps["main"] = main
        taps["resid"] = resid
    return (main + resid) * std + mean                          # :38, :41, :202-203


# ----------------------------------------------------------------------------------------------
# shift-compensated losses (models/loss.py)
# ----------------------------------------------------------------------------------------------
def _shift_terms(hr, mask, pred, cropBorder=3):
    """Common scaffolding of stackL1Loss / stackL2Loss / stackcPSNR (models/loss.py:140-180):
    returns l1[49,B], mse[49,B] in the reference's (i outer, j inner) order."""
    hr = np.asarray(hr, F64)[..., 0]
    m = np.asarray(mask).astype(F64)[..., 0]                    # cropImage casts to f32 (utils/utils.py:44)
    p = np.asarray(pred, F64)[..., 0]
    S = hr.shape[1]
    c = cropBorder
    L = S - 2 * c                                               # loss.py:24-25
    P = p[:, c:c + L, c:c + L]                                  # loss.py:74-75
    l1, l2 = [], []
    for i in range(2 * c + 1):                                  # loss.py:79-81
        for j in range(2 * c + 1):
            H = hr[:, i:i + L, j:j + L]                         # :141
            M = m[:, i:i + L, j:j + L]                          # :142
            n = M.sum(axis=(1, 2))                              # :144
            b = (1.0 / n) * (H - P * M).sum(axis=(1, 2))        # :146, :182-187 (HR NOT masked)
            C = (P + b[:, None, None]) * M                      # :148-149
            l1.append((1.0 / n) * np.abs(H - C).sum(axis=(1, 2)))      # :226-228
            l2.append((1.0 / n) * np.square(H - C).sum(axis=(1, 2)))   # :230-232
    return np.stack(l1), np.stack(l2)


def shift_l1_loss(hr, mask, pred, cropBorder=3):
    """Losses.shiftCompensatedL1Loss (models/loss.py:73-84): mean_B min_shift L1."""
    l1, _ = _shift_terms(hr, mask, pred, cropBorder)
    return l1.min(axis=0).mean()


def shift_l2_loss(hr, mask, pred, cropBorder=3):
    """Losses.shiftCompensatedL2Loss (models/loss.py:55-71)."""
    _, l2 = _shift_terms(hr, mask, pred, cropBorder)
    return l2.min(axis=0).mean()


def shift_cpsnr(hr, mask, pred, cropBorder=3, bitDepth=16):
    """Losses.shiftCompensatedcPSNR (models/loss.py:37-53, 234-238): per-sample max over shifts."""
    _, l2 = _shift_terms(hr, mask, pred, cropBorder)
    nb = 2.0 ** bitDepth - 1.0
    return (10.0 * (np.log(nb * nb / l2) / np.log(10.0))).max(axis=0)


def shift_l1_grad(hr, mask, pred, cropBorder=3):
    """d(shiftCompensatedL1Loss)/d(pred), derived in SURVEY.md A.4: the gradient of the arg-min
    shift only (ties split equally, as tf.reduce_min does), through both C and the bias b."""
    hr64 = np.asarray(hr, F64)[..., 0]
    m = np.asarray(mask).astype(F64)[..., 0]
    p = np.asarray(pred, F64)[..., 0]
    B, S = hr64.shape[0], hr64.shape[1]
    c = cropBorder
    L = S - 2 * c
    l1, _ = _shift_terms(hr, mask, pred, cropBorder)
    lmin = l1.min(axis=0)
    g = np.zeros_like(p)
    P = p[:, c:c + L, c:c + L]
    for bi in range(B):
        ties = np.flatnonzero(l1[:, bi] == lmin[bi])
        for s in ties:
            i, j = divmod(int(s), 2 * c + 1)
            H = hr64[bi, i:i + L, j:j + L]
            M = m[bi, i:i + L, j:j + L]
            n = M.sum()
            b = (H - P[bi] * M).sum() / n
            sg = np.sign(H - (P[bi] + b) * M)
            gk = -(M / n) * (sg - (sg * M).sum() / n)
            g[bi, c:c + L, c:c + L] += gk / (len(ties) * B)
    return g[..., None]
