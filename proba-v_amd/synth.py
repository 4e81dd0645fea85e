"""Seeded synthetic weights and patches of the shapes the hot path sees.

There is no dataset and no trained checkpoint available, so bench.py, the smoke test
and the parity tests run on these: LR/HR ~ clip(N(mu, sigma), 0, 16383) with the reference's NIR
statistics (train.py:47-52), masks Bernoulli(0.9) with >= 85 % clear pixels
(cfg/p16t9c85r12.cfg `high_res_threshold`), weights in the state the reference has after its first
call: Glorot-uniform `v`, g = ||v||, bias = 0 (TFA WeightNormalization, data_init=False).
Everything is generated with numpy's PCG64 on the host so every device sees identical bits.
"""
import numpy as np

from .arch import layer_table

NIR_MEAN, NIR_STD = 8075.2045, 3160.7272      # train.py:48-49
RED_MEAN, RED_STD = 5266.2245, 3431.8614      # train.py:51-52


def synth_params(seed=1234, perturb=False, **arch):
    """{name: {"g","v","bias"}} float32.  `perturb=True` moves g and bias off their init values so
    that tests exercise the bias and scale paths (bias = 0, g = ||v|| hides bugs)."""
    rng = np.random.default_rng(seed)
    layers, _ = layer_table(**arch)
    out = {}
    for L in layers:
        vs = L.vshape
        recept = int(np.prod(vs[:-2]))
        limit = np.sqrt(6.0 / (recept * vs[-2] + recept * vs[-1]))     # Keras glorot_uniform
        v = rng.uniform(-limit, limit, size=vs).astype(np.float32)
        g = np.sqrt((v.astype(np.float64) ** 2).reshape(-1, vs[-1]).sum(0)).astype(np.float32)
        b = np.zeros(vs[-1], np.float32)
        if perturb:
            g = (g * rng.uniform(0.8, 1.2, size=g.shape)).astype(np.float32)
            b = rng.normal(0.0, 0.05, size=b.shape).astype(np.float32)
        out[L.name] = {"g": g, "v": v, "bias": b}
    return out


def flatten_params(params, **arch):
    layers, total = layer_table(**arch)
    flat = np.zeros(total, np.float32)
    for L in layers:
        p = params[L.name]
        flat[L.g_off:L.v_off] = p["g"]
        flat[L.v_off:L.b_off] = p["v"].reshape(-1)
        flat[L.b_off:L.b_off + L.cout] = p["bias"]
    return flat


def unflatten_params(flat, **arch):
    layers, _ = layer_table(**arch)
    return {L.name: {"g": np.array(flat[L.g_off:L.v_off]),
                     "v": np.array(flat[L.v_off:L.b_off]).reshape(L.vshape),
                     "bias": np.array(flat[L.b_off:L.b_off + L.cout])} for L in layers}


def synth_batch(batch, seed=1234, numImgLR=9, patchSizeLR=16, maxShift=6, scale=3,
                mean=NIR_MEAN, std=NIR_STD, inChannels=1):
    """(x [B,P+s,P+s,T,C] f32, hr [B,3P,3P,1] f32, mask [B,3P,3P,1] bool)."""
    rng = np.random.default_rng(seed)
    hin, hout = patchSizeLR + maxShift, scale * patchSizeLR
    x = np.clip(rng.normal(mean, std, size=(batch, hin, hin, numImgLR, inChannels)), 0, 16383).astype(np.float32)
    hr = np.clip(rng.normal(mean, std, size=(batch, hout, hout, 1)), 0, 16383).astype(np.float32)
    mask = rng.random(size=(batch, hout, hout, 1)) < 0.9
    for b in range(batch):                                   # keep >= 85 % clear pixels per sample
        while mask[b].mean() < 0.85:
            mask[b] |= rng.random(size=mask[b].shape) < 0.5
    return x, hr, mask
