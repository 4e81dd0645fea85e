"""Generates tests/golden/*.npz from the fp64 oracle (oracle/wdsr_numpy + oracle/wdsr_torch).

The reference (TensorFlow + tensorflow-addons) is not installable here and ships no golden vectors
(SURVEY.md F1, F4), so these fixtures are produced by the in-repo restatement -- "parity unpinned".
They pin the restatement itself: any later edit of the oracle or of the HIP kernels must reproduce them.

    python tests/golden/make_golden.py        # rewrites the .npz files next to this script
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from probav_amd import synth                     # noqa: E402
from oracle import wdsr_numpy as on              # noqa: E402
from oracle import wdsr_torch as ot              # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
FULL_GRADS = ["mainConv1", "expConv_0", "decConv_11", "normConv_5", "convReducer_1", "residConv1", "upscaleConv1", "residConv3"]
EX_GRADS = {19: ["convReducer_2", "convReducer_5", "convReducer_10"]}   # convReducer_1 is 5x5x5 there (128 000 weights: norms only)


def make(T, seed_w, seed_x, batch=2):
    params = synth.synth_params(seed=seed_w, perturb=True, numImgLR=T)
    x, hr, mask = synth.synth_batch(batch, seed=seed_x, numImgLR=T)
    flat = synth.flatten_params(params, numImgLR=T)
    taps = {}
    y_np = on.wdsr_forward(x, params, synth.NIR_MEAN, synth.NIR_STD, numImgLR=T, taps=taps)
    pt = ot.to_torch_params(params)
    pred, loss, grads = ot.train_step_grads(torch.tensor(x, dtype=torch.float64), torch.tensor(hr), torch.tensor(mask),
                                            pt, synth.NIR_MEAN, synth.NIR_STD, numImgLR=T)
    assert np.abs(y_np - pred.numpy()).max() < 1e-7 * np.abs(y_np).max(), "numpy and torch restatements disagree"
    out = {
        "x": x, "hr": hr, "mask": mask,
        "param_checksum": np.array([flat.astype(np.float64).sum(), (flat.astype(np.float64) ** 2).sum()]),
        "pred": pred.numpy(), "loss_l1": np.array(float(loss)),
        "loss_l2": np.array(on.shift_l2_loss(hr, mask, y_np)),
        "cpsnr": on.shift_cpsnr(hr, mask, y_np),
        "dpred": on.shift_l1_grad(hr, mask, y_np),
        "block_5_mean_abs": np.array(np.abs(taps["block_5"]).mean()),
        "main": taps["main"], "resid": taps["resid"],
    }
    norms = []
    for name, g in grads.items():
        for key in ("g", "v", "bias"):
            a = g[key].numpy()
            norms.append([np.sqrt((a ** 2).sum()), np.abs(a).max()])
            if name in FULL_GRADS + EX_GRADS.get(T, []) and a.size <= 30000:
                out["grad/%s/%s" % (name, key)] = a
    out["grad_norms"] = np.array(norms)
    np.savez_compressed(os.path.join(HERE, "wdsr_t%d_b%d.npz" % (T, batch)), **out)
    if T == 9:
        # ALL 132 gradient tensors of the headline network, element-wise (a norm cannot see a wrong direction): the fp64
        # gradient rounded to fp32, flat in the engine's parameter order ([g | v | bias] per layer, probav_amd.arch.layer_table)
        from probav_amd.arch import layer_table
        layers, total = layer_table(numImgLR=T)
        gflat = np.zeros(total, np.float64)
        for L in layers:
            g = grads[L.name]
            gflat[L.g_off:L.v_off] = g["g"].numpy()
            gflat[L.v_off:L.b_off] = g["v"].numpy().reshape(-1)
            gflat[L.b_off:L.b_off + L.cout] = g["bias"].numpy()
        np.savez_compressed(os.path.join(HERE, "wdsr_t%d_b%d_grads.npz" % (T, batch)), grad_flat=gflat.astype(np.float32))
    print("T=%d: loss %.6f  cpsnr %s  |pred| max %.1f" % (T, float(loss), out["cpsnr"], np.abs(out["pred"]).max()))


if __name__ == "__main__":
    make(9, 101, 102)
    make(13, 131, 132)
    make(7, 71, 72)
    make(19, 191, 192)
