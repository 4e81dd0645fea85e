"""Diagnostic: the same short Nadam run on the native fp32-MFMA kernels (impl 2), the x6 kernels (3) and the H3 kernels (4) -- loss per step.
    python tools/compare_training.py [steps]
The three runs start from the same weights and see the same batches; they differ only in the arithmetic of the hot layers."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
synth = importlib.import_module("probav_amd.synth")
from probav_amd.modelsTF import WDSRConv3D          # noqa: E402
from probav_amd.loss import Losses                  # noqa: E402
from probav_amd.trainClass import HipNadam          # noqa: E402


def run(impl, steps, batch=32):
    dev = "cuda:0"
    m = WDSRConv3D("t", "NIR", synth.NIR_MEAN, synth.NIR_STD, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True, seed=0)
    m.load_variables(synth.synth_params(seed=3, perturb=True))
    m = m.to(dev)
    m.set_impl(impl)
    lo = Losses(targetShape=(48, 48, 1))
    opt = HipNadam([m.flat], lr=5e-4)
    out = []
    for k in range(steps):
        x, hr, mask = (torch.as_tensor(a).to(dev) for a in synth.synth_batch(batch, seed=100 + k % 8))
        m.flat.grad = None
        loss = lo.shiftCompensatedL1Loss(hr, mask, m(x, training=True))
        loss.backward()
        opt.step()
        out.append(float(loss))
    return np.array(out)


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    res = {impl: run(impl, steps) for impl in (2, 3, 4)}
    print("step   impl2        impl3        impl4        |3-2|/2     |4-2|/2")
    for k in range(steps):
        if k < 5 or k % 5 == 4:
            a, b, c = res[2][k], res[3][k], res[4][k]
            print("%4d  %11.4f  %11.4f  %11.4f   %.2e   %.2e" % (k, a, b, c, abs(b - a) / a, abs(c - a) / a))
