#!/usr/bin/env python3
"""Diagnostic (not part of the product): what a plain 71-MB device copy costs INSIDE a training step (right behind the backward pass, and
between forward and loss) against the same copy alone.  Run under rocprofv3 --kernel-trace --stats and read the copy kernel's durations:

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/copyprobe -o p -- python3 tools/instep_copy_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probav_amd import synth                                   # noqa: E402
from probav_amd.loss import Losses                              # noqa: E402
from probav_amd.modelsTF import WDSRConv3D                      # noqa: E402

dev = torch.device("cuda", 0)
B = 128
model = WDSRConv3D("probe", "NIR", synth.NIR_MEAN, synth.NIR_STD, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True)
model.load_variables(synth.synth_params(seed=1234))
model = model.to(dev)
x, hr, mask = (torch.as_tensor(a).to(dev) for a in synth.synth_batch(B, seed=1234))
losses = Losses(targetShape=(48, 48, 1))
n = 71 * (1 << 20) // 4
a = torch.zeros(n, device=dev)
b = torch.empty(n, device=dev)
c = torch.empty(n, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "instep"
for it in range(8):
    if mode == "alone":
        for _ in range(4):
            b.copy_(a)
            torch.cuda.synchronize()
        continue
    pred = model(x, training=True)
    b.copy_(a)                                                  # between forward and loss
    loss = losses.shiftCompensatedL1Loss(hr, mask, pred)
    model.flat.grad = None
    loss.backward()
    c.copy_(a)                                                  # right behind the backward pass
    c.add_(1.0)                                                 # (an elementwise kernel of the same size: reads 71 MB, writes 71 MB)
torch.cuda.synchronize()
print("done", mode)
