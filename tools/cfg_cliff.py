"""Diagnostic (not part of the product): what a cfg value other than the shipped one costs.  The one-wave-per-SIMD kernels of round 5 have
instances for the shipped configuration only (patch 16 -> 22 x 22 rows, 9 or 7 frames in the residual blocks, decay 0.8 -> 25 channels, 32 filters); every other value
runs the general kernels of rounds 2 - 4 -- correct (tests/test_gpu_parity.py::test_other_patch_sizes_agree_across_engines), slower per voxel.  This prints, per
configuration, ms per training step (forward + shift-L1 + backward), patches/s, and the throughput per VOXEL-MAC relative to the shipped configuration.
    python tools/cfg_cliff.py            (on the GPU box, from the repo root)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from probav_amd import synth
from probav_amd.loss import Losses
from probav_amd.modelsTF import WDSRConv3D

dev = torch.device("cuda:0")


def macs_per_patch(P, T, F=32, E=8, decay=0.8, R=12):
    """forward MACs of one patch (SURVEY.md section 8a's table, for any P / T / decay): residual blocks only -- the part the round-5 kernels serve."""
    H = P + 6
    D = int(F * decay)
    V = H * H * T
    return R * V * (F * F * E + F * E * D + 27 * D * F)


def run(P, T, B, decay=0.8, F=32, steps=30):
    model = WDSRConv3D("cliff", "NIR", synth.NIR_MEAN, synth.NIR_STD, 6).build(3, F, (3, 3, 3), 12, 8, decay, T, P, True, seed=0)
    model = model.to(dev)
    rng = np.random.default_rng(1)
    H = P + 6
    x = torch.as_tensor(np.clip(rng.normal(synth.NIR_MEAN, synth.NIR_STD, (B, H, H, T, 1)), 0, 16383).astype(np.float32)).to(dev)
    hr = torch.as_tensor(np.clip(rng.normal(synth.NIR_MEAN, synth.NIR_STD, (B, 3 * P, 3 * P, 1)), 0, 16383).astype(np.float32)).to(dev)
    mask = torch.as_tensor(rng.random((B, 3 * P, 3 * P, 1)) < 0.9).to(dev)
    losses = Losses(targetShape=(3 * P, 3 * P, 1))

    def step():
        pred = model(x, training=True)
        loss = losses.shiftCompensatedL1Loss(hr, mask, pred)
        model.flat.grad = None
        loss.backward()
    for _ in range(5):
        step()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    torch.cuda.synchronize()
    evs[0].record()
    for i in range(steps):
        step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
    med = ms[len(ms) // 2]
    model._ws.clear()
    del model
    torch.cuda.empty_cache()
    return med


if __name__ == "__main__":
    rows = []
    base = None
    for name, P, T, B, decay in (("shipped: patch 16, 9 frames, decay 0.8", 16, 9, 128, 0.8),
                                 ("7 frames", 16, 7, 128, 0.8),
                                 ("13 frames (reducer v3)", 16, 13, 128, 0.8),
                                 ("19 frames (experimental reducer)", 16, 19, 64, 0.8),
                                 ("patch 32 (38 x 38 rows)", 32, 9, 40, 0.8),
                                 ("patch 24 (30 x 30 rows)", 24, 9, 64, 0.8),
                                 ("decay 0.5 (16 channels)", 16, 9, 128, 0.5),
                                 ("decay 0.9 (28 channels)", 16, 9, 128, 0.9)):
        try:
            med = run(P, T, B, decay)
        except Exception as ex:                                   # a configuration the engine refuses is a row of the table too
            rows.append({"config": name, "error": str(ex)[:200]})
            print(json.dumps(rows[-1]), flush=True)
            continue
        gmacs = macs_per_patch(P, T, decay=decay) * B / (med * 1e-3) / 1e9
        if base is None:
            base = gmacs
        rows.append({"config": name, "batch": B, "ms_per_step_median": round(med, 3), "patches_per_s": round(B / (med * 1e-3), 1),
                     "block_GMAC_per_s": round(gmacs, 1), "per_mac_throughput_vs_shipped": round(gmacs / base, 3)})
        print(json.dumps(rows[-1]), flush=True)
