#!/usr/bin/env python3
"""BASELINE.json config 4: full 128x128 LR -> 384x384 HR inference (test.py path), 9 frames, 12 blocks, 32 image sets.

    python tools/bench_infer.py [--sets 32] [--steps 10] [--micro-batch 2048]

Times, with the LR frames resident in HBM: unfold of the reflect-padded 134^2 frames into 64 patches of 22x22x9, WDSR-B
forward, clip to [0, 2**16] + round-half-even, 8x8 stitch to 384^2.  Prints one JSON line (images/s, patches/s); also
times the reference's own micro-batching (64 patches per image in batches of 16, test.py:125-134)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--micro-batch", type=int, default=2048)
    args = ap.parse_args()
    import numpy as np
    import torch
    import __graft_entry__ as ge
    ge.build()
    from probav_amd import synth, testClass
    from probav_amd.modelsTF import WDSRConv3D
    dev = torch.device("cuda:0")
    model = WDSRConv3D("infer", "NIR", synth.NIR_MEAN, synth.NIR_STD, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True)
    model.load_variables(synth.synth_params(seed=1234))
    model = model.to(dev)
    rng = np.random.default_rng(7)
    frames = torch.as_tensor(np.clip(rng.normal(synth.NIR_MEAN, synth.NIR_STD, (args.sets, 9, 128, 128)), 0, 16383).astype(np.float32)).to(dev)

    def run(mb):
        patches = testClass.unfold_frames(frames)
        return testClass.resolve_images(model, patches, micro_batch=mb)

    out = {}
    for name, mb in (("batched", args.micro_batch), ("reference_micro_batch_16", 16)):
        for _ in range(2):
            img = run(mb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            img = run(mb)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        out[name] = {"micro_batch": mb, "ms_per_%d_images" % args.sets: round(dt * 1e3, 3), "images_per_s": round(args.sets / dt, 2),
                     "patches_per_s": round(args.sets * 64 / dt, 1)}
    assert img.shape == (args.sets, 384, 384) and float(img.min()) >= 0 and float(img.max()) <= 65536
    print(json.dumps({"metric": "384x384 HR images/s, fwd only (WDSR-B r12 t9, 64 patches of 22x22x9 per image)", "n_gpus": 1,
                      "dtype": "f32", "data": "synthetic", "sets": args.sets, **out}))


if __name__ == "__main__":
    main()
