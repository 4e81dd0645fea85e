"""Diagnostic (not part of the product): wall time per step of ModelTrainer.fitTrainData (host pipeline + prefetch + train step + logging)
on synthetic numpy data, against the bare step bench.py times.   python tools/trainer_probe.py [steps]"""
import logging, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from probav_amd import synth
from probav_amd.loss import Losses
from probav_amd.modelsTF import WDSRConv3D
from probav_amd.trainClass import ModelTrainer, make_optimizer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = 128
logging.disable(logging.CRITICAL)
model = WDSRConv3D("t", "NIR", synth.NIR_MEAN, synth.NIR_STD, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True)
model.load_variables(synth.synth_params(seed=1))
model = model.to("cuda:0")
n = B * 8
x, hr, mask = synth.synth_batch(n, seed=2)
losses = Losses(targetShape=(48, 48, 1))
opt = make_optimizer("nadam", model, 5e-4)
with tempfile.TemporaryDirectory() as d:
    tr = ModelTrainer(model=model, loss=losses.shiftCompensatedL1Loss, metric=losses.shiftCompensatedcPSNR, optimizer=opt, ckptDir=d, logDir=d,
                      evalStep=10 ** 9)
    epochs = (steps * B + n - 1) // n
    tr.fitTrainData(x[:B * 2], (hr[:B * 2], mask[:B * 2]), B, 5, (x[:B], hr[:B], mask[:B]))       # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter(); s0 = tr.step
    tr.fitTrainData(x, (hr, mask), B, epochs, (x[:B], hr[:B], mask[:B]))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    k = tr.step - s0
    print("trainer: %d steps, %.3f ms/step, %.0f patches/s (forward + loss + backward + Nadam + cPSNR + host pipeline)" % (k, dt / k * 1e3, k * B / dt))
