"""Diagnostic (not part of the product): the same 300-step training run twice from the same seed must end in bit-identical parameters
(no race between the side stream, the prefetch thread and the main chain; no run-to-run reduction-order freedom)."""
import hashlib, logging, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probav_amd import synth
from probav_amd.loss import Losses
from probav_amd.modelsTF import WDSRConv3D
from probav_amd.trainClass import ModelTrainer, make_optimizer
logging.disable(logging.CRITICAL)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = 64
x, hr, mask = synth.synth_batch(B * 6, seed=2)

def run():
    model = WDSRConv3D("t", "NIR", synth.NIR_MEAN, synth.NIR_STD, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True)
    model.load_variables(synth.synth_params(seed=1, perturb=True)); model = model.to("cuda:0")
    losses = Losses(targetShape=(48, 48, 1)); opt = make_optimizer("nadam", model, 5e-4)
    with tempfile.TemporaryDirectory() as d:
        tr = ModelTrainer(model=model, loss=losses.shiftCompensatedL1Loss, metric=losses.shiftCompensatedcPSNR, optimizer=opt, ckptDir=d, logDir=d, evalStep=100)
        tr.fitTrainData(x, (hr, mask), B, (steps * B + len(x) - 1) // len(x), (x[:B], hr[:B], mask[:B]), seed=5)
        torch.cuda.synchronize()
        p = model.flat.detach().cpu().numpy()
        return hashlib.sha256(p.tobytes()).hexdigest(), float(abs(p).max()), tr.step

a = run(); b = run()
print("run 1:", a); print("run 2:", b); print("IDENTICAL" if a[0] == b[0] else "DIFFERENT")
