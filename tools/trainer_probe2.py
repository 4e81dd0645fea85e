import logging, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from probav_amd import synth
from probav_amd.loss import Losses
from probav_amd.modelsTF import WDSRConv3D
from probav_amd.trainClass import ModelTrainer, make_optimizer, BatchPrefetcher, shuffle_repeat_batch, _LateScalars
B = 128
logging.disable(logging.CRITICAL)
model = WDSRConv3D("t", "NIR", synth.NIR_MEAN, synth.NIR_STD, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True)
model.load_variables(synth.synth_params(seed=1)); model = model.to("cuda:0")
n = B * 8
x, hr, mask = synth.synth_batch(n, seed=2)
print("dtypes", x.dtype, hr.dtype, mask.dtype, x.shape)
losses = Losses(targetShape=(48, 48, 1))
opt = make_optimizer("nadam", model, 5e-4)
dev = torch.device("cuda:0")
# 1. the prefetcher alone
rng = np.random.default_rng(0)
mask_dtype = torch.as_tensor(np.asarray(mask[:1])).dtype
t0 = time.perf_counter(); k = 0
for xb, hb, mb in BatchPrefetcher((x, hr, mask), (torch.float32, torch.float32, mask_dtype), shuffle_repeat_batch(n, 25, B, 256, rng), dev):
    k += 1
torch.cuda.synchronize(); print("prefetcher alone: %.3f ms/batch over %d" % ((time.perf_counter() - t0) / k * 1e3, k))
# 2. trainStep alone on resident tensors
with tempfile.TemporaryDirectory() as d:
    tr = ModelTrainer(model=model, loss=losses.shiftCompensatedL1Loss, metric=losses.shiftCompensatedcPSNR, optimizer=opt, ckptDir=d, logDir=d, evalStep=10 ** 9)
    for _ in range(5): tr.trainStep(xb, hb, mb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): tr.trainStep(xb, hb, mb)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("trainStep alone: host enqueue %.3f ms/step, with sync %.3f ms/step" % ((t1 - t0) / 100 * 1e3, (t2 - t0) / 100 * 1e3))
    late = _LateScalars(lambda v, m: None, dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(100):
        tr.trainStep(xb, hb, mb); late.push((tr.trainLoss, tr.trainPSNR), (0, i, i))
    torch.cuda.synchronize(); print("trainStep + late scalars: %.3f ms/step" % ((time.perf_counter() - t0) / 100 * 1e3))
