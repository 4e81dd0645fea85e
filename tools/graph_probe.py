"""Diagnostic (not part of the product): does replaying the training step as a captured HIP graph shorten it?
   python tools/graph_probe.py    (on the GPU box, from the repo root)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probav_amd import synth
from probav_amd.loss import Losses
from probav_amd.modelsTF import WDSRConv3D

dev = torch.device("cuda:0")
B, T = 128, 9
losses = Losses(targetShape=(48, 48, 1))
model = WDSRConv3D("bench", "NIR", synth.NIR_MEAN, synth.NIR_STD, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, T, 16, True)
model.load_variables(synth.synth_params(seed=1234, numImgLR=T))
model = model.to(dev)
x, hr, mask = (torch.as_tensor(a).to(dev) for a in synth.synth_batch(B, seed=1234, numImgLR=T))

def step():
    pred = model(x, training=True)
    loss = losses.shiftCompensatedL1Loss(hr, mask, pred)
    model.flat.grad = None
    loss.backward()
    return loss

def timeit(f, k=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3

print("eager   %.3f ms/step" % timeit(step))
# capture (side stream created by the eager passes above)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    model.flat.grad = None
    with torch.cuda.graph(g):
        loss = step()
    g0 = model.flat.grad.clone()
    print("graph   %.3f ms/step" % timeit(g.replay))
    l_e = float(step()); ge = model.flat.grad.clone()
    g.replay(); torch.cuda.synchronize()
    print("loss eager %.6f graph %.6f; max |grad diff| %.3e" % (l_e, float(loss), float((model.flat.grad - ge).abs().max())))
except Exception as e:
    print("capture failed:", repr(e)[:500])
