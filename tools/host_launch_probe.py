import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probav_amd import synth, _lib
from probav_amd.modelsTF import WDSRConv3D
B = 128
model = WDSRConv3D("t", "NIR", synth.NIR_MEAN, synth.NIR_STD, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True)
model.load_variables(synth.synth_params(seed=1)); model = model.to("cuda:0")
x = torch.as_tensor(synth.synth_batch(B, seed=2)[0]).to("cuda:0")
L = _lib.lib(); h = model._handle()
nb = L.probav_workspace_bytes(h, B, 1)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda:0")
y = torch.empty(B, 48, 48, 1, device="cuda:0"); dy = torch.randn_like(y); g = torch.empty_like(model.flat)
s = _lib.current_stream()
for _ in range(3):
    _lib.check(L.probav_forward(h, _lib.ptr(model.flat), _lib.ptr(x), _lib.ptr(y), _lib.ptr(ws), nb, B, 1, s)); _lib.check(L.probav_backward(h, _lib.ptr(model.flat), _lib.ptr(dy), _lib.ptr(g), _lib.ptr(ws), nb, B, s))
torch.cuda.synchronize()
tf = tb = 0.0
for _ in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    L.probav_forward(h, _lib.ptr(model.flat), _lib.ptr(x), _lib.ptr(y), _lib.ptr(ws), nb, B, 1, s); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    L.probav_backward(h, _lib.ptr(model.flat), _lib.ptr(dy), _lib.ptr(g), _lib.ptr(ws), nb, B, s); t3 = time.perf_counter()
    tf += t1 - t0; tb += t3 - t2
print("host time of the C calls on an idle GPU: forward %.3f ms, backward %.3f ms" % (tf / 20 * 1e3, tb / 20 * 1e3))
