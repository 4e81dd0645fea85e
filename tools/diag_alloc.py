import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
ge.build()
from probav_amd import synth
from probav_amd.modelsTF import WDSRConv3D
from probav_amd.loss import Losses
dev = torch.device("cuda:0")
m = WDSRConv3D("b", "NIR", synth.NIR_MEAN, synth.NIR_STD, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True)
m.load_variables(synth.synth_params(seed=1234)); m = m.to(dev)
x, hr, mask = (torch.as_tensor(a).to(dev) for a in synth.synth_batch(128, seed=1))
lo = Losses(targetShape=(48, 48, 1))
def sync(): torch.cuda.synchronize()
def t(f, n=6, name=""):
    ts = []
    for _ in range(n):
        sync(); t0 = time.perf_counter(); r = f(); sync(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%-40s %s" % (name, " ".join("%.1f" % v for v in ts)), flush=True)
    return r
nb = 3_400_000_000 // 4
t(lambda: torch.empty(nb, device=dev), name="torch.empty 3.4GB (dropped)")
keep = []
t(lambda: keep.append(torch.empty(nb, device=dev)), name="torch.empty 3.4GB (kept)")
keep.clear()
t(lambda: keep.append(torch.empty(nb, device=dev)), name="torch.empty 3.4GB (kept, after clear)")
keep.clear()
eng = int(m._handle().value)
def fwd_only():
    y, ws = torch.ops.probav.wdsr_forward(m.flat.detach(), x, eng, 48, True)
    return None
t(fwd_only, name="op forward (no grad, outputs dropped)")
def fwd_grad():
    y, ws = torch.ops.probav.wdsr_forward(m.flat, x, eng, 48, True)
    return None
t(fwd_grad, name="op forward (grad, outputs dropped)")
def full():
    pred = m(x, training=True)
    loss = lo.shiftCompensatedL1Loss(hr, mask, pred)
    m.flat.grad = None
    loss.backward()
    return loss
t(full, n=8, name="full step, result dropped")
h = []
def full_keep():
    h[:] = [full()]
t(full_keep, n=8, name="full step, loss kept until next")
import gc
print("allocated GiB after the loops: %.2f (reserved %.2f)" % (torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30))
print(torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_stats()["num_device_alloc"], torch.cuda.memory_stats()["num_device_free"])
print(torch.cuda.memory_summary()[:1500])

# --- with the engine's per-launch HIP events on (bench.py's roofline leg)
import ctypes
from probav_amd import _lib
L = _lib.lib(); hnd = m._handle()
_lib.check(L.probav_engine_profile_classes(hnd, 1 << 12))
_lib.check(L.probav_engine_profile(hnd, 1, 64 * 40))
def timed_parts():
    t0 = time.perf_counter(); pred = m(x, training=True); t1 = time.perf_counter()
    loss = lo.shiftCompensatedL1Loss(hr, mask, pred); t2 = time.perf_counter()
    m.flat.grad = None; loss.backward(); t3 = time.perf_counter()
    sync(); t4 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3
for k in range(6):
    print("events on: host ms fwd %.2f loss %.2f bwd %.2f sync %.2f" % timed_parts(), flush=True)
_lib.check(L.probav_engine_profile(hnd, 0, 0))
for k in range(3):
    print("events off: host ms fwd %.2f loss %.2f bwd %.2f sync %.2f" % timed_parts(), flush=True)
