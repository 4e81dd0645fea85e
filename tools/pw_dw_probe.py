"""Diagnostic (not part of the product): per-ROW error of dW1 and per-COLUMN error of dW2 of the fused pointwise backward under channel gains of 2^-24 .. 1 on the block's
input channels / on d dec's channels (every slice against its own maximum), impl 4 (H3) | 3 (x6) | 2 (fp32 MFMA)."""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_h3_range as T
L = T._L(); dev = torch.device('cuda:0')
D, vps, ns = 25, 22 * 22 * 9, 4
nvox = vps * ns
rng = np.random.default_rng(3)
x = rng.normal(size=(ns, vps, 32)).astype(np.float32)
w1 = (rng.normal(size=(32, 256)) / np.sqrt(32)).astype(np.float32); b1 = rng.normal(scale=0.3, size=256).astype(np.float32)
w2 = (rng.normal(size=(256, D)) / 16).astype(np.float32); b2 = np.zeros(D, np.float32)
ddec = rng.normal(size=(ns, vps, D)).astype(np.float32); dskip = np.zeros((ns, vps, 32), np.float32)
gx, gd = T._gains(rng, 32, -24), T._gains(rng, D, -24)
which = sys.argv[1] if len(sys.argv) > 1 else 'both'
if which in ('x', 'both'): x *= gx
if which in ('d', 'both'): ddec *= gd
print('gains on', which)
X, W1, W2 = x.reshape(nvox, 32).astype(np.float64), w1.astype(np.float64), w2.astype(np.float64)
Hpre = X @ W1 + b1; Hh = np.maximum(Hpre, 0)
dH = (ddec.reshape(nvox, D).astype(np.float64) @ W2.T) * (Hpre > 0)
r1, r2 = X.T @ dH, Hh.T @ ddec.reshape(nvox, D)
for impl in (4, 3, 2):
    xd, w1d, b1d, w2d, ddd, dsd = (T._t(a, dev) for a in (x, w1, b1, w2, ddec, dskip))
    nbytes = L.lib().probav_pw_backward_scratch_bytes(D); scratch = torch.empty(nbytes // 4 + 1, device=dev)
    dx, dw1, db1 = torch.empty((nvox, 32), device=dev), torch.empty((32, 256), device=dev), torch.empty((256,), device=dev)
    dw2, db2 = torch.empty((256, D), device=dev), torch.empty((D,), device=dev)
    L.check(L.lib().probav_pw_backward(L.ptr(xd), L.ptr(ddd), L.ptr(dsd), L.ptr(w1d), L.ptr(b1d), L.ptr(w2d), L.ptr(dx), L.ptr(dw1), L.ptr(db1), L.ptr(dw2), L.ptr(db2),
                                       L.ptr(scratch), nbytes, nvox, vps, D, impl, L.current_stream()))
    e1 = np.abs(dw1.cpu().double().numpy() - r1).max(axis=1) / np.abs(r1).max(axis=1)
    e2 = np.abs(dw2.cpu().double().numpy() - r2).max(axis=0) / np.abs(r2).max(axis=0)
    o1, o2 = np.argsort(gx), np.argsort(gd)
    if impl == 4: keep, keep2 = dw1.cpu().double().numpy().copy(), dw2.cpu().double().numpy().copy()
    if impl == 3:                                     # (against the oracle a single ReLU gate that the device decides the other way is 1e-3 of a small row: a row is a random-walk sum of 17 424 terms --
        d1 = np.abs(keep - dw1.cpu().double().numpy()).max(axis=1) / np.abs(r1).max(axis=1)      #  the two split families flip the same gates, so H3 against x6 shows what the SCALES cost)
        d2 = np.abs(keep2 - dw2.cpu().double().numpy()).max(axis=0) / np.abs(r2).max(axis=0)
        print('   H3 - x6 dW1 rows: worst %.2e | by gain ' % d1.max() + ' '.join('2^%.0f:%.1e' % (np.log2(gx[c]), d1[c]) for c in o1[::4]))
        print('   H3 - x6 dW2 cols: worst %.2e | by gain ' % d2.max() + ' '.join('2^%.0f:%.1e' % (np.log2(gd[c]), d2[c]) for c in o2[::3]))
    print('impl %d  dW1 rows: worst %.2e | by gain ' % (impl, e1.max()) + ' '.join('2^%.0f:%.1e' % (np.log2(gx[c]), e1[c]) for c in o1[::4]))
    print('        dW2 cols: worst %.2e | by gain ' % e2.max() + ' '.join('2^%.0f:%.1e' % (np.log2(gd[c]), e2[c]) for c in o2[::3]))
