import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import test_gpu_h3_range as T
dev = torch.device('cuda:0')
for which in ('x', 'dy', 'both'):
    rng = np.random.default_rng(7)
    N, hwt, Cin, Cout = 2, (22, 22, 9), 25, 32
    x = rng.normal(size=(N,) + hwt + (Cin,)).astype(np.float32)
    dy = rng.normal(size=(N,) + hwt + (Cout,)).astype(np.float32)
    gx = T._gains(rng, Cin, -24); gd = T._gains(rng, Cout, -24)
    if which in ('x', 'both'): x *= gx
    if which in ('dy', 'both'): dy *= gd
    g = T._geom(N, 22, 22, 9, Cin, 22, 22, 9, Cout, (3, 3, 3), (1, 1, 1))
    ref = T._oracle_wgrad(x, dy, 1)
    got4, got3, got2 = T._wgrad(dev, 4, g, x, dy), T._wgrad(dev, 3, g, x, dy), T._wgrad(dev, 2, g, x, dy)
    print(which, 'whole-tensor', [float(np.abs(a - ref).max() / np.abs(ref).max()) for a in (got4, got3, got2)])
    # slices per (ci, co)
    den = np.abs(ref).max(axis=(0, 1, 2))
    for name, got in (('H3', got4), ('x6', got3), ('f32', got2)):
        e = np.abs(got - ref).max(axis=(0, 1, 2)) / den
        print('  %s per (ci,co) slice: max %.2e  median %.2e   per-ci max %.2e  per-co max %.2e' % (name, e.max(), np.median(e),
              (np.abs(got - ref).max(axis=(0, 1, 2, 4)) / np.abs(ref).max(axis=(0, 1, 2, 4))).max(), (np.abs(got - ref).max(axis=(0, 1, 2, 3)) / np.abs(ref).max(axis=(0, 1, 2, 3))).max()))
    if which == 'x':
        per4 = np.abs(got4 - ref).max(axis=(0, 1, 2, 4)) / np.abs(ref).max(axis=(0, 1, 2, 4))
        for c in np.argsort(gx): print('   gain 2^%6.2f  H3 %.2e' % (np.log2(gx[c]), per4[c]))
