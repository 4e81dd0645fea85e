"""Launch single operators of libprobav_hip.so at the bench workload's size, for `rocprofv3 --kernel-trace --stats`.

    rocprofv3 --kernel-trace --stats -d gpurun_out/ops -- python3 tools/bench_ops.py pw_fwd
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from probav_amd import _lib as L  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "pw_fwd"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda:0")
    nvox, D = 128 * 22 * 22 * 9, 25
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(nvox, 32, generator=g).to(dev)
    w1 = (torch.randn(32, 256, generator=g) / 32 ** 0.5).to(dev)
    b1 = (0.3 * torch.randn(256, generator=g)).to(dev)
    w2 = (torch.randn(256, D, generator=g) / 16).to(dev)
    b2 = (0.3 * torch.randn(D, generator=g)).to(dev)
    dec = torch.empty(nvox, D, device=dev)
    if what == "pw_fwd":
        outs = {}
        for impl in (2, 3, 4):
            for _ in range(reps):
                L.check(L.lib().probav_pw_forward(L.ptr(x), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2), L.ptr(dec), nvox, 0, D, impl,
                                                  L.current_stream()))
            torch.cuda.synchronize()
            outs[impl] = dec.double().cpu()
        ref = torch.relu(x.double().cpu() @ w1.double().cpu() + b1.double().cpu()) @ w2.double().cpu() + b2.double().cpu()
        for impl, o in outs.items():
            print("impl %d: max err %.3g  rms err %.3g (relative to max |ref|)" % (
                impl, float((o - ref).abs().max() / ref.abs().max()), float((o - ref).pow(2).mean().sqrt() / ref.abs().max())))


    if what == "pw_bwd":
        ddec = torch.randn(nvox, D, generator=g).to(dev)
        dskip = torch.randn(nvox, 32, generator=g).to(dev)
        nbytes = L.lib().probav_pw_backward_scratch_bytes(D)
        scratch = torch.empty(nbytes // 4 + 1, device=dev)
        dx = torch.empty(nvox, 32, device=dev)
        dw1, db1, dw2, db2 = (torch.empty(s, device=dev) for s in ((32, 256), (256,), (256, D), (D,)))
        res = {}
        for impl in (2, 3, 4):
            for _ in range(reps):
                L.check(L.lib().probav_pw_backward(L.ptr(x), L.ptr(ddec), L.ptr(dskip), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(dx), L.ptr(dw1),
                                                   L.ptr(db1), L.ptr(dw2), L.ptr(db2), L.ptr(scratch), nbytes, nvox, 0, D, impl, L.current_stream()))
            torch.cuda.synchronize()
            res[impl] = [t.double().cpu().clone() for t in (dx, dw1, db1, dw2, db2)]
        for a, b, c, name in zip(res[2], res[3], res[4], ("dx", "dw1", "db1", "dw2", "db2")):
            print("%s: impl 3 vs impl 2 max diff / max = %.3g ; impl 4 vs impl 2 = %.3g" % (
                name, float((a - b).abs().max() / a.abs().max()), float((a - c).abs().max() / a.abs().max())))


    if what == "wgrad":
        import ctypes
        N, H, W, T, Cin, Cout = 128, 22, 22, 9, 25, 32
        gm = (ctypes.c_int32 * 17)(N, H, W, T, Cin, H, W, T, Cout, 3, 3, 3, 1, 1, 1, 0, 0)
        xx = torch.randn(N, H, W, T, Cin, generator=g).to(dev)
        dyy = torch.randn(N, H, W, T, Cout, generator=g).to(dev)
        outs = {}
        for impl in (1, 3, 4):
            nbytes = L.lib().probav_conv3d_wgrad_scratch_bytes(ctypes.byref(gm), impl)
            scratch = torch.empty(nbytes // 4 + 1, device=dev)
            dw = torch.empty(3, 3, 3, Cin, Cout, device=dev)
            db = torch.empty(Cout, device=dev)
            for _ in range(reps):
                L.check(L.lib().probav_conv3d_wgrad(ctypes.byref(gm), L.ptr(xx), L.ptr(dyy), None, L.ptr(dw), L.ptr(db), L.ptr(scratch), nbytes,
                                                    impl, L.current_stream()))
            torch.cuda.synchronize()
            outs[impl] = dw.double().cpu()
        print("wgrad impl 3 vs impl 1: max diff / max = %.3g ; impl 4 vs impl 1 = %.3g" % (
            float((outs[3] - outs[1]).abs().max() / outs[1].abs().max()), float((outs[4] - outs[1]).abs().max() / outs[1].abs().max())))


if __name__ == "__main__":
    main()
