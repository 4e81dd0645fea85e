"""Introspection of a forward pass for parity checks (used by tests/ and __graft_entry__.smoke(); nothing on the product path calls it).

The gradient of the network is discontinuous in its ReLU gates; an fp32 and an fp64 evaluation decide the few gates whose pre-activation
is ~0 differently.  `device_gates` reads the decisions the HIP forward pass actually took, so that a checker can evaluate its own
gradient at the same gates and compare what is left: the arithmetic of the kernels (1e-3 of the per-tensor max norm).
"""
import ctypes

import torch

from . import _lib


def device_gates(m, flat_used, B, T=9):
    """{layer: bool array}: the ReLU decisions of the last training forward of `m` (batch B, T frames), read from the saved activations
    (a post-ReLU value is > 0 exactly where the gate is open: `probav_workspace_view`) and, for the 256-channel hidden tiles that never
    reach memory, recomputed by the forward kernel itself (`probav_debug_hidden`).  Needs the pass's workspace alive: run the forward
    with PROBAV_KEEP_WS=1, or call this before the backward pass releases it."""
    L = _lib.lib()
    h, ws = m._handle(), m._workspace(B, True)
    wc = m.weight_cache()                    # the cache the forward pass ran from (None: it recomputed the weights into its workspace)
    hin = m.patchSizeLR + m.maxShift

    def view(kind, idx):
        off, cnt = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(L.probav_workspace_view(h, B, 1, kind, idx, ctypes.byref(off), ctypes.byref(cnt)), "probav_workspace_view")
        return ws[off.value: off.value + cnt.value]
    gates = {"mainConv1": (view(0, 0) > 0).cpu().numpy(), "residConv1": (view(3, 0) > 0).cpu().numpy()}
    nvox = B * hin * hin * T
    hid = torch.empty(nvox * m.numFilters * m.expRate, device=ws.device)
    dec = torch.empty(nvox * 32, device=ws.device)                     # the launch's regular output (decConv), discarded
    for i in range(m.numResBlocks):
        _lib.check(L.probav_debug_hidden(h, _lib.ptr(flat_used), _lib.ptr(ws), ws.numel() * 4, B, i, _lib.ptr(hid), _lib.ptr(dec), _lib.ptr(wc),
                                         _lib.current_stream()), "probav_debug_hidden")
        gates["expConv_%d" % i] = (hid > 0).cpu().numpy()
    k = 0
    while True:
        try:
            gates["convReducer_%d" % (k + 1)] = (view(2, k) > 0).cpu().numpy()
        except ValueError:
            break
        k += 1
    return gates
