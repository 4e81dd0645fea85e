"""Introspection of a forward pass for parity checks (used by tests/ and __graft_entry__.smoke(); nothing on the product path calls it).

The gradient of the network is discontinuous in its ReLU gates; an fp32 and an fp64 evaluation decide the few gates whose pre-activation
is ~0 differently.  `device_gates` reads the decisions the HIP forward pass actually took, so that a checker can evaluate its own
gradient at the same gates and compare what is left: the arithmetic of the kernels (1e-3 of the per-tensor max norm).
"""
import ctypes

import torch

from . import _lib


def device_gates(m, flat_used, B, T=9, samples=None):
    """{layer: bool array}: the ReLU decisions of the last training forward of `m` (batch B, T frames), read from the saved activations
    (a post-ReLU value is > 0 exactly where the gate is open: `probav_workspace_view`) and, for the 256-channel hidden tiles that never
    reach memory, recomputed by the forward kernel itself (`probav_debug_hidden`).  Needs the pass's workspace alive: run the forward
    with PROBAV_KEEP_WS=1, or call this before the backward pass releases it.
    samples (optional): only these samples of the batch, in this order -- the gates of a sub-batch of a large batch (every saved tensor is
    sample-major), sliced on the device.
    The hidden tiles come from the 32x32x16 arrangement of the fused forward kernel, which is the one the BACKWARD pass recomputes them
    with: these are the gates the gradient was taken at.  The forward pass proper (pw_fwd_h3k_kernel: 16x16x32) sums the same products in
    another order; a pre-activation that is zero to the last bit can be open in one and closed in the other -- its forward contribution
    is its value, ~0 (csrc/kernels_x6.hip, the comment above that kernel)."""
    L = _lib.lib()
    h, ws = m._handle(), m._workspace(B, True)
    wc = m.weight_cache()                    # the cache the forward pass ran from (None: it recomputed the weights into its workspace)
    hin = m.patchSizeLR + m.maxShift

    def view(kind, idx):
        off, cnt = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(L.probav_workspace_view(h, B, 1, kind, idx, ctypes.byref(off), ctypes.byref(cnt)), "probav_workspace_view")
        return ws[off.value: off.value + cnt.value]
    def pick(t):                                     # [B * per] (or any sample-major flat tensor) -> the chosen samples
        if samples is None:
            return t
        return t.view(B, -1)[torch.as_tensor(list(samples), device=t.device)].reshape(-1)
    gates = {"mainConv1": (pick(view(0, 0)) > 0).cpu().numpy(), "residConv1": (pick(view(3, 0)) > 0).cpu().numpy()}
    nvox = B * hin * hin * T
    hid = torch.empty(nvox * m.numFilters * m.expRate, device=ws.device)
    dec = torch.empty(nvox * 32, device=ws.device)                     # the launch's regular output (decConv), discarded
    for i in range(m.numResBlocks):
        _lib.check(L.probav_debug_hidden(h, _lib.ptr(flat_used), _lib.ptr(ws), ws.numel() * 4, B, i, _lib.ptr(hid), _lib.ptr(dec), _lib.ptr(wc),
                                         _lib.current_stream()), "probav_debug_hidden")
        gates["expConv_%d" % i] = (pick(hid) > 0).cpu().numpy()
    k = 0
    while True:
        try:
            gates["convReducer_%d" % (k + 1)] = (pick(view(2, k)) > 0).cpu().numpy()
        except ValueError:
            break
        k += 1
    return gates


def hidden_tile(m, flat_used, B, block, T=9, from_forward_kernel=False):
    """relu(expConv_block(x)) [B, voxels, 256] of the last training forward of `m`, each sample at its hidden tile's own power-of-two scale, as the
    32x32x16 arrangement evaluates it (default: the order of additions the reverse pass recomputes the tile -- and decides its gates -- in) or as the
    forward kernel itself does (pw_fwd_h3k_kernel; `probav_debug_hidden_from_forward_kernel`)."""
    L = _lib.lib()
    h, ws = m._handle(), m._workspace(B, True)
    wc = m.weight_cache()
    hin = m.patchSizeLR + m.maxShift
    nvox = B * hin * hin * T
    hid = torch.empty(nvox * m.numFilters * m.expRate, device=ws.device)
    dec = torch.empty(nvox * 32, device=ws.device)
    _lib.check(L.probav_debug_hidden_from_forward_kernel(1 if from_forward_kernel else 0), "probav_debug_hidden_from_forward_kernel")
    try:
        _lib.check(L.probav_debug_hidden(h, _lib.ptr(flat_used), _lib.ptr(ws), ws.numel() * 4, B, block, _lib.ptr(hid), _lib.ptr(dec), _lib.ptr(wc),
                                         _lib.current_stream()), "probav_debug_hidden")
    finally:
        _lib.check(L.probav_debug_hidden_from_forward_kernel(0), "probav_debug_hidden_from_forward_kernel")
    return hid.view(B, -1, m.numFilters * m.expRate)
