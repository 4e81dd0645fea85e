"""ctypes binding of csrc/libprobav_hip.so (the C ABI declared in include/probav_hip.h).

There is no fallback: if the shared library has not been built, or a compute entry point is handed a
tensor that does not live on a HIP device, the call raises.  Build with
``python -c "import __graft_entry__ as g; g.build()"`` (hipcc --offload-arch=gfx950).
"""
import ctypes
import os
from ctypes import POINTER, c_char, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint8, c_void_p

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(CSRC, "libprobav_hip.so")

PROBAV_OK, PROBAV_EINVAL, PROBAV_ENOSPACE, PROBAV_EHIP = 0, -1, -2, -3


class NetCfg(ctypes.Structure):
    """struct probav_net_cfg (include/probav_hip.h)."""
    _fields_ = [("scale", c_int32), ("num_filters", c_int32), ("num_res_blocks", c_int32), ("exp_rate", c_int32),
                ("dec_channels", c_int32), ("num_img_lr", c_int32), ("patch_size_lr", c_int32),
                ("max_shift", c_int32), ("mean", c_float), ("std", c_float), ("in_channels", c_int32)]


# every symbol include/probav_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "probav_abi_version": (c_int, []),
    "probav_last_error": (c_char_p, []),
    "probav_engine_create": (c_int, [POINTER(NetCfg), POINTER(c_void_p)]),
    "probav_engine_destroy": (None, [c_void_p]),
    "probav_param_count": (c_int64, [c_void_p]),
    "probav_num_layers": (c_int, [c_void_p]),
    "probav_layer_info": (c_int, [c_void_p, c_int, POINTER(c_char * 32), POINTER(c_int64), POINTER(c_int64),
                                  POINTER(c_int64), POINTER(c_int32 * 5)]),
    "probav_engine_set_impl": (c_int, [c_void_p, c_int]),
    "probav_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "probav_engine_profile": (c_int, [c_void_p, c_int, c_int]),
    "probav_engine_profile_classes": (c_int, [c_void_p, ctypes.c_uint32]),
    "probav_engine_profile_read": (c_int, [c_void_p, c_int, POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(c_int64)]),
    "probav_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "probav_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "probav_shift_loss_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "probav_shift_loss_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                           c_void_p, c_void_p, c_void_p]),
    "probav_shift_l1edge_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "probav_shift_l1edge_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "probav_revssim_scratch_bytes": (c_size_t, [c_int, c_int]),
    "probav_revssim_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "probav_revssim_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "probav_clip_round": (c_int, [c_void_p, c_void_p, c_size_t, c_float, c_float, c_void_p]),
    "probav_nadam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                                  c_float, c_float, c_float, c_void_p]),
    "probav_conv3d_forward": (c_int, [POINTER(c_int32 * 17), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_int, c_void_p]),
    "probav_conv3d_wgrad_scratch_bytes": (c_size_t, [POINTER(c_int32 * 17), c_int]),
    "probav_conv3d_wgrad": (c_int, [POINTER(c_int32 * 17), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_size_t, c_int, c_void_p]),
    "probav_pw_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    "probav_pw_backward_scratch_bytes": (c_size_t, [c_int]),
    "probav_pw_backward": (c_int, [c_void_p] * 12 + [c_size_t, c_int64, c_int64, c_int, c_int, c_void_p]),
    "probav_weff_count": (c_int64, [c_void_p]),
    "probav_cout_total": (c_int64, [c_void_p]),
    "probav_wn_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "probav_wn_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "probav_workspace_view": (c_int, [c_void_p, c_int, c_int, c_int, c_int, POINTER(c_int64), POINTER(c_int64)]),
    "probav_engine_side_stream": (c_int, [c_void_p, c_int]),
    "probav_weight_cache_build": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "probav_mfma_probe": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "probav_mfma_probe_shape": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "probav_debug_hidden": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "probav_debug_hidden_from_forward_kernel": (c_int, [c_int]),
    "probav_workspace_split": (c_int, [c_void_p, c_int, POINTER(c_size_t), POINTER(c_size_t)]),
    "probav_backward_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_int, c_void_p, c_size_t, c_void_p]),
    "probav_weight_cache_bytes": (c_size_t, [c_void_p]),
    "probav_optimizer_step_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_float, c_float, c_float,
                                            c_void_p, c_size_t, c_void_p]),
    "probav_forward_wc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "probav_backward_wc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_size_t, c_void_p]),
}

_lib = None


def lib():
    """Load the shared library once; raise loudly if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "HIP library %s is missing: build it with `python -c \"import __graft_entry__ as g; g.build()\"` "
                "(there is no CPU/PyTorch fallback for the WDSR-B hot path)" % LIB_PATH)
        # torch first: its wheel carries its own HIP runtime, and the process must end up with ONE libamdhip64 -- the one torch's
        # allocator and streams live in.  Loaded the other way round, this library binds to the system copy, which then owns no device
        # ("no ROCm-capable device is detected" from the first hipMemcpy of probav_engine_create).
        import torch                                        # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)            # AttributeError if the library does not export it
            fn.restype, fn.argtypes = res, args
        if L.probav_abi_version() != 7:
            raise RuntimeError("libprobav_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc, what=""):
    """Map the C ABI's return codes onto Python exceptions."""
    if rc == PROBAV_OK:
        return
    msg = lib().probav_last_error()
    msg = msg.decode() if msg else ""
    text = "%s failed (%d): %s" % (what or "libprobav_hip call", rc, msg)
    if rc == PROBAV_EINVAL:
        raise ValueError(text)
    if rc == PROBAV_ENOSPACE:
        raise MemoryError(text)
    raise RuntimeError(text)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def current_stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def require_device(t, name):
    import torch
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor, got %r" % (name, type(t)))
    if not t.is_cuda:
        raise RuntimeError("%s lives on %s: the WDSR-B hot path runs only as HIP kernels on a gfx950 device "
                           "(no CPU fallback)" % (name, t.device))
    return t
