"""torch custom ops (`torch.ops.probav.*`) over the C ABI of libprobav_hip.so -- the north-star boundary: the hot path is
"hand-written HIP kernels exposed as torch custom ops"; PyTorch owns device memory, the stream and the autograd graph, nothing else.

Registered with `torch.library.custom_op` (schema, dispatcher entry, fake-tensor rule, autograd formula), so they are visible to the
dispatcher, `torch.compile` / AOT-autograd and `torch.library.opcheck` (tests/test_gpu_ops.py):

  probav::wdsr_forward(flat, x, engine, out_size, training) -> (y, ws)      model(x, training=...)      models/trainClass.py:127,139
  probav::wdsr_backward(flat, dy, ws, engine) -> dflat                      tape.gradient(loss, vars)   models/trainClass.py:131
  probav::shift_loss(pred, hr, mask, border, bit_depth, which) -> (loss, arg, per_sample)
                                                                            Losses.shiftCompensatedL1Loss / L2Loss   models/loss.py:55-84
  probav::shift_loss_backward(hr, mask, pred, arg, upstream, border, which) -> dpred
  probav::shift_metrics(hr, mask, pred, border, bit_depth) -> (f[3,B], arg[2,B], means[2])     one launch: L1, L2, cPSNR of every sample
  probav::nadam_step(theta, grad, m, v, lr, b1, b2, eps, c_g, c_m, c_v) -> ()                 optimizer.apply_gradients   trainClass.py:132
  probav::optimizer_wn_step(theta, grad, m, v, wcache, engine, lr, ...) -> ()                 the same update fused with the weight normalisation
                                                                            and operand packing of the NEXT step (wdsr_forward's optional `wcache`)
  probav::clip_round(x, lo, hi) -> y                                        tf.clip_by_value + tf.round   test.py:118-119

`engine` is the probav_engine* of include/probav_hip.h as an integer (the ops are stateless; the handle owns only the layer table),
`ws` the workspace of one forward call: an OUTPUT of wdsr_forward (it carries the activations to the reverse pass, like the residuals of
any differentiable op; torch's caching allocator recycles the block from step to step), an input of wdsr_backward.  Every op raises on CPU tensors:
there is no fallback implementation.
"""
from ctypes import c_void_p
from typing import Optional

import torch
from torch import Tensor

from . import _lib

def _dev(t, name):
    return _lib.require_device(t, name)


# ---------------------------------------------------------------------------------------------------------------------------------
# the network
# ---------------------------------------------------------------------------------------------------------------------------------
_WS_POOLS = {}


def _ws_pool(device, kind="ws"):
    """A private, never-split allocator pool for the engine workspaces (torch.cuda.MemPool): a multi-GB block that returns to the general
    pool gets carved up by the next megabyte-sized request, and the following step then pays a hipMalloc of the full size (~80 ms,
    device-synchronous).  In its own pool a freed workspace block can only be taken by the next workspace.  The reverse pass's scratch
    blocks (kind "scratch") have a pool of their own: a freed 2 GB scratch block must not be handed to the next 2 GB workspace request
    while the following scratch request then finds nothing."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), kind)
    pool = _WS_POOLS.get(key)
    if pool is None:
        try:
            pool = torch.cuda.MemPool(no_split=True)
        except TypeError:                                  # (a torch without the keyword: blocks of this pool may then be split)
            pool = torch.cuda.MemPool()
        _WS_POOLS[key] = pool
    return pool


def release_workspaces(device=None):
    """Return the workspace blocks to the driver.  Every distinct (batch, training, frames) shape pins its own multi-GB block in the
    private pool for as long as the pool lives (a trainer alternating training batches, a partial last batch, validation batches and
    inference micro-batches holds one block each); call this between such phases of a long-lived process.  Blocks still referenced
    (an un-run backward) are freed when their tensors die."""
    dev = None if device is None else (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device())
    for k in [k for k in _WS_POOLS if dev is None or k[0] == dev]:
        _WS_POOLS.pop(k, None)
    torch.cuda.empty_cache()


def _ws_floats(engine, batch, training):
    """Floats of the workspace wdsr_forward returns: in training mode the SAVED STATE of the pass only (probav_workspace_split) -- the reverse
    pass brings its own scratch."""
    import ctypes
    L = _lib.lib()
    if training:
        saved, scratch = ctypes.c_size_t(), ctypes.c_size_t()
        _lib.check(L.probav_workspace_split(c_void_p(engine), int(batch), ctypes.byref(saved), ctypes.byref(scratch)), "probav_workspace_split")
        nbytes = saved.value
    else:
        nbytes = L.probav_workspace_bytes(c_void_p(engine), int(batch), 0)
    if nbytes == 0:
        raise RuntimeError("probav_workspace_bytes returned 0")
    return (nbytes + 3) // 4


def _scratch_floats(engine, batch):
    import ctypes
    saved, scratch = ctypes.c_size_t(), ctypes.c_size_t()
    _lib.check(_lib.lib().probav_workspace_split(c_void_p(engine), int(batch), ctypes.byref(saved), ctypes.byref(scratch)), "probav_workspace_split")
    return (scratch.value + 3) // 4


@torch.library.custom_op("probav::wdsr_forward", mutates_args=(), device_types="cuda")
def wdsr_forward(flat: Tensor, x: Tensor, engine: int, out_size: int, training: bool, wcache: Optional[Tensor] = None) -> tuple[Tensor, Tensor]:
    """-> (y [B, out_size, out_size, 1], ws): ws is the engine workspace of this call -- with training=True it holds the activations the
    reverse pass needs (the op is functional: the saved state is an OUTPUT, like the residuals of any differentiable op).
    wcache (optional): the weight cache probav::optimizer_wn_step filled for exactly these parameters; the weight-norm and packing
    launches are then skipped."""
    _dev(x, "model input")
    B = x.shape[0]
    y = torch.empty((B, out_size, out_size, 1), dtype=torch.float32, device=x.device)
    with torch.cuda.use_mem_pool(_ws_pool(x.device), device=x.device):
        ws = torch.empty(_ws_floats(engine, B, training), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    if wcache is None:
        _lib.check(L.probav_forward(c_void_p(engine), _lib.ptr(flat), _lib.ptr(x), _lib.ptr(y), _lib.ptr(ws), ws.numel() * 4, B,
                                    1 if training else 0, _lib.current_stream()), "probav_forward")
    else:
        _lib.check(L.probav_forward_wc(c_void_p(engine), _lib.ptr(flat), _lib.ptr(x), _lib.ptr(y), _lib.ptr(ws), ws.numel() * 4, B,
                                       1 if training else 0, _lib.ptr(wcache), wcache.numel() * 4, _lib.current_stream()), "probav_forward_wc")
    return y, ws


@wdsr_forward.register_fake
def _(flat, x, engine, out_size, training, wcache=None):
    B = x.shape[0]
    return (x.new_empty((B, out_size, out_size, 1), dtype=torch.float32), x.new_empty((_ws_floats(engine, int(B), training),), dtype=torch.float32))


@torch.library.custom_op("probav::wdsr_backward", mutates_args=(), device_types="cuda")
def wdsr_backward(flat: Tensor, dy: Tensor, ws: Tensor, engine: int, wcache: Optional[Tensor] = None) -> Tensor:
    """d loss / d flat from d loss / d y; `ws` = the saved state the matching forward returned, only READ here (probav_backward_split): the
    reverse pass's gradient buffers and partial-sum slabs live in a scratch block of this call's own, from the same private pool as the
    workspaces.  The op is functional -- a second backward over the same graph (retain_graph, a gradient check) reads the same
    activations, and AOT-autograd traces the formula of wdsr_forward without any storage aliasing.  wcache = the weight cache that
    forward ran from, if any."""
    _dev(dy, "output gradient")
    grads = torch.empty_like(flat)
    B = dy.shape[0]
    with torch.cuda.use_mem_pool(_ws_pool(dy.device, "scratch"), device=dy.device):
        scratch = torch.empty(_scratch_floats(engine, B), dtype=torch.float32, device=dy.device)
    _lib.check(_lib.lib().probav_backward_split(c_void_p(engine), _lib.ptr(flat), _lib.ptr(dy), _lib.ptr(grads), _lib.ptr(ws), ws.numel() * 4,
                                                _lib.ptr(scratch), scratch.numel() * 4, B, _lib.ptr(wcache),
                                                0 if wcache is None else wcache.numel() * 4, _lib.current_stream()), "probav_backward_split")
    return grads


@wdsr_backward.register_fake
def _(flat, dy, ws, engine, wcache=None):
    return torch.empty_like(flat)


def _wdsr_setup(ctx, inputs, output):
    flat, x, engine, out_size, training, wcache = inputs
    ctx.wcache = wcache
    # the workspace is an OUTPUT of this node: it must go through save_for_backward (a plain attribute would close the reference cycle
    # node -> ctx -> ws -> grad_fn -> node and every step's 3 GB would stay alive until the cycle collector runs, if ever)
    ctx.save_for_backward(flat, output[1])
    ctx.engine, ctx.training = engine, training
    ctx.mark_non_differentiable(output[1])
    ctx.set_materialize_grads(False)           # ... and it never carries a gradient: do not let autograd zero-fill 3 GB for it


def _wdsr_bwd(ctx, dy, dws):
    if dy is None:
        return None, None, None, None, None, None
    if not ctx.training:
        raise RuntimeError("backward through model(x, training=False): call the model with training=True "
                           "to keep the activations the reverse pass needs")
    flat, ws = ctx.saved_tensors
    g = torch.ops.probav.wdsr_backward(flat, dy.contiguous().float(), ws, ctx.engine, ctx.wcache)
    return g, None, None, None, None, None


wdsr_forward.register_autograd(_wdsr_bwd, setup_context=_wdsr_setup)


# ---------------------------------------------------------------------------------------------------------------------------------
# shift-compensated loss / metric
# ---------------------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op("probav::shift_metrics", mutates_args=(), device_types="cuda")
def shift_metrics(hr: Tensor, mask: Tensor, pred: Tensor, border: int, bit_depth: int) -> tuple[Tensor, Tensor, Tensor]:
    _dev(pred, "predPatchHR")
    B, S = pred.shape[0], pred.shape[1]
    dev = pred.device
    f = torch.empty((3, B), dtype=torch.float32, device=dev)          # l1 | l2 | cpsnr
    arg = torch.empty((2, B), dtype=torch.int32, device=dev)
    means = torch.empty(2, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().probav_shift_loss_forward(
        _lib.ptr(hr), _lib.ptr(mask), _lib.ptr(pred), B, S, border, bit_depth, _lib.ptr(f[0]), _lib.ptr(f[1]),
        _lib.ptr(f[2]), _lib.ptr(arg[0]), _lib.ptr(arg[1]), _lib.ptr(means[0:1]), _lib.ptr(means[1:2]),
        _lib.current_stream()), "probav_shift_loss_forward")
    return f, arg, means


@shift_metrics.register_fake
def _(hr, mask, pred, border, bit_depth):
    B = pred.shape[0]
    return (pred.new_empty((3, B), dtype=torch.float32), pred.new_empty((2, B), dtype=torch.int32), pred.new_empty((2,), dtype=torch.float32))


@torch.library.custom_op("probav::shift_loss_backward", mutates_args=(), device_types="cuda")
def shift_loss_backward(hr: Tensor, mask: Tensor, pred: Tensor, arg: Tensor, upstream: Tensor, border: int, which: int) -> Tensor:
    _dev(pred, "predPatchHR")
    dpred = torch.empty_like(pred)
    _lib.check(_lib.lib().probav_shift_loss_backward(
        _lib.ptr(hr), _lib.ptr(mask), _lib.ptr(pred), _lib.ptr(arg), pred.shape[0], pred.shape[1], border,
        which, _lib.ptr(upstream), _lib.ptr(dpred), _lib.current_stream()), "probav_shift_loss_backward")
    return dpred


@shift_loss_backward.register_fake
def _(hr, mask, pred, arg, upstream, border, which):
    return torch.empty_like(pred)


@torch.library.custom_op("probav::shift_loss", mutates_args=(), device_types="cuda")
def shift_loss(pred: Tensor, hr: Tensor, mask: Tensor, border: int, bit_depth: int, which: int) -> tuple[Tensor, Tensor, Tensor]:
    """which = 1: L1 (models/loss.py:73-84), 2: L2 (:55-71) -> (batch-mean loss [scalar], arg-min shift per sample [B], per-sample minima [B])."""
    # the launch writes through seven separate pointers: the three results asked for go straight into tensors of their own (no copies: three
    # launches less per step), the other four into one scratch allocation
    _dev(pred, "predPatchHR")
    B, S = pred.shape[0], pred.shape[1]
    dev = pred.device
    loss = torch.empty((), dtype=torch.float32, device=dev)
    amin = torch.empty((B,), dtype=torch.int32, device=dev)
    per = torch.empty((B,), dtype=torch.float32, device=dev)
    scr = torch.empty((2 * B + 1,), dtype=torch.float32, device=dev)        # the other per-sample loss, cPSNR, the other mean
    sarg = torch.empty((B,), dtype=torch.int32, device=dev)
    l1 = which == 1
    _lib.check(_lib.lib().probav_shift_loss_forward(
        _lib.ptr(hr), _lib.ptr(mask), _lib.ptr(pred), B, S, border, bit_depth,
        _lib.ptr(per if l1 else scr[0:B]), _lib.ptr(scr[0:B] if l1 else per), _lib.ptr(scr[B:2 * B]),
        _lib.ptr(amin if l1 else sarg), _lib.ptr(sarg if l1 else amin),
        _lib.ptr(loss if l1 else scr[2 * B:]), _lib.ptr(scr[2 * B:] if l1 else loss),
        _lib.current_stream()), "probav_shift_loss_forward")
    return loss, amin, per


@shift_loss.register_fake
def _(pred, hr, mask, border, bit_depth, which):
    B = pred.shape[0]
    return (pred.new_empty((), dtype=torch.float32), pred.new_empty((B,), dtype=torch.int32), pred.new_empty((B,), dtype=torch.float32))


def _shift_setup(ctx, inputs, output):
    pred, hr, mask, border, bit_depth, which = inputs
    ctx.save_for_backward(pred, hr, mask, output[1])
    ctx.border, ctx.which = border, which
    ctx.set_materialize_grads(False)


def _shift_bwd(ctx, g_loss, g_arg, g_per):
    if g_loss is None:                         # (only the batch-mean loss is differentiable; the per-sample minima are a by-product)
        return None, None, None, None, None, None
    pred, hr, mask, arg = ctx.saved_tensors
    dpred = torch.ops.probav.shift_loss_backward(hr, mask, pred, arg, g_loss.contiguous().float().reshape(1), ctx.border, ctx.which)
    return dpred, None, None, None, None, None


shift_loss.register_autograd(_shift_bwd, setup_context=_shift_setup)


# ---------------------------------------------------------------------------------------------------------------------------------
# optimizer update, inference epilogue
# ---------------------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op("probav::nadam_step", mutates_args=("theta", "m", "v"), device_types="cuda")
def nadam_step(theta: Tensor, grad: Tensor, m: Tensor, v: Tensor, lr: float, beta_1: float, beta_2: float, eps: float,
               c_g: float, c_m: float, c_v: float) -> None:
    _dev(theta, "parameter")
    _lib.check(_lib.lib().probav_nadam_step(_lib.ptr(theta), _lib.ptr(grad), _lib.ptr(m), _lib.ptr(v), theta.numel(), lr, beta_1, beta_2, eps,
                                            c_g, c_m, c_v, _lib.current_stream()), "probav_nadam_step")


@nadam_step.register_fake
def _(theta, grad, m, v, lr, beta_1, beta_2, eps, c_g, c_m, c_v):
    return None


@torch.library.custom_op("probav::optimizer_wn_step", mutates_args=("theta", "m", "v", "wcache"), device_types="cuda")
def optimizer_wn_step(theta: Tensor, grad: Tensor, m: Tensor, v: Tensor, wcache: Tensor, engine: int, lr: float, beta_1: float, beta_2: float,
                      eps: float, c_g: float, c_m: float, c_v: float) -> None:
    """nadam_step on the engine's flat parameter buffer, fused with the weight normalisation (and operand packing) of the UPDATED
    parameters into `wcache` (probav_weight_cache_bytes): the next wdsr_forward(..., wcache) starts at its first convolution."""
    _dev(theta, "parameter")
    _lib.check(_lib.lib().probav_optimizer_step_fused(c_void_p(engine), _lib.ptr(theta), _lib.ptr(grad), _lib.ptr(m), _lib.ptr(v), lr, beta_1, beta_2,
                                                      eps, c_g, c_m, c_v, _lib.ptr(wcache), wcache.numel() * 4, _lib.current_stream()),
               "probav_optimizer_step_fused")


@optimizer_wn_step.register_fake
def _(theta, grad, m, v, wcache, engine, lr, beta_1, beta_2, eps, c_g, c_m, c_v):
    return None


@torch.library.custom_op("probav::weight_cache_build", mutates_args=("wcache",), device_types="cuda")
def weight_cache_build(theta: Tensor, wcache: Tensor, engine: int) -> None:
    """Weight normalisation + operand packing of the CURRENT parameters into `wcache`, for forwards on weights nothing is updating."""
    _dev(theta, "parameter")
    _lib.check(_lib.lib().probav_weight_cache_build(c_void_p(engine), _lib.ptr(theta), _lib.ptr(wcache), wcache.numel() * 4, _lib.current_stream()),
               "probav_weight_cache_build")


@weight_cache_build.register_fake
def _(theta, wcache, engine):
    return None


@torch.library.custom_op("probav::clip_round", mutates_args=(), device_types="cuda")
def clip_round(x: Tensor, lo: float, hi: float) -> Tensor:
    _dev(x, "clip_round input")
    out = torch.empty_like(x)
    _lib.check(_lib.lib().probav_clip_round(_lib.ptr(x), _lib.ptr(out), x.numel(), lo, hi, _lib.current_stream()), "probav_clip_round")
    return out


@clip_round.register_fake
def _(x, lo, hi):
    return torch.empty_like(x)


# ---------------------------------------------------------------------------------------------------------------------------------
# per-layer operators (SURVEY.md section 8b's minimum op set): what the engine is made of, as torch ops of their own, so that a variant
# network (another block order, an iWDSR-style head) can be composed from them.  Each has its autograd formula registered in terms of
# the other ops of the set; the whole-network op above stays the fast path (one workspace, fused scales, side stream).
# Layouts are the reference's: activations [N, H, W, T, C] (channels last), filters in Keras layout [kh, kw, kt, Cin, Cout].
# ---------------------------------------------------------------------------------------------------------------------------------
import ctypes as _ct


def _geom17(N, hwt, cin, out_hwt, cout, k, pad, reflect, relu):
    return (_ct.c_int32 * 17)(N, hwt[0], hwt[1], hwt[2], cin, out_hwt[0], out_hwt[1], out_hwt[2], cout, k[0], k[1], k[2], pad[0], pad[1], pad[2],
                              1 if reflect else 0, 1 if relu else 0)


def _conv_out(x, w, pad):
    k = tuple(w.shape[:3])
    out = tuple(int(x.shape[1 + i]) + 2 * pad[i] - k[i] + 1 for i in range(3))
    return k, out


@torch.library.custom_op("probav::conv3d_k3_fwd", mutates_args=(), device_types="cuda")
def conv3d_k3_fwd(x: Tensor, w: Tensor, bias: Tensor, pad: list[int], reflect_hw: bool, relu: bool, skip: Optional[Tensor] = None,
                  gate: Optional[Tensor] = None, impl: int = 4) -> Tensor:
    """y = act(conv(x * [gate > 0], w) + bias) + skip: Conv3D of models/modelsTF.py:168-183,187-203 (`same`: pad 1, `valid`: pad 0; reflect_hw: the
    tf.pad(REFLECT) of the height / width axes in front of convReducer_1) -- and, with flipped taps and swapped channels, its backward-data."""
    _dev(x, "conv input")
    k, out = _conv_out(x, w, pad)
    y = torch.empty((x.shape[0],) + out + (w.shape[4],), dtype=torch.float32, device=x.device)
    g = _geom17(x.shape[0], x.shape[1:4], x.shape[4], out, w.shape[4], k, pad, reflect_hw, relu)
    rc = _lib.lib().probav_conv3d_forward(_ct.byref(g), _lib.ptr(x), _lib.ptr(gate), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(skip), _lib.ptr(y), impl,
                                          _lib.current_stream())
    if rc == _lib.PROBAV_EINVAL and impl != 0:             # a geometry this MFMA family does not cover: the shape-agnostic kernels, as the engine does
        rc = _lib.lib().probav_conv3d_forward(_ct.byref(g), _lib.ptr(x), _lib.ptr(gate), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(skip), _lib.ptr(y), 0,
                                              _lib.current_stream())
    _lib.check(rc, "probav_conv3d_forward")
    return y


@conv3d_k3_fwd.register_fake
def _(x, w, bias, pad, reflect_hw, relu, skip=None, gate=None, impl=4):
    k, out = _conv_out(x, w, pad)
    return x.new_empty((x.shape[0],) + out + (w.shape[4],), dtype=torch.float32)


@torch.library.custom_op("probav::conv3d_k3_bwd_weight", mutates_args=(), device_types="cuda")
def conv3d_k3_bwd_weight(x: Tensor, dy: Tensor, ksize: list[int], pad: list[int], reflect_hw: bool, gate: Optional[Tensor] = None,
                         impl: int = 4) -> tuple[Tensor, Tensor]:
    """(dw [kh, kw, kt, Cin, Cout], db [Cout]) of the same layer from its input and the gradient of its output (gate: the layer's own
    post-ReLU output, when it has one)."""
    _dev(x, "conv input")
    out = tuple(int(dy.shape[1 + i]) for i in range(3))
    g = _geom17(x.shape[0], x.shape[1:4], x.shape[4], out, dy.shape[4], ksize, pad, reflect_hw, gate is not None)
    L = _lib.lib()
    use = impl
    nbytes = L.probav_conv3d_wgrad_scratch_bytes(_ct.byref(g), use)
    if nbytes == 0 and use != 0:
        use = 0
        nbytes = L.probav_conv3d_wgrad_scratch_bytes(_ct.byref(g), 0)
    scratch = torch.empty(nbytes // 4 + 1, dtype=torch.float32, device=x.device)
    dw = torch.empty(tuple(ksize) + (x.shape[4], dy.shape[4]), dtype=torch.float32, device=x.device)
    db = torch.empty((dy.shape[4],), dtype=torch.float32, device=x.device)
    _lib.check(L.probav_conv3d_wgrad(_ct.byref(g), _lib.ptr(x), _lib.ptr(dy), _lib.ptr(gate), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(scratch), nbytes, use,
                                     _lib.current_stream()), "probav_conv3d_wgrad")
    return dw, db


@conv3d_k3_bwd_weight.register_fake
def _(x, dy, ksize, pad, reflect_hw, gate=None, impl=4):
    return x.new_empty(tuple(ksize) + (x.shape[4], dy.shape[4]), dtype=torch.float32), x.new_empty((dy.shape[4],), dtype=torch.float32)


def _conv_setup(ctx, inputs, output):
    x, w, bias, pad, reflect_hw, relu, skip, gate, impl = inputs
    if gate is not None:
        raise RuntimeError("probav::conv3d_k3_fwd: the `gate` form is the backward-data operator of another layer and has no autograd formula")
    ctx.save_for_backward(x, w, output if relu else None)
    ctx.pad, ctx.reflect_hw, ctx.relu, ctx.impl, ctx.has_skip = list(pad), reflect_hw, relu, impl, skip is not None


def _conv_bwd(ctx, dy):
    x, w, y = ctx.saved_tensors
    dy = dy.contiguous().float()
    k = tuple(w.shape[:3])
    gate = None
    if ctx.relu:                                            # y = relu(z) + skip: the gate is z > 0; with a skip the saved output no longer shows it
        if ctx.has_skip:
            raise RuntimeError("probav::conv3d_k3_fwd: relu together with skip has no autograd formula (the network never combines them on one layer)")
        gate = y
    dw, db = torch.ops.probav.conv3d_k3_bwd_weight(x, dy, list(k), ctx.pad, ctx.reflect_hw, gate, ctx.impl)
    dx = None
    if ctx.needs_input_grad[0]:
        if ctx.reflect_hw:
            raise RuntimeError("probav::conv3d_k3_fwd: the input gradient of a reflect-padded layer goes through the engine (fold of the mirrored border)")
        wT = torch.flip(w, dims=(0, 1, 2)).transpose(3, 4).contiguous()                 # flipped taps, swapped channels
        bpad = [k[i] - 1 - ctx.pad[i] for i in range(3)]                                 # "full" correlation minus the forward padding
        zero = torch.zeros(w.shape[3], dtype=torch.float32, device=x.device)
        dx = torch.ops.probav.conv3d_k3_fwd(dy, wT, zero, bpad, False, False, None, gate, ctx.impl)
    return dx, dw, db, None, None, None, (dy if ctx.has_skip else None), None, None


conv3d_k3_fwd.register_autograd(_conv_bwd, setup_context=_conv_setup)


@torch.library.custom_op("probav::pw_expand_relu_decay_fwd", mutates_args=(), device_types="cuda")
def pw_expand_relu_decay_fwd(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, vox_per_sample: int = 0, impl: int = 4) -> Tensor:
    """expConv_i (1x1x1, 32 -> 256) + ReLU + decConv_i (1x1x1, 256 -> D), fused: models/modelsTF.py:179-183.  x [..., 32] -> [..., D]; the
    256-channel tensor never reaches memory.  vox_per_sample: voxels of one patch (the unit the H3 arithmetic scales by; 0 = one sample)."""
    _dev(x, "pointwise input")
    nvox = x.numel() // x.shape[-1]
    dec = torch.empty(tuple(x.shape[:-1]) + (w2.shape[-1],), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().probav_pw_forward(_lib.ptr(x), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(dec), nvox, vox_per_sample,
                                            w2.shape[-1], impl, _lib.current_stream()), "probav_pw_forward")
    return dec


@pw_expand_relu_decay_fwd.register_fake
def _(x, w1, b1, w2, b2, vox_per_sample=0, impl=4):
    return x.new_empty(tuple(x.shape[:-1]) + (w2.shape[-1],), dtype=torch.float32)


@torch.library.custom_op("probav::pw_expand_relu_decay_bwd", mutates_args=(), device_types="cuda")
def pw_expand_relu_decay_bwd(x: Tensor, d_dec: Tensor, d_skip: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, vox_per_sample: int = 0,
                             impl: int = 4) -> tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """The fused reverse pass: (dx = d_skip + dL/dx, dw1, db1, dw2, db2); the hidden tile is recomputed, never stored."""
    _dev(x, "pointwise input")
    nvox, D = x.numel() // x.shape[-1], w2.shape[-1]
    L = _lib.lib()
    nbytes = L.probav_pw_backward_scratch_bytes(D)
    scratch = torch.empty(nbytes // 4 + 1, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    dw1, db1, dw2, db2 = torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), torch.empty((D,), dtype=torch.float32, device=x.device)
    _lib.check(L.probav_pw_backward(_lib.ptr(x), _lib.ptr(d_dec), _lib.ptr(d_skip), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(dx), _lib.ptr(dw1),
                                    _lib.ptr(db1), _lib.ptr(dw2), _lib.ptr(db2), _lib.ptr(scratch), nbytes, nvox, vox_per_sample, D, impl,
                                    _lib.current_stream()), "probav_pw_backward")
    return dx, dw1, db1, dw2, db2


@pw_expand_relu_decay_bwd.register_fake
def _(x, d_dec, d_skip, w1, b1, w2, vox_per_sample=0, impl=4):
    return torch.empty_like(x), torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), x.new_empty((w2.shape[-1],), dtype=torch.float32)


def _pw_setup(ctx, inputs, output):
    x, w1, b1, w2, b2, vps, impl = inputs
    ctx.save_for_backward(x, w1, b1, w2)
    ctx.vps, ctx.impl = vps, impl


def _pw_bwd(ctx, d_dec):
    x, w1, b1, w2 = ctx.saved_tensors
    d_dec = d_dec.contiguous().float()
    dx, dw1, db1, dw2, db2 = torch.ops.probav.pw_expand_relu_decay_bwd(x, d_dec, torch.zeros_like(x), w1, b1, w2, ctx.vps, ctx.impl)
    return dx, dw1, db1, dw2, db2, None, None


pw_expand_relu_decay_fwd.register_autograd(_pw_bwd, setup_context=_pw_setup)


@torch.library.custom_op("probav::wn_weight_fwd", mutates_args=(), device_types="cuda")
def wn_weight_fwd(flat: Tensor, engine: int) -> tuple[Tensor, Tensor, Tensor]:
    """TFA WeightNormalization of every layer of the engine's flat parameter buffer (w = g v / ||v||, per output channel):
    -> (weff: layers in order, Keras layout; weffT: flipped taps / swapped channels for the backward-data operators; inv_norm per output channel)."""
    _dev(flat, "parameter")
    L, h = _lib.lib(), c_void_p(engine)
    nw, nc = L.probav_weff_count(h), L.probav_cout_total(h)
    weff, weffT = torch.empty(nw, dtype=torch.float32, device=flat.device), torch.empty(nw, dtype=torch.float32, device=flat.device)
    inv = torch.empty(nc, dtype=torch.float32, device=flat.device)
    _lib.check(L.probav_wn_forward(h, _lib.ptr(flat), _lib.ptr(weff), _lib.ptr(weffT), _lib.ptr(inv), _lib.current_stream()), "probav_wn_forward")
    return weff, weffT, inv


@wn_weight_fwd.register_fake
def _(flat, engine):
    L, h = _lib.lib(), c_void_p(engine)
    nw, nc = L.probav_weff_count(h), L.probav_cout_total(h)
    return flat.new_empty((nw,)), flat.new_empty((nw,)), flat.new_empty((nc,))


@torch.library.custom_op("probav::wn_weight_bwd", mutates_args=(), device_types="cuda")
def wn_weight_bwd(flat: Tensor, dweff: Tensor, inv_norm: Tensor, engine: int) -> Tensor:
    """Gradient of the flat parameter buffer (g, v of every layer; the bias slots are left to the caller) from the gradient of `weff`."""
    _dev(flat, "parameter")
    grads = torch.zeros_like(flat)
    _lib.check(_lib.lib().probav_wn_backward(c_void_p(engine), _lib.ptr(flat), _lib.ptr(dweff), _lib.ptr(inv_norm), _lib.ptr(grads), _lib.current_stream()),
               "probav_wn_backward")
    return grads


@wn_weight_bwd.register_fake
def _(flat, dweff, inv_norm, engine):
    return torch.zeros_like(flat)


def _wn_setup(ctx, inputs, output):
    flat, engine = inputs
    ctx.save_for_backward(flat, output[2])
    ctx.engine = engine
    ctx.mark_non_differentiable(output[1], output[2])
    ctx.set_materialize_grads(False)


def _wn_bwd(ctx, dweff, dweffT, dinv):
    if dweff is None:
        return None, None
    flat, inv = ctx.saved_tensors
    return torch.ops.probav.wn_weight_bwd(flat, dweff.contiguous().float(), inv, ctx.engine), None


wn_weight_fwd.register_autograd(_wn_bwd, setup_context=_wn_setup)
