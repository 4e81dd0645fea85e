"""Host-side mirror of the reference's models/modelsTF.py::WDSRConv3D (the class train.py / test.py build).

``WDSRConv3D(name, band, mean, std, maxShift).build(scale, numFilters, kernelSize, numResBlocks, expRate,
decayRate, numImgLR, patchSizeLR, isGrayScale)`` keeps the reference's names, argument order and meaning
(models/modelsTF.py:8-17) and returns a callable model: ``model(x, training=False)`` maps
float32 ``[N, P+maxShift, P+maxShift, T, 1]`` to ``[N, scale*P, scale*P, 1]`` and exposes
``trainable_variables`` in the reference checkpoint's order.  All arithmetic happens in
libprobav_hip.so (csrc/); torch only owns the device memory, the stream and the autograd edge: the model is ONE
torch custom op, `torch.ops.probav.wdsr_forward`, whose registered autograd formula is `torch.ops.probav.wdsr_backward`
(the engine's reverse pass) -- probav_amd/ops.py.
"""
import ctypes
import os
import weakref
from ctypes import c_char, c_int32, c_int64, c_void_p, byref

import torch

from . import _lib, ops                                 # noqa: F401  (ops registers torch.ops.probav.*)
from .arch import layer_table


class WDSRModel(torch.nn.Module):
    """What ``WDSRConv3D.build`` returns (the Keras ``Model`` of models/modelsTF.py:43)."""

    def __init__(self, name, band, mean, std, maxShift, scale, numFilters, numResBlocks, expRate, decayRate,
                 numImgLR, patchSizeLR, seed=None, inChannels=1):
        super().__init__()
        self.name, self.band = name, band
        self.mean, self.std, self.maxShift = float(mean), float(std), int(maxShift)
        self.scale, self.numFilters, self.numResBlocks = int(scale), int(numFilters), int(numResBlocks)
        self.expRate, self.decayRate = int(expRate), float(decayRate)
        self.numImgLR, self.patchSizeLR = int(numImgLR), int(patchSizeLR)
        self.inChannels = int(inChannels)                  # 1, or 3 for isGrayScale=False (models/modelsTF.py:19-20)
        self.arch = dict(numFilters=self.numFilters, numResBlocks=self.numResBlocks, expRate=self.expRate,
                         decayRate=self.decayRate, numImgLR=self.numImgLR, scale=self.scale, inChannels=self.inChannels)
        self.layers, total = layer_table(**self.arch)
        # state after the reference's first call: v ~ glorot_uniform, g = ||v||, bias = 0
        # (tensorflow_addons WeightNormalization with data_init=False)
        gen = torch.Generator().manual_seed(int(seed)) if seed is not None else None
        flat = torch.zeros(total, dtype=torch.float32)
        for L in self.layers:
            vs = L.vshape
            recept = 1
            for d in vs[:-2]:
                recept *= d
            limit = (6.0 / (recept * vs[-2] + recept * vs[-1])) ** 0.5
            v = (torch.rand(vs, generator=gen) * 2.0 - 1.0) * limit
            flat[L.v_off:L.b_off] = v.reshape(-1)
            flat[L.g_off:L.v_off] = v.double().reshape(-1, vs[-1]).pow(2).sum(0).sqrt().float()
        self.flat = torch.nn.Parameter(flat)
        self._engine = None
        self._ws = {}
        self._wcache, self._wcache_version = None, None

    # -- reference-facing surface ------------------------------------------------------------------
    @property
    def variable_names(self):
        """Keras-style names, `<layer>/g`, `<layer>/v`, `<layer>/bias`, checkpoint order."""
        return [n for L in self.layers for n in (L.name + "/g", L.name + "/v", L.name + "/bias")]

    @property
    def trainable_variables(self):
        """132 views into the flat buffer, ordered like the reference's model.trainable_variables."""
        out = []
        for L in self.layers:
            out += [self.flat[L.g_off:L.v_off], self.flat[L.v_off:L.b_off].view(L.vshape),
                    self.flat[L.b_off:L.b_off + L.cout]]
        return out

    def variable_gradients(self):
        """Gradients of the last backward pass, shaped and ordered like trainable_variables."""
        g = self.flat.grad
        if g is None:
            return None
        out = []
        for L in self.layers:
            out += [g[L.g_off:L.v_off], g[L.v_off:L.b_off].view(L.vshape), g[L.b_off:L.b_off + L.cout]]
        return out

    def load_variables(self, params):
        """params: {layer: {"g","v","bias"}} numpy/torch -> flat buffer (e.g. a converted checkpoint)."""
        with torch.no_grad():
            for L in self.layers:
                p = params[L.name]
                for key, lo, hi in (("g", L.g_off, L.v_off), ("v", L.v_off, L.b_off), ("bias", L.b_off, L.b_off + L.cout)):
                    t = torch.as_tensor(p[key], dtype=torch.float32).reshape(-1)
                    if t.numel() != hi - lo:
                        raise ValueError("%s/%s: expected %d values, got %d" % (L.name, key, hi - lo, t.numel()))
                    self.flat[lo:hi] = t.to(self.flat.device)
        self.invalidate_weight_cache()

    def _apply(self, fn, recurse=True):
        """`.to(device)` / `.float()` / ... replace the parameter's storage: whatever the cache holds was built from the old one."""
        out = super()._apply(fn, recurse)
        self.invalidate_weight_cache()
        return out

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_weight_cache()
        return out

    def forward(self, x, training=False):
        x = _lib.require_device(x, "model input")
        hin = self.patchSizeLR + self.maxShift
        if x.dim() != 5 or tuple(x.shape[1:]) != (hin, hin, self.numImgLR, self.inChannels):
            raise ValueError("model input must be [N, %d, %d, %d, %d] (models/modelsTF.py:19-20), got %s"
                             % (hin, hin, self.numImgLR, self.inChannels, tuple(x.shape)))
        if self.flat.device != x.device:
            raise RuntimeError("model parameters are on %s but the input is on %s" % (self.flat.device, x.device))
        x = x.contiguous().float()
        need_grad = bool(training) and torch.is_grad_enabled() and self.flat.requires_grad
        wc = self.weight_cache()
        if wc is None and not need_grad:
            # inference on weights nothing is updating (test.py:117 calls the model once per micro-batch of 16): normalise and pack them once
            wc = self.weight_cache_buffer()
            torch.ops.probav.weight_cache_build(self.flat, wc, int(self._handle().value))
            self.mark_weight_cache()
        y, ws = torch.ops.probav.wdsr_forward(self.flat, x, int(self._handle().value), self.scale * self.patchSizeLR, need_grad, wc)
        # the last call's workspace (saved activations), for introspection and the parity tests: a WEAK reference -- the autograd graph
        # owns a training workspace until its backward has run; a strong one here would keep the previous step's multi-GB block
        # resident beside the next one (PROBAV_KEEP_WS=1 pins it, e.g. to inspect an inference pass)
        self._ws = {(int(x.shape[0]), need_grad, self.flat.device): ws if os.environ.get("PROBAV_KEEP_WS") == "1" else weakref.ref(ws)}
        return y

    # -- engine plumbing ---------------------------------------------------------------------------
    def _handle(self):
        if self._engine is None:
            L = _lib.lib()
            cfg = _lib.NetCfg(self.scale, self.numFilters, self.numResBlocks, self.expRate,
                              int(self.numFilters * self.decayRate), self.numImgLR, self.patchSizeLR,
                              self.maxShift, self.mean, self.std, self.inChannels)
            h = c_void_p()
            _lib.check(L.probav_engine_create(byref(cfg), byref(h)), "probav_engine_create")
            if L.probav_param_count(h) != self.flat.numel():
                raise RuntimeError("host/native layer tables disagree: %d vs %d parameters"
                                   % (self.flat.numel(), L.probav_param_count(h)))
            self._engine = h
        return self._engine

    def native_layer_table(self):
        """[(name, g_off, v_off, b_off, shape)] as the native library lays the flat buffer out."""
        L, h, out = _lib.lib(), self._handle(), []
        for i in range(L.probav_num_layers(h)):
            name = (c_char * 32)()
            g, v, b = c_int64(), c_int64(), c_int64()
            shp = (c_int32 * 5)()
            _lib.check(L.probav_layer_info(h, i, byref(name), byref(g), byref(v), byref(b), byref(shp)))
            out.append((name.value.decode(), g.value, v.value, b.value, tuple(shp)))
        return out

    def set_impl(self, impl):
        """0 = generic direct kernels, 1 = fp32-MFMA row-tile kernels, 2 = fp32 MFMA + strip convolution,
        3 = 2 with the x6 kernels (fp32 products as six bf16-piece products on the bf16 MFMA pipe),
        4 (default) = the same kernels with the H3 arithmetic (three products of fp16 piece pairs, operands scaled by a power
        of two per sample / per filter column).  Every family computes a sample independently of its batch mates, bit for bit."""
        _lib.check(_lib.lib().probav_engine_set_impl(self._handle(), int(impl)), "probav_engine_set_impl")

    def set_side_stream_mode(self, mode):
        """probav_engine_side_stream: 0 = everything on the caller's stream; 1 = the slab sums, the residual path and the upscale layer's backward-filter on the
        engine's low-priority side stream; 2 = also the blocks' backward-filter kernels (the engine's default)."""
        _lib.check(_lib.lib().probav_engine_side_stream(self._handle(), int(mode)), "probav_engine_side_stream")

    def tune_side_stream(self, step, steps=8, rounds=2, modes=(2, 1)):
        """Which of the two side-stream modes is faster depends on the BOX: where the workgroups of a launch finish unevenly (the pool's slower boxes) the
        backward-filter kernels fill the gaps from the side stream (mode 2: -1.1 % there), elsewhere they are better off in the chain (mode 1: -1.2 %).
        Times `steps` calls of `step()` (one training step on the caller's current stream) per mode, `rounds` times alternating, keeps the faster mode and
        returns (mode, {mode: median ms per step}).  Both modes compute the same bits."""
        import statistics
        ms = {m: [] for m in modes}
        for _ in range(rounds):
            for m in modes:
                self.set_side_stream_mode(m)
                step()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                a.record()
                for _ in range(steps):
                    step()
                b.record()
                torch.cuda.synchronize()
                ms[m].append(a.elapsed_time(b) / steps)
        med = {m: statistics.median(v) for m, v in ms.items()}
        best = min(med, key=med.get)
        self.set_side_stream_mode(best)
        return best, med

    # -- weight cache -----------------------------------------------------------------
    def weight_cache_buffer(self):
        """The device buffer the fused optimizer step writes the next step's effective weights / operand fragments into."""
        if self._wcache is None or self._wcache.device != self.flat.device:
            n = _lib.lib().probav_weight_cache_bytes(self._handle())
            self._wcache = torch.empty((n + 3) // 4, dtype=torch.float32, device=self.flat.device)
            self._wcache_version = None
        return self._wcache

    def mark_weight_cache(self):
        """Called by the fused optimizer right after it has updated `flat` and filled the cache: valid until `flat` changes again."""
        self._wcache_version = (self.flat._version, self.flat.data_ptr())

    def invalidate_weight_cache(self):
        """Forget the cached effective weights / operand fragments: the next forward normalises and packs the parameters again.
        Called by `load_variables`, `load_state_dict` and `.to()`; CALL IT YOURSELF after any write to the parameters that torch's version
        counter of `flat` does not see -- `model.flat.data.copy_(...)` (`.data` carries a counter of its own), a raw-pointer write through
        the C ABI (`probav_nadam_step` via ctypes), `torch.distributed.broadcast(model.flat.data)`.  In-place writes to `flat` itself
        (`flat.copy_`, an optimizer's `add_`) are seen without help.  PROBAV_CHECK_WCACHE=1 makes every use of the cache verify it
        against a fresh build (one extra launch and a device comparison per forward: debugging only)."""
        self._wcache_version = None

    def weight_cache(self):
        """The cache if it still matches the parameters (any in-place change of `flat` bumps its version counter), else None."""
        if self._wcache is not None and self._wcache_version == (self.flat._version, self.flat.data_ptr()) \
                and self._wcache.device == self.flat.device:
            if os.environ.get("PROBAV_CHECK_WCACHE") == "1":
                fresh = torch.empty_like(self._wcache)
                torch.ops.probav.weight_cache_build(self.flat.detach(), fresh, int(self._handle().value))
                n = _lib.lib().probav_weff_count(self._handle())       # the cache's first block: the effective weights of all layers (the
                if not torch.equal(fresh[:n].view(torch.int32), self._wcache[:n].view(torch.int32)):      # blocks behind it have alignment gaps nobody writes)
                    raise RuntimeError("stale weight cache: the parameters were written behind torch's version counter "
                                       "(flat.data / raw pointer / broadcast) without model.invalidate_weight_cache()")
            return self._wcache
        return None

    def _workspace(self, batch, training):
        """The workspace the LAST forward call of this (batch, training) shape produced (every call gets its own: an output of
        torch.ops.probav.wdsr_forward, held by the autograd graph until its backward has run)."""
        ws = self._ws.get((int(batch), bool(training), self.flat.device))
        if ws is None:
            raise RuntimeError("no forward pass of batch %d (training=%s) has run on this model" % (batch, training))
        if isinstance(ws, weakref.ref):
            ws = ws()
            if ws is None:
                raise RuntimeError("the workspace of the last forward pass of batch %d (training=%s) has already been released (its backward "
                                   "has run, or nothing holds its output): set PROBAV_KEEP_WS=1 to keep it for inspection" % (batch, training))
        return ws

    def __del__(self):
        try:
            if self._engine is not None:
                _lib.lib().probav_engine_destroy(self._engine)
        except Exception:
            pass


class WDSRConv3D:
    """models/modelsTF.py:7-13."""

    def __init__(self, name, band, mean, std, maxShift):
        self.name = name
        self.band = band
        self.mean = mean
        self.std = std
        self.maxShift = maxShift

    def build(self, scale, numFilters, kernelSize, numResBlocks, expRate, decayRate, numImgLR, patchSizeLR,
              isGrayScale, seed=None):
        """models/modelsTF.py:15-43.  isGrayScale=False builds the reference's other input branch (:19-20): three input channels, seen by
        mainConv1 and residConv1 only; the output stays one channel.  kernelSize must be (3, 3, 3): it is the only value for which the
        reference's own graph closes -- its valid reducers take (k - 1) frames and pixels each, so with k = 5 the temporal axis of the
        9-frame graph runs 9 -> 5 -> 1 -> negative and Keras refuses to build it, and the residual path (scale valid k x k convolutions
        on a patch of P + maxShift, :45-53) only lands on P x P for k = 3."""
        ks = tuple(kernelSize) if not isinstance(kernelSize, int) else (kernelSize,) * 3
        if ks != (3, 3, 3):
            raise ValueError("kernelSize %r: the reference's graph (models/modelsTF.py:45-53, :152-164) only closes for 3x3x3" % (ks,))
        return WDSRModel(self.name, self.band, self.mean, self.std, self.maxShift, scale, numFilters,
                         numResBlocks, expRate, decayRate, numImgLR, patchSizeLR, seed=seed, inChannels=1 if isGrayScale else 3)

    def normalize(self, x):
        return (x - self.mean) / self.std

    def denormalize(self, x):
        return x * self.std + self.mean
