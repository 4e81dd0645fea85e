"""Host-side mirror of the reference's models/trainClass.py::ModelTrainer.

Same constructor, ``fitTrainData`` / ``trainStep`` / ``testStep`` / ``restore`` surface and the same
step / epoch / evaluation / checkpoint cadence (models/trainClass.py:25-143).  What TensorFlow provided
is restated on torch plumbing: tf.data shuffle/repeat/batch (utils/utils.py:32-39) as a host iterator,
tf.GradientTape as ``loss.backward()`` through the engine's custom Function, Keras ``Mean`` as running
sums kept on the device, tf.train.CheckpointManager(max_to_keep=5) as rotating ``ckpt-N.pt`` files with a
``checkpoint`` index, tf.summary scalars as JSON lines (tags 'Train PSNR', 'Train loss', 'Test loss',
'Test PSNR').  Under torch.distributed (one process per GPU, backend "nccl" = RCCL) the flat gradient
buffer is averaged with ONE all-reduce per step -- the data-parallel semantics of the reference's
unfinished MirroredStrategy experiment (debug/trainMultiGPU.py:65-68).
"""
import json
import logging
import os

import numpy as np
import torch

logging.basicConfig(format="%(asctime)s - %(message)s", level=logging.INFO)
logger = logging.getLogger("probav_amd")


class Mean:
    """Keras ``Mean`` metric: running mean of everything it is called with; state stays on the device
    so a training step never synchronises unless ``result()`` is read."""

    def __init__(self, name=""):
        self.name = name
        self.reset_states()

    def reset_states(self):
        self.total, self.count = None, 0

    def __call__(self, value):
        v = value.detach().double()
        s = v.sum()
        self.total = s if self.total is None else self.total + s
        self.count += v.numel()

    def result(self):
        return float(self.total) / self.count if self.count else 0.0

    def snapshot(self):
        """(running sum as a device scalar or None, count): what `result()` would divide, without reading the device."""
        return self.total, self.count


def shuffle_repeat_batch(n, epochs, batchSize, bufferSize, rng, repeat=True):
    """Index stream with tf.data semantics: from_tensor_slices -> shuffle(bufferSize, reshuffle each
    iteration) -> repeat(epochs) -> batch(batchSize) (utils/utils.py:32-34).  The shuffle buffer holds
    `bufferSize` pending elements and emits a uniformly chosen one; batches may straddle epochs because
    repeat() precedes batch(); the final partial batch is emitted (drop_remainder=False).
    One uniform draw per emitted element, drawn per epoch in bulk: the per-element cost is a few list operations (a generator with
    a numpy call per ELEMENT cost 3-4 ms per batch of 128 and capped the whole trainer at 70 steps/s)."""
    def epoch_order():
        u = rng.random(n).tolist()
        out = [0] * n
        m = min(int(bufferSize), n)
        buf = list(range(m))
        o = 0
        for i in range(m, n):                             # steady state: the buffer holds m + 1 elements when it emits
            buf.append(i)
            k = int(u[o] * (m + 1))
            out[o] = buf[k]
            buf[k] = buf[-1]
            buf.pop()
            o += 1
        while buf:                                        # the source is exhausted: drain
            k = int(u[o] * len(buf))
            out[o] = buf[k]
            buf[k] = buf[-1]
            buf.pop()
            o += 1
        return np.asarray(out, dtype=np.int64)

    pending = np.empty(0, dtype=np.int64)
    for _ in range(epochs if repeat else 1):
        pending = np.concatenate([pending, epoch_order()])
        nb = len(pending) // batchSize
        for b in range(nb):
            yield pending[b * batchSize:(b + 1) * batchSize]
        pending = pending[nb * batchSize:]
    if len(pending):
        yield pending


class BatchPrefetcher:
    """The `.prefetch(AUTOTUNE)` of the reference's tf.data pipeline (utils/utils.py:32-39) for a HIP device: a background thread
    gathers the next batches from the host arrays into PINNED staging buffers (`depth` rotating slots) and enqueues their
    host-to-device copies on a side stream; the consumer's stream only waits on the copy's event.  Iterating yields tuples of
    device tensors in exactly the order of `index_batches`.  On a CPU device it degrades to a plain gather (host-logic tests)."""

    def __init__(self, arrays, dtypes, index_batches, device, depth=2):
        import queue
        import threading
        self.arrays, self.dtypes, self.device, self.depth = arrays, dtypes, torch.device(device), max(1, int(depth))
        self._it = iter(index_batches)
        self._q = queue.Queue(maxsize=self.depth)
        self._cuda = self.device.type == "cuda"
        self._stream = torch.cuda.Stream(device=self.device) if self._cuda else None
        self._slots = [None] * (self.depth + 1)               # one more slot than the queue holds: a slot is rewritten only after
        self._thread = threading.Thread(target=self._work, daemon=True)     # its batch was handed out AND the next one produced
        self._thread.start()

    def _stage(self, k, idx):
        outs = []
        if self._slots[k] is None or self._slots[k][0].shape[0] < len(idx):
            self._slots[k] = [torch.empty((len(idx),) + tuple(a.shape[1:]), dtype=dt, pin_memory=self._cuda)
                              for a, dt in zip(self.arrays, self.dtypes)]
        for a, dt, pin in zip(self.arrays, self.dtypes, self._slots[k]):
            view = pin[:len(idx)]
            dst = view.numpy()
            if isinstance(a, np.ndarray) and a.dtype == dst.dtype:
                np.take(a, idx, axis=0, out=dst)                                    # host gather straight into the pinned slot
            else:
                view.copy_(torch.as_tensor(np.ascontiguousarray(a[idx])).to(dt))   # (other containers / dtypes: gather, cast, copy)
            outs.append(view)
        return outs

    def _work(self):
        try:
            k = 0
            for idx in self._it:
                host = self._stage(k, idx)
                if self._cuda:
                    with torch.cuda.stream(self._stream):
                        dev = [h.to(self.device, non_blocking=True) for h in host]
                        ev = torch.cuda.Event()
                        ev.record(self._stream)
                    ev.synchronize()                          # the pinned slot may be rewritten once its copy has left the host
                    self._q.put((dev, ev))
                else:
                    self._q.put(([h.clone() for h in host], None))
                k = (k + 1) % len(self._slots)
            self._q.put(None)
        except BaseException as exc:                          # surface worker failures in the consumer
            self._q.put(exc)

    def __iter__(self):
        return self

    def __next__(self):
        item = self._q.get()
        if item is None:
            raise StopIteration
        if isinstance(item, BaseException):
            raise item
        dev, ev = item
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
            for t in dev:
                t.record_stream(torch.cuda.current_stream(self.device))
        return tuple(dev)


def dp_state():
    """(collectives active?, world size).  Active when a process group exists and either spans more than one rank or
    PROBAV_FORCE_DP=1 is set: a world-size-1 RCCL group then really executes every collective of the data-parallel step (RCCL init
    with a device id, the stream hand-over between the engine's side-stream join, torch's launch stream and the NCCL stream, the
    private workspace pool beside RCCL's own buffers) on the one GPU a box has; with one rank a mean over ranks changes no bit."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False, 1
    world = dist.get_world_size()
    return (world > 1 or os.environ.get("PROBAV_FORCE_DP", "0") == "1"), world


def allreduce_mean_(flat_grad):
    """Average the flat gradient buffer over the data-parallel ranks: one collective per step
    (2.14 MB for p16t9c85r12).  No-op when torch.distributed is not initialised."""
    import torch.distributed as dist
    active, world = dp_state()
    if active:
        flat_grad.div_(world)
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad


def allreduce_metrics_(loss, metric):
    """the scalar reduce of the reference's multi-GPU trainer (debug/trainClassMultiGPU0.py:162-178:
    `strategy.reduce(MEAN, ...)` of the per-replica loss and metric): ONE 2-float all-reduce per evaluation step (a training step
    carries the pair in the tail of its gradient bucket instead: GradBucket).  Returns (mean loss, mean metric) as 0-d tensors;
    without torch.distributed it only takes the local means."""
    import torch.distributed as dist
    pair = torch.stack([loss.detach().double().mean(), metric.detach().double().mean()])
    active, world = dp_state()
    if active:
        dist.all_reduce(pair, op=dist.ReduceOp.SUM)
        pair = pair / world
    return pair[0], pair[1]


class GradBucket:
    """The ONE collective of a data-parallel training step (C1 + C2 in one message): a persistent fp32 buffer
    [n + 2] = the flat gradient (for the engine's model: one tensor of 535 267 floats) followed by the step's two scalars, the
    replica's mean loss and mean metric -- what debug/trainClassMultiGPU0.py:162-178 sends as a gradient all-reduce plus two
    `strategy.reduce(MEAN)` calls.  `reduce_` pre-scales by 1 / world, all-reduces (SUM) once, re-points every `p.grad` at its
    slice of the bucket (no copy back) and returns the two global means.  At 8 ranks a latency-bound RCCL call costs more than the
    2 MB of wire time, so the pair rides with the gradient instead of taking a launch of its own.
    A deliberate deviation from the reference's replica context: there `optimizer.apply_gradients` SUM-aggregates the gradients of
    the per-replica MEAN losses (debug/trainClassMultiGPU0.py:153: `computeLoss` with the global batch size is defined and not used), so
    its effective gradient is `world` times the global-batch mean; here the update is that of ONE batch of world x 128 patches, the
    mean, so that the learning rate means the same at every world size.  With Nadam / Adam the factor only moves
    the update through epsilon.  The two scalars pass through fp32 in the bucket's tail (they are logged, not trained on)."""

    def __init__(self):
        self.buf = None

    def reduce_(self, params, loss, metric):
        import torch.distributed as dist
        active, world = dp_state()
        ps = [p for p in params if p.grad is not None]
        n = sum(p.grad.numel() for p in ps)
        dev = ps[0].grad.device
        if self.buf is None or self.buf.numel() != n + 2 or self.buf.device != dev:
            self.buf = torch.empty(n + 2, dtype=torch.float32, device=dev)
        o = 0
        for p in ps:
            k = p.grad.numel()
            self.buf[o:o + k].copy_(p.grad.reshape(-1))
            o += k
        self.buf[n:].copy_(torch.stack([loss.detach().double().mean(), metric.detach().double().mean()]))
        if active:
            if world > 1:
                self.buf.div_(world)
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM)
        o = 0
        for p in ps:
            k = p.grad.numel()
            p.grad = self.buf[o:o + k].view(p.shape)
            o += k
        tail = self.buf[n:].double()                   # (a copy: the bucket is rewritten by the next step)
        return tail[0], tail[1]


class _LateScalars:
    """Per-step log lines and summary scalars WITHOUT a host sync per step: the running sums of the two train metrics are copied to a
    pinned host buffer on the compute stream (non-blocking) with an event behind them, and the line of step k is emitted once that
    event has completed -- normally while step k+1 is being enqueued.  Same text, same values, same order as the reference's
    per-step `logger.info` / `tf.summary.scalar` (models/trainClass.py:104-108); `flush()` drains what is pending."""

    def __init__(self, emit, device, depth=2):
        self.emit, self.depth, self.pending = emit, depth, []
        self.cuda = torch.device(device).type == "cuda"

    def push(self, means, meta):
        sums = [m.snapshot() for m in means]
        vals = torch.stack([(t if t is not None else torch.zeros((), dtype=torch.float64)).to(torch.float64) for t, _ in sums])
        counts = [c for _, c in sums]
        if self.cuda:
            host = torch.empty(vals.shape, dtype=torch.float64, pin_memory=True)
            host.copy_(vals, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            host, ev = vals.clone(), None
        self.pending.append((host, ev, counts, meta))
        while self.pending and (len(self.pending) > self.depth or self.pending[0][1] is None or self.pending[0][1].query()):
            self._pop()

    def _pop(self):
        host, ev, counts, meta = self.pending.pop(0)
        if ev is not None:
            ev.synchronize()
        self.emit([float(h) / c if c else 0.0 for h, c in zip(host, counts)], meta)

    def flush(self):
        while self.pending:
            self._pop()


def load_nadam_state(optimizer, step, momentum_cache, m, v):
    """Put Keras-Nadam state (iteration count, running momentum product, first / second moments as flat arrays in the engine's
    parameter order) into a HipNadam or torch.optim.NAdam over the model's single flat parameter."""
    (p,) = [q for g in optimizer.param_groups for q in g["params"]]
    mt = torch.as_tensor(m, dtype=torch.float32).reshape(p.shape).to(p.device)
    vt = torch.as_tensor(v, dtype=torch.float32).reshape(p.shape).to(p.device)
    if isinstance(optimizer, HipNadam):
        optimizer.state[p] = {"step": int(step), "momentum_cache": float(momentum_cache), "m": mt.clone(), "v": vt.clone()}
    elif isinstance(optimizer, torch.optim.NAdam):
        optimizer.state[p] = {"step": torch.tensor(float(step)), "mu_product": torch.tensor(float(momentum_cache)),
                              "exp_avg": mt.clone(), "exp_avg_sq": vt.clone()}
    else:
        raise TypeError("the reference checkpoint carries Nadam slots; optimizer is %s" % type(optimizer).__name__)


def _fused_update(model, p, g, st, lr, b1, b2, eps, c_g, c_m, c_v):
    """One optimizer update of the flat parameter `p`: with a WDSRModel behind it (make_optimizer passes it) the update is fused with the
    weight normalisation and operand packing of the NEXT forward pass (torch.ops.probav.optimizer_wn_step);
    otherwise the plain fused element-wise launch."""
    if model is not None and p is model.flat:
        torch.ops.probav.optimizer_wn_step(p, g, st["m"], st["v"], model.weight_cache_buffer(), int(model._handle().value), lr, b1, b2, eps, c_g, c_m, c_v)
        model.mark_weight_cache()
    else:
        torch.ops.probav.nadam_step(p, g, st["m"], st["v"], lr, b1, b2, eps, c_g, c_m, c_v)


class HipNadam(torch.optim.Optimizer):
    """Keras ``Nadam`` (optimizer_v2 defaults: beta_1 0.9, beta_2 0.999, epsilon 1e-7, schedule_decay 0.004)
    as ONE fused HIP launch per parameter tensor (the model has a single flat one).  The momentum schedule
    (mu_t, the running product Pi_t = `momentum_cache`) is tracked on the host in double, like Keras tracks it in
    variables; `state_dict()` carries step, momentum cache and the two slots, so checkpoints resume exactly."""

    def __init__(self, params, lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, schedule_decay=0.004, model=None):
        super().__init__(params, dict(lr=lr, beta_1=beta_1, beta_2=beta_2, epsilon=epsilon, schedule_decay=schedule_decay))
        self.model = model

    @torch.no_grad()
    def step(self, closure=None):
        from . import _lib, ops                          # noqa: F401  (ops registers torch.ops.probav.nadam_step)
        for group in self.param_groups:
            b1, b2 = group["beta_1"], group["beta_2"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                _lib.require_device(p, "parameter")
                st = self.state[p]
                if not st:
                    st["step"], st["momentum_cache"] = 0, 1.0
                    st["m"], st["v"] = torch.zeros_like(p), torch.zeros_like(p)
                st["step"] += 1
                t = st["step"]
                mu_t = b1 * (1.0 - 0.5 * 0.96 ** (t * group["schedule_decay"]))
                mu_t1 = b1 * (1.0 - 0.5 * 0.96 ** ((t + 1) * group["schedule_decay"]))
                pi_t = st["momentum_cache"] * mu_t
                st["momentum_cache"] = pi_t
                c_g = (1.0 - mu_t) / (1.0 - pi_t)
                c_m = mu_t1 / (1.0 - pi_t * mu_t1)
                c_v = 1.0 / (1.0 - b2 ** t)
                g = p.grad.contiguous()
                _fused_update(self.model, p, g, st, group["lr"], b1, b2, group["epsilon"], c_g, c_m, c_v)


class HipAdam(torch.optim.Optimizer):
    """Keras ``Adam`` (optimizer_v2 defaults, amsgrad off): lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); theta -= lr_t m / (sqrt(v) + eps), eps
    outside the bias correction.  Same fused launch as HipNadam (the kernel computes
    theta -= lr (c_g g + c_m m) / (sqrt(c_v v) + eps); here c_g = 0, c_m = sqrt(1 - b2^t) / (1 - b1^t), c_v = 1)."""

    def __init__(self, params, lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, model=None):
        super().__init__(params, dict(lr=lr, beta_1=beta_1, beta_2=beta_2, epsilon=epsilon))
        self.model = model

    @torch.no_grad()
    def step(self, closure=None):
        from . import _lib, ops                          # noqa: F401  (ops registers torch.ops.probav.nadam_step)
        for group in self.param_groups:
            b1, b2 = group["beta_1"], group["beta_2"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                _lib.require_device(p, "parameter")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["m"], st["v"] = torch.zeros_like(p), torch.zeros_like(p)
                st["step"] += 1
                t = st["step"]
                c_m = (1.0 - b2 ** t) ** 0.5 / (1.0 - b1 ** t)
                _fused_update(self.model, p, p.grad.contiguous(), st, group["lr"], b1, b2, group["epsilon"], 0.0, c_m, 1.0)


class HipSGD(torch.optim.Optimizer):
    """Keras ``SGD`` without momentum through the same fused launch (c_g = 1, c_m = 0, c_v = 0, eps = 1: theta -= lr g)."""

    def __init__(self, params, lr=1e-2, model=None):
        super().__init__(params, dict(lr=lr))
        self.model = model

    @torch.no_grad()
    def step(self, closure=None):
        from . import _lib, ops                          # noqa: F401  (ops registers torch.ops.probav.nadam_step)
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                _lib.require_device(p, "parameter")
                st = self.state[p]
                if not st:
                    st["m"], st["v"] = torch.zeros_like(p), torch.zeros_like(p)
                _fused_update(self.model, p, p.grad.contiguous(), st, group["lr"], 0.0, 0.0, 1.0, 1.0, 0.0, 0.0)


def make_optimizer(name, model, learning_rate):
    """train.py:77-83: 'adam' -> Keras Adam, 'nadam' -> Keras Nadam, anything else -> SGD, with the Keras
    defaults restated (epsilon 1e-7; Nadam schedule_decay 0.004 -- SURVEY.md A.5), each ONE fused HIP launch over the model's
    flat parameter buffer.  There is no CPU implementation: `step()` on parameters that do not live on a HIP device raises."""
    params = list(model.parameters())
    fused = model if hasattr(model, "weight_cache_buffer") else None
    if name == "adam":
        return HipAdam(params, lr=learning_rate, model=fused)
    if name == "nadam":
        return HipNadam(params, lr=learning_rate, model=fused)
    return HipSGD(params, lr=learning_rate, model=fused)


class _SideStreamTuner:
    """The first training steps pick the engine's side-stream mode (probav_amd/modelsTF.py: WDSRModel.tune_side_stream has the why): windows of 8 steps, modes 2, 1, 2, 1,
    timed by events on the training stream; the faster mode stays.  The steps are ordinary training steps -- both modes compute the same bits."""
    ORDER, WINDOW, DEFAULT, NOISE = (2, 1, 2, 1), 8, 2, 0.004

    def __init__(self, model):
        self.model, self.k, self.ms, self.ev, self.mode = model, 0, {1: [], 2: []}, None, 2
        self.voided = False
        flat = getattr(model, "flat", None)
        self.ok = hasattr(model, "set_side_stream_mode") and flat is not None and flat.is_cuda

    def tick(self):
        """Call at the top of every step; True once the choice is made."""
        if not self.ok:
            return True
        w, pos = divmod(self.k, self.WINDOW + 1)                    # one untimed step at the head of every window
        self.k += 1
        if pos == 0:
            if self.ev is not None:
                self.ev[1].record()
                self.ev[1].synchronize()
                if not self.voided:
                    self.ms[self.mode].append(self.ev[0].elapsed_time(self.ev[1]) / self.WINDOW)
                self.ev, self.voided = None, False
            if w == len(self.ORDER):
                best = self.choose()
                self.model.set_side_stream_mode(best)
                logger.info("[ INFO ] engine side-stream mode %d (ms per step by mode: %s)", best, {m: [round(v, 3) for v in vs] for m, vs in self.ms.items()})
                return True
            self.mode = self.ORDER[w]
            self.model.set_side_stream_mode(self.mode)
        elif pos == 1:
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record()
        return False

    def void(self):
        """The current window contains something that is not a training step (evaluation, checkpoint): drop it."""
        if self.ev is not None:
            self.voided = True

    def choose(self):
        """The faster mode by the windows' medians; the default when a mode has no valid window or the two are within NOISE of each other.
        Data parallel: every rank takes rank 0's choice (a rank's windows include its wait for the others in the all-reduce; ranks in
        different modes would compute the same bits at different speeds)."""
        med = {m: (sorted(v)[len(v) // 2] if v else None) for m, v in self.ms.items()}
        best = self.DEFAULT
        if all(v is not None for v in med.values()):
            cand = min(med, key=med.get)
            other = max(med, key=med.get)
            if cand != self.DEFAULT and med[other] - med[cand] <= self.NOISE * med[other]:
                logger.info("[ INFO ] side-stream modes within %.1f %% of each other: the default stays", 100 * self.NOISE)
            else:
                best = cand
        active, world = dp_state()
        if active and world > 1:
            import torch.distributed as dist
            t = torch.tensor([best], dtype=torch.int32, device=self.model.flat.device)
            dist.broadcast(t, src=0)
            best = int(t.item())
        return best


class ModelTrainer:
    """models/trainClass.py:17-143."""

    def __init__(self, model, loss, metric, optimizer, ckptDir, logDir, multiGPU=True, evalStep=1000):
        os.makedirs(ckptDir, exist_ok=True)
        os.makedirs(logDir, exist_ok=True)
        self._model, self.optimizer = model, optimizer
        self.loss, self.metric = loss, metric
        self.ckptDir, self.logDir = ckptDir, logDir
        self.step, self.psnr, self.save_counter = 0, 1.0, 0       # tf.train.Checkpoint(step, psnr, ...) (:33-36)
        self.max_to_keep = 5                                       # CheckpointManager(max_to_keep=5)   (:37-39)
        self.trainLoss, self.trainPSNR = Mean("trainLoss"), Mean("trainPSNR")
        self.testLoss, self.testPSNR = Mean("testLoss"), Mean("testPSNR")
        self.evalStep = evalStep
        self.multiGPU = multiGPU
        self.strategy = None
        self._log = None
        self._bucket = GradBucket()
        self.restore()

    @property
    def model(self):
        return self._model

    # -- checkpointing -------------------------------------------------------------------------------
    def _index_path(self):
        """Index of the .pt checkpoints this trainer writes.  NOT `checkpoint`: that name is TensorFlow's CheckpointState file, and a
        directory written by the reference must stay readable by the reference after this trainer has saved into it."""
        return os.path.join(self.ckptDir, "checkpoint.pt-index")

    def _tf_state_path(self):
        return os.path.join(self.ckptDir, "checkpoint")

    def _read_index(self):
        if os.path.exists(self._index_path()):
            with open(self._index_path()) as fh:
                return [ln.strip() for ln in fh if ln.strip()]
        # no index of this revision.  A directory written by an earlier revision kept its index in `checkpoint` (lines `ckpt-N.pt`);
        # failing that, whatever ckpt-N.pt files exist, by N: never restart at ckpt-1.pt over existing files
        legacy = []
        if os.path.exists(self._tf_state_path()):
            with open(self._tf_state_path()) as fh:
                legacy = [ln.strip() for ln in fh if ln.strip().endswith(".pt")]
            legacy = [n for n in legacy if os.path.exists(os.path.join(self.ckptDir, n))]
        if not legacy:
            import glob
            import re
            found = [os.path.basename(f) for f in glob.glob(os.path.join(self.ckptDir, "ckpt-*.pt"))]
            found = [n for n in found if re.fullmatch(r"ckpt-\d+\.pt", n)]
            legacy = sorted(found, key=lambda n: int(n[5:-3]))
            if legacy:
                logger.warning("[ WARN ] %s holds %d .pt checkpoints but no index (checkpoint.pt-index): taking them in numeric order",
                               self.ckptDir, len(legacy))
        return legacy

    @property
    def latest_checkpoint(self):
        names = self._read_index()
        return os.path.join(self.ckptDir, names[-1]) if names else None

    def _tf_latest(self):
        """Prefix of the latest TensorFlow-format checkpoint if `ckptDir/checkpoint` is a TF CheckpointState text file
        (`model_checkpoint_path: "ckpt-124"`), i.e. a directory written by the reference itself."""
        if not os.path.exists(self._tf_state_path()):
            return None
        with open(self._tf_state_path()) as fh:
            first = fh.readline().strip()
        if not first.startswith("model_checkpoint_path:"):
            return None
        return os.path.join(self.ckptDir, first.split(":", 1)[1].strip().strip('"'))

    def restore(self):
        path = self.latest_checkpoint                   # checkpoints of this trainer are newer than a TF bundle in the same directory
        tf_prefix = self._tf_latest() if not (path and os.path.exists(path)) else None
        if tf_prefix is not None:                       # weights trained by the reference (tf.train.Checkpoint bundle)
            from .tfckpt import load_reference_checkpoint, load_reference_optimizer
            step = load_reference_checkpoint(self._model, tf_prefix)
            self.step = int(step or 0)
            slots = load_reference_optimizer(self._model, tf_prefix)
            if slots is not None:
                self.psnr = slots["psnr"]
            if slots is not None and isinstance(self.optimizer, (HipNadam, torch.optim.NAdam)):
                load_nadam_state(self.optimizer, slots["iter"], slots["momentum_cache"], slots["m"], slots["v"])
                print(f"[ INFO ] Model and Nadam state (iteration {slots['iter']}) restored from TensorFlow checkpoint {tf_prefix} at step {self.step}.")
            else:
                if self.optimizer is not None:
                    logger.warning("[ WARN ] %s: optimizer state NOT restored (%s); the optimizer restarts from zero moments",
                                   tf_prefix, "the bundle has no optimizer slots" if slots is None else "optimizer is not Nadam")
                print(f"[ INFO ] Model restored from TensorFlow checkpoint {tf_prefix} at step {self.step}.")
            return
        if path and os.path.exists(path):
            state = torch.load(path, map_location="cpu")
            self._model.load_variables(state["model"])
            if self.optimizer is not None and state.get("optimizer") is not None:
                self.optimizer.load_state_dict(state["optimizer"])
            self.step, self.psnr = int(state["step"]), float(state["psnr"])
            self.save_counter = max(int(state.get("save_counter", 0)), int(os.path.basename(path)[5:-3]))
            print(f"[ INFO ] Model restored from checkpoint at step {self.step}.")

    def save(self):
        if self._rank() != 0:
            return None
        self.save_counter += 1
        name = "ckpt-%d.pt" % self.save_counter
        before = [n for n in self._read_index() if n != name]
        names = self._model.variable_names
        tensors = [t.detach().cpu() for t in self._model.trainable_variables]
        model_state = {}
        for n, t in zip(names, tensors):
            layer, key = n.split("/")
            model_state.setdefault(layer, {})[key] = t
        torch.save({"model": model_state, "optimizer": self.optimizer.state_dict() if self.optimizer else None,
                    "step": self.step, "psnr": self.psnr, "save_counter": self.save_counter},
                   os.path.join(self.ckptDir, name))
        kept = before + [name]
        for old in kept[:-self.max_to_keep]:
            try:
                os.remove(os.path.join(self.ckptDir, old))
            except OSError:
                pass
        with open(self._index_path(), "w") as fh:
            fh.write("\n".join(kept[-self.max_to_keep:]) + "\n")
        return name

    # -- summaries ---------------------------------------------------------------------------------
    def _scalar(self, tag, value, step):
        if self._rank() != 0:
            return
        if self._log is None:
            self._log = open(os.path.join(self.logDir, "events.jsonl"), "a")
        self._log.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")

    @staticmethod
    def _rank():
        import torch.distributed as dist
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0

    @staticmethod
    def _world():
        import torch.distributed as dist
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def _device(self):
        return self._model.flat.device

    def _to_dev(self, a, dtype=None):
        t = torch.as_tensor(np.ascontiguousarray(a))
        return t.to(device=self._device(), dtype=dtype, non_blocking=True) if dtype else t.to(self._device(), non_blocking=True)

    # -- training loop (models/trainClass.py:61-122) ---------------------------------------------------
    def fitTrainData(self, X, y, globalBatchSize, epochs, valData, bufferSize=256, valSteps=64,
                     saveBestOnly=True, initEpoch=0, seed=0):
        logger.info("[ INFO ] Loading data set to buffer cache...")
        yHR, yMask = y[0], y[1]
        rank, world = self._rank(), self._world()
        if world > 1 and self.multiGPU:
            # per-replica batch = cfg batch (debug/trainClassMultiGPU0.py:67-73).  Every rank gets EXACTLY len(X) // world samples
            # (the tail of an indivisible data set is dropped): equal shard lengths give every rank the same number of batches, so the
            # per-step gradient all-reduce never waits for a rank that has already finished.
            per = len(X) // world
            X, yHR, yMask = X[rank::world][:per], yHR[rank::world][:per], yMask[rank::world][:per]
        rng = np.random.default_rng(seed + rank)                 # training order: consumed by the prefetch thread only
        vrng = np.random.default_rng(seed + 7919)                # validation order: its own stream, the same on every rank
        dataSetLength = len(X)
        totalSteps = int(dataSetLength / globalBatchSize)           # tf.cast(len/batch, int64) truncates (:75)
        if totalSteps < 1:
            raise ValueError("data set (%d) smaller than one batch (%d)" % (dataSetLength, globalBatchSize))
        globalStep = self.step
        step = globalStep % totalSteps
        epoch = initEpoch
        logger.info("[ INFO ] Begin training...")
        mask_dtype = torch.as_tensor(np.asarray(yMask[:1])).dtype
        batches = BatchPrefetcher((X, yHR, yMask), (torch.float32, torch.float32, mask_dtype),
                                  shuffle_repeat_batch(dataSetLength, epochs, globalBatchSize, bufferSize, rng), self._device())

        def emit(vals, meta):
            ep, st, gs = meta
            logger.info(f"[ EPOCH {ep}/{epochs} ] - [ STEP {st}/{int(totalSteps)} ] Loss: {vals[0]:.6f}, cPSNR: {vals[1]:.3f}")
            self._scalar("Train PSNR", vals[1], gs)
            self._scalar("Train loss", vals[0], gs)
        late = _LateScalars(emit, self._device())
        tuner = _SideStreamTuner(self.model) if getattr(self, "tune_side_stream", True) else None       # (set trainer.tune_side_stream = False to keep the engine's mode)
        # The host is at most one step (134 launches) ahead of the device: a generation-2 sweep of the interpreter's collector is a 10-20 ms hole in the launch
        # stream.  The loop allocates no reference cycles of its own: the automatic collector is OFF inside it (as in bench.py's timed region) and runs by
        # hand at the evaluation points and every 2000 steps; the caller's setting comes back in the `finally` below.
        import gc
        gc.collect()
        gc_was = gc.isenabled()
        gc.disable()
        try:
            self._fit_loop(batches, late, tuner, gc, step, epoch, epochs, globalStep, totalSteps, valData, globalBatchSize, bufferSize, vrng, valSteps, saveBestOnly)
        finally:
            if gc_was:
                gc.enable()
        late.flush()
        if self._log is not None:
            self._log.flush()

    def _fit_loop(self, batches, late, tuner, gc, step, epoch, epochs, globalStep, totalSteps, valData, globalBatchSize, bufferSize, vrng, valSteps, saveBestOnly):
        gc_every = 2000
        for xb, hb, mb in batches:
            if gc_every and self.step % gc_every == gc_every - 1:
                gc.collect()
            if tuner is not None and tuner.tick():
                tuner = None
            if (totalSteps - step) == 0:
                epoch += 1
                step = self.step % totalSteps
                late.flush()
                logger.info(f"[ ***************  NEW EPOCH  *************** ] Epoch number {epoch}")
                for m in (self.trainLoss, self.trainPSNR, self.testLoss, self.testPSNR):
                    m.reset_states()
            step += 1
            globalStep += 1
            self.trainStep(xb, hb, mb)
            self.step += 1
            late.push((self.trainLoss, self.trainPSNR), (epoch, step, globalStep))      # the reference's per-step line, one step late, no sync

            if step != 0 and (step % self.evalStep) == 0:
                late.flush()
                if tuner is not None:
                    tuner.void()                        # an evaluation / checkpoint inside a timing window: the window does not count
                gc.collect()
                self.testLoss.reset_states()
                self.testPSNR.reset_states()
                for k, vidx in enumerate(shuffle_repeat_batch(len(valData[0]), 1, globalBatchSize, bufferSize, vrng, repeat=False)):
                    if k >= valSteps:                   # .take(valSteps) (utils/utils.py:37-39)
                        break
                    self.testStep(self._to_dev(valData[0][vidx], torch.float32), self._to_dev(valData[1][vidx], torch.float32),
                                  self._to_dev(valData[2][vidx]))
                testLoss, testPSNR = self.testLoss.result(), self.testPSNR.result()
                self._scalar("Test loss", testLoss, globalStep)
                self._scalar("Test PSNR", testPSNR, globalStep)
                logger.info(f"[ *************** VAL INFO *************** ] Validation Loss: {testLoss:.6f}, Validation PSNR: {testPSNR:.3f}")
                if self._log is not None:
                    self._log.flush()
                if saveBestOnly and (testPSNR <= self.psnr):
                    continue
                logger.info("[ SAVE ] Saving checkpoint...")
                self.psnr = testPSNR
                self.save()

    # -- one step (models/trainClass.py:124-143) -----------------------------------------------------------
    def _dp(self):
        return self.multiGPU and dp_state()[0]

    def trainStep(self, patchLR, patchHR, maskHR):
        predPatchHR = self._model(patchLR, training=True)
        loss = self.loss(patchHR, maskHR, predPatchHR)             # Loss(patchHR, maskHR, predPatchHR)
        self.optimizer.zero_grad(set_to_none=True)
        # tape.gradient(loss, trainable_variables).  The seed d loss / d loss = 1 is a cached tensor: `loss.backward()` launches a fill kernel for it every step
        seed = getattr(self, "_grad_seed", None)
        if seed is None or seed.device != loss.device or seed.dtype != loss.dtype or seed.shape != loss.shape:
            seed = self._grad_seed = torch.ones_like(loss)
        loss.backward(seed)
        metric = self.metric(patchHR, maskHR, predPatchHR.detach())  # (does not depend on the update: evaluated before the exchange)
        if self._dp():
            # C1 + C2 as ONE collective: the flat gradient with the replica's loss / metric means in its tail
            # (debug/trainClassMultiGPU0.py:153 gradient all-reduce inside apply_gradients, :162-178 strategy.reduce(MEAN))
            loss, metric = self._bucket.reduce_(list(self._model.parameters()), loss, metric)
        self.optimizer.step()                                      # optimizer.apply_gradients
        self.trainLoss(loss)
        self.trainPSNR(metric)

    def testStep(self, patchLR, patchHR, maskHR):
        with torch.no_grad():
            predPatchHR = self._model(patchLR, training=False)
            loss = self.loss(patchHR, maskHR, predPatchHR)
            metric = self.metric(patchHR, maskHR, predPatchHR)
        if self._dp():
            loss, metric = allreduce_metrics_(loss, metric)
        self.testLoss(loss)
        self.testPSNR(metric)
