"""The torch custom ops of probav_amd/ops.py (the north-star boundary: "hand-written HIP kernels exposed as torch custom ops"):
registered schemas, fake-tensor rules and autograd formulas checked with torch.library.opcheck, and the ops used directly -- without
the WDSRModel / Losses wrappers -- give the same numbers as through them."""
import ctypes

import numpy as np
import pytest
import torch

from probav_amd import synth

pytestmark = pytest.mark.gpu


def _model(dev, T=9):
    from probav_amd.modelsTF import WDSRConv3D
    m = WDSRConv3D("t", "NIR", synth.NIR_MEAN, synth.NIR_STD, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, T, 16, True, seed=0)
    m.load_variables(synth.synth_params(seed=61, perturb=True, numImgLR=T))
    return m.to(dev)


def test_ops_are_registered_with_schemas():
    import probav_amd.ops  # noqa: F401
    want = {"wdsr_forward": "(Tensor flat, Tensor x, SymInt engine, SymInt out_size, bool training, Tensor? wcache=None) -> (Tensor, Tensor)",
            "wdsr_backward": "(Tensor flat, Tensor dy, Tensor ws, SymInt engine, Tensor? wcache=None) -> Tensor",      # functional: the saved state is only read
            "nadam_step": None, "optimizer_wn_step": None, "shift_loss": None, "shift_loss_backward": None, "shift_metrics": None, "clip_round": None}
    for name, schema in want.items():
        op = getattr(torch.ops.probav, name).default
        if schema is not None:
            assert str(op._schema).endswith(schema), str(op._schema)


def test_opcheck_network_ops(dev):
    m = _model(dev)
    x = torch.as_tensor(synth.synth_batch(2, seed=62)[0]).to(dev)
    eng = int(m._handle().value)
    flat = m.flat.detach().clone().requires_grad_(True)
    # the workspace output holds uninitialised scratch beyond the saved activations, so the output-comparing AOT test is left out;
    # schema, fake-tensor rule and autograd registration are checked
    tests = ("test_schema", "test_autograd_registration", "test_faketensor")
    torch.library.opcheck(torch.ops.probav.wdsr_forward.default, (flat, x, eng, 48, True), test_utils=tests)
    torch.library.opcheck(torch.ops.probav.wdsr_forward.default, (flat.detach(), x, eng, 48, False), test_utils=tests)
    y, ws = torch.ops.probav.wdsr_forward(flat, x, eng, 48, True)
    dy = torch.randn_like(y)
    torch.library.opcheck(torch.ops.probav.wdsr_backward.default, (flat.detach(), dy, ws.clone(), eng), test_utils=("test_schema", "test_faketensor"))
    # the op used directly == the model wrapper, forward and gradient, bit for bit
    (y * dy).sum().backward()
    m.flat.grad = None
    y2 = m(x, training=True)
    assert torch.equal(y2, y)
    (y2 * dy).sum().backward()
    assert torch.equal(m.flat.grad, flat.grad)


def test_opcheck_loss_optimizer_and_epilogue_ops(dev):
    rng = np.random.default_rng(3)
    _, hr, mask = synth.synth_batch(3, seed=63)
    pred = (hr + rng.normal(0, 200, hr.shape)).astype(np.float32)
    hd, md = torch.as_tensor(hr).to(dev), torch.as_tensor(mask).to(dev).view(torch.uint8)
    pd = torch.as_tensor(pred).to(dev).requires_grad_(True)
    torch.library.opcheck(torch.ops.probav.shift_metrics.default, (hd, md, pd.detach(), 3, 16))
    torch.library.opcheck(torch.ops.probav.shift_loss.default, (pd, hd, md, 3, 16, 1))
    torch.library.opcheck(torch.ops.probav.shift_loss.default, (pd, hd, md, 3, 16, 2))
    loss, arg, per = torch.ops.probav.shift_loss(pd, hd, md, 3, 16, 1)
    torch.library.opcheck(torch.ops.probav.shift_loss_backward.default, (hd, md, pd.detach(), arg, torch.ones(1, device=dev), 3, 1))
    from probav_amd.loss import Losses
    lo = Losses(targetShape=(48, 48, 1))
    assert float(lo.shiftCompensatedL1Loss(hd, md, pd.detach())) == float(loss) and abs(float(per.mean()) - float(loss)) < 1e-6 * float(loss)
    theta = torch.randn(1000, device=dev)
    g, mm, vv = torch.randn(1000, device=dev), torch.zeros(1000, device=dev), torch.zeros(1000, device=dev)
    torch.library.opcheck(torch.ops.probav.nadam_step.default, (theta, g, mm, vv, 5e-4, 0.9, 0.999, 1e-7, 1.0, 0.5, 2.0))
    xx = torch.tensor([-3.2, 0.5, 1.5, 2.5, 65535.5, 70000.0], device=dev)
    torch.library.opcheck(torch.ops.probav.clip_round.default, (xx, 0.0, 65536.0))
    assert torch.ops.probav.clip_round(xx, 0.0, 65536.0).tolist() == [0.0, 0.0, 2.0, 2.0, 65536.0, 65536.0]


def test_ops_trace_under_torch_compile_fullgraph(dev):
    """The ops are opaque, well-typed graph nodes: a step function using them traces with fullgraph=True (aot_eager backend: no codegen
    toolchain needed) and gives the eager numbers."""
    m = _model(dev)
    x, hr, mask = (torch.as_tensor(a).to(dev) for a in synth.synth_batch(2, seed=64))
    mk = mask.view(torch.uint8)
    eng = int(m._handle().value)

    def step(flat, x, hr, mk):
        y, _ = torch.ops.probav.wdsr_forward(flat, x, eng, 48, True)
        return torch.ops.probav.shift_loss(y, hr, mk, 3, 16, 1)[0]
    flat = m.flat.detach().clone().requires_grad_(True)
    want = step(flat, x, hr, mk)
    want.backward()
    gw = flat.grad.clone()
    flat.grad = None
    try:                                                       # ONLY a torch build without dynamo may skip; "aot_eager" needs no codegen toolchain
        import importlib
        importlib.import_module("torch._dynamo")               # (a plain `import torch._dynamo` here would make `torch` a local of this function)
        cstep = torch.compile(step, backend="aot_eager", fullgraph=True)
    except (ImportError, ModuleNotFoundError) as exc:
        pytest.skip("torch.compile (dynamo) is not part of this torch build: %r" % (exc,))
    got = cstep(flat, x, hr, mk)                               # a failure of the trace, of the call or of its backward FAILS the test
    got.backward(retain_graph=True)
    assert float(got) == float(want) and torch.equal(flat.grad, gw)
    flat.grad = None
    got.backward()                                             # the saved state is only read: a second reverse pass over the same graph gives the same bits
    assert torch.equal(flat.grad, gw)


def test_backward_is_functional_the_saved_state_is_untouched(dev):
    """probav::wdsr_backward reads the saved state of the forward pass and writes a scratch block of its own (probav_backward_split): the
    workspace tensor is bitwise unchanged by a reverse pass, two reverse passes over one retained graph agree bit for bit, and the one-piece
    C entry point (probav_backward on a workspace of probav_workspace_bytes) gives the same gradient."""
    import ctypes
    from probav_amd import _lib as L
    m = _model(dev)
    x, hr, mask = (torch.as_tensor(a).to(dev) for a in synth.synth_batch(2, seed=65))
    eng = m._handle()
    flat = m.flat.detach().clone().requires_grad_(True)
    y, ws = torch.ops.probav.wdsr_forward(flat, x, int(eng.value), 48, True)
    saved, scratch = ctypes.c_size_t(), ctypes.c_size_t()
    L.check(L.lib().probav_workspace_split(eng, 2, ctypes.byref(saved), ctypes.byref(scratch)))
    assert ws.numel() * 4 == saved.value and saved.value + scratch.value == L.lib().probav_workspace_bytes(eng, 2, 1)
    before = ws.clone()
    dy = torch.randn_like(y)
    (g1,) = torch.autograd.grad(y, flat, dy, retain_graph=True)
    assert torch.equal(ws.view(torch.int32), before.view(torch.int32))
    (g2,) = torch.autograd.grad(y, flat, dy)
    assert torch.equal(g1, g2)
    whole = torch.empty((saved.value + scratch.value) // 4, device=dev)
    y2, g3 = torch.empty_like(y), torch.empty_like(g1)
    s = L.current_stream()
    L.check(L.lib().probav_forward(eng, L.ptr(flat), L.ptr(x), L.ptr(y2), L.ptr(whole), whole.numel() * 4, 2, 1, s))
    L.check(L.lib().probav_backward(eng, L.ptr(flat), L.ptr(dy), L.ptr(g3), L.ptr(whole), whole.numel() * 4, 2, s))
    assert torch.equal(y2, y.detach()) and torch.equal(g3, g1)


def test_fused_optimizer_weight_norm_step(dev):
    """probav::optimizer_wn_step = Keras Nadam on all 132 tensors + the weight normalisation (and operand packing)
    of the UPDATED parameters in the same call.  (1) the parameters equal the plain fused Nadam launch and the fp64 restatement,
    (2) the cached effective weights equal oracle.weight_norm(updated v, updated g), (3) a forward/backward pass from the cache equals the
    pass that recomputes everything, bit for bit, and stops being used as soon as the parameters change behind its back."""
    from oracle import wdsr_numpy as on
    from oracle.nadam_numpy import Nadam
    from probav_amd import _lib as L
    from probav_amd.loss import Losses
    from probav_amd.trainClass import HipNadam, make_optimizer
    m = _model(dev)
    lo = Losses(targetShape=(48, 48, 1))
    x, hr, mask = (torch.as_tensor(a).to(dev) for a in synth.synth_batch(2, seed=71))
    opt = make_optimizer("nadam", m, 5e-4)
    assert isinstance(opt, HipNadam) and opt.model is m
    ref = Nadam(lr=5e-4)
    theta = m.flat.detach().cpu().double().numpy()
    plain = m.flat.detach().clone()
    popt = HipNadam([torch.nn.Parameter(plain)], lr=5e-4)              # the same rule through the plain element-wise launch
    for k in range(3):
        m.flat.grad = None
        assert (m.weight_cache() is not None) == (k > 0)
        lo.shiftCompensatedL1Loss(hr, mask, m(x, training=True)).backward()
        g = m.flat.grad.detach().clone()
        opt.step()
        popt.param_groups[0]["params"][0].grad = g
        popt.step()
        theta = ref.step(theta, g.cpu().double().numpy())
        d = (m.flat.detach() - popt.param_groups[0]["params"][0].detach()).abs().max()                        # (1) same rule, two kernels: rounding only
        assert float(d) < 1e-6 * float(m.flat.detach().abs().max()), float(d)
        assert np.abs(m.flat.detach().cpu().double().numpy() - theta).max() < 2e-6 * np.abs(theta).max()
    wc = m.weight_cache()
    assert wc is not None
    # (2) weff inside the cache (its first block, layers in order, [tap][ci][co]) against the oracle on the updated parameters
    flat = m.flat.detach().cpu().numpy()
    params = synth.unflatten_params(flat)
    nw = L.lib().probav_weff_count(m._handle())
    weff = wc[:nw].cpu().double().numpy()
    off = 0
    for Lh in m.layers:
        w = on.weight_norm(params[Lh.name]["v"], params[Lh.name]["g"])
        assert np.abs(weff[off:off + w.size].reshape(w.shape) - w).max() < 2e-6 * np.abs(w).max(), Lh.name
        off += w.size
    # (3) cached pass == recomputing pass, bit for bit
    m.flat.grad = None
    y_c = m(x, training=True)
    lo.shiftCompensatedL1Loss(hr, mask, y_c).backward()
    g_c = m.flat.grad.detach().clone()
    m._wcache_version = None                                            # drop the cache: the same parameters, everything recomputed
    m.flat.grad = None
    y_r = m(x, training=True)
    lo.shiftCompensatedL1Loss(hr, mask, y_r).backward()
    assert torch.equal(y_c, y_r) and torch.equal(g_c, m.flat.grad)
    opt.step()
    assert m.weight_cache() is not None
    with torch.no_grad():
        m.flat.mul_(1.0)                                                # any in-place change of the parameters invalidates the cache
    assert m.weight_cache() is None


def test_weight_cache_invalidation_after_writes_behind_the_version_counter(dev, monkeypatch):
    """`flat.data.copy_()`, a raw-pointer write (ctypes probav_nadam_step) or `dist.broadcast(flat.data)` do not bump the
    version counter the cache is keyed on.  `invalidate_weight_cache()` is the documented hand-over; `load_variables`,
    `load_state_dict` and `.to()` call it themselves; PROBAV_CHECK_WCACHE=1 turns a forgotten call into an error instead of a pass on
    stale weights; a second backward over one graph (retain_graph) is legal (the reverse pass mutates scratch, not saved state)."""
    m = _model(dev)
    x = torch.as_tensor(synth.synth_batch(2, seed=81)[0]).to(dev)
    with torch.no_grad():
        y0 = m(x)                                                       # inference: builds the cache for the current weights
    assert m.weight_cache() is not None
    new = synth.flatten_params(synth.synth_params(seed=82, perturb=True))
    m.flat.data.copy_(torch.as_tensor(new))                             # behind the counter: the cache still claims to be valid
    assert m.weight_cache() is not None
    monkeypatch.setenv("PROBAV_CHECK_WCACHE", "1")
    with pytest.raises(RuntimeError, match="stale weight cache"):
        m.weight_cache()
    monkeypatch.delenv("PROBAV_CHECK_WCACHE")
    m.invalidate_weight_cache()
    assert m.weight_cache() is None
    with torch.no_grad():
        y1 = m(x)
    fresh = _model(dev)
    fresh.load_variables(synth.synth_params(seed=82, perturb=True))
    with torch.no_grad():
        y_ref = fresh(x)
    assert torch.equal(y1, y_ref) and not torch.equal(y1, y0)
    # the module-level writers invalidate by themselves
    m.load_variables(synth.synth_params(seed=83, perturb=True))
    assert m.weight_cache() is None
    with torch.no_grad():
        m(x)
    assert m.weight_cache() is not None
    m.load_state_dict(fresh.state_dict())
    assert m.weight_cache() is None
    with torch.no_grad():
        assert torch.equal(m(x), y_ref)
    monkeypatch.setenv("PROBAV_CHECK_WCACHE", "1")
    with torch.no_grad():
        assert torch.equal(m(x), y_ref)                                 # a valid cache passes the check
    monkeypatch.delenv("PROBAV_CHECK_WCACHE")
    # retain_graph: two reverse passes over one forward give the same gradient
    y = m(x, training=True)
    s = y.sum()
    s.backward(retain_graph=True)
    g1 = m.flat.grad.clone()
    m.flat.grad = None
    s.backward()
    assert torch.equal(g1, m.flat.grad)
    from probav_amd import ops
    del y, s
    ops.release_workspaces()
    with torch.no_grad():
        assert torch.equal(m(x), y_ref)                                 # a released pool is simply rebuilt


def test_default_workspace_reference_is_weak(dev, monkeypatch):
    """Without PROBAV_KEEP_WS the model holds no strong reference to a pass's multi-GB workspace: it is alive while the autograd graph
    needs it, gone after the backward, and asking for it then says so (not 'no forward pass has run')."""
    import gc
    monkeypatch.delenv("PROBAV_KEEP_WS", raising=False)
    m = _model(dev)
    x = torch.as_tensor(synth.synth_batch(2, seed=66)[0]).to(dev)
    with pytest.raises(RuntimeError, match="no forward pass"):
        m._workspace(2, True)
    y = m(x, training=True)
    assert m._workspace(2, True).numel() > 0                   # held by the graph of y
    y.sum().backward()
    del y
    gc.collect()
    with pytest.raises(RuntimeError, match="already been released"):
        m._workspace(2, True)
