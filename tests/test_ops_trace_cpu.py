"""The autograd formulas of the torch custom ops trace under AOT-autograd -- checked WITHOUT a GPU: torch.compile(fullgraph=True) runs the
fake-tensor rules and the registered backward formulas on fake tensors and hands the joint forward + backward graph to the partitioner
before anything executes.  A formula that is not traceable (round 3: a raw storage alias inside the backward of probav::wdsr_forward) fails
here, in the CPU suite, instead of hiding behind a skip on the GPU box (tests/test_gpu_ops.py runs the compiled step for real)."""
import pytest
import torch


class _Stop(Exception):
    pass


def test_train_step_traces_with_its_backward_on_fake_tensors(built_lib, monkeypatch):
    pytest.importorskip("torch._dynamo")
    from functorch.compile import min_cut_rematerialization_partition
    from torch._functorch.aot_autograd import aot_module_simplified
    from probav_amd import ops
    # the sizes come from the engine (a device object): any constant does for the trace
    monkeypatch.setattr(ops, "_ws_floats", lambda engine, batch, training: 4096)
    seen = {}

    def fw_compiler(gm, example_inputs):
        seen["fw"] = gm.print_readable(print_output=False)

        def run(*args):                                    # nothing can execute here: the kernels exist for gfx950 only
            raise _Stop()
        return run

    def partition(joint, inputs, **kw):
        seen["joint"] = joint.print_readable(print_output=False)
        return min_cut_rematerialization_partition(joint, inputs, **kw)

    def backend(gm, example_inputs):
        return aot_module_simplified(gm, example_inputs, fw_compiler=fw_compiler, bw_compiler=lambda g, e: g, partition_fn=partition)

    def step(flat, x, hr, mk):
        y, _ = torch.ops.probav.wdsr_forward(flat, x, 1234, 48, True)
        return torch.ops.probav.shift_loss(y, hr, mk, 3, 16, 1)[0]

    flat = torch.zeros(535267, requires_grad=True)
    x, hr = torch.zeros(2, 22, 22, 9, 1), torch.zeros(2, 48, 48, 1)
    mk = torch.ones(2, 48, 48, 1, dtype=torch.uint8)
    torch._dynamo.reset()
    with pytest.raises(_Stop):
        torch.compile(step, backend=backend, fullgraph=True)(flat, x, hr, mk)
    torch._dynamo.reset()
    assert "probav.wdsr_forward" in seen["fw"]
    # the joint graph holds the reverse pass as two opaque, FUNCTIONAL nodes (no copy_ back into the saved workspace)
    assert "probav.wdsr_backward" in seen["joint"] and "probav.shift_loss_backward" in seen["joint"]
    assert "copy_" not in seen["joint"]


def test_wdsr_backward_declares_no_mutation():
    import probav_amd.ops  # noqa: F401
    schema = str(torch.ops.probav.wdsr_backward.default._schema)
    assert "!" not in schema, schema
