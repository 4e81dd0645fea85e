"""The engine's side stream (slab sums, the low-frequency residual path) must not change a single bit: every sum keeps its fixed order
whichever stream it runs on.  PROBAV_NO_SIDE_STREAM is read once per process, so the two runs are child processes."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import hashlib, sys
sys.path.insert(0, %r)
import torch
from probav_amd import synth
from probav_amd.loss import Losses
from probav_amd.modelsTF import WDSRConv3D
dev = torch.device("cuda:0")
B, T, REPS = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
model = WDSRConv3D("s", "NIR", synth.NIR_MEAN, synth.NIR_STD, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, T, 16, True)
model.load_variables(synth.synth_params(seed=11, perturb=True, numImgLR=T))
model = model.to(dev)
x, hr, mask = (torch.as_tensor(a).to(dev) for a in synth.synth_batch(B, seed=12, numImgLR=T))
losses = Losses(targetShape=(48, 48, 1))
h = hashlib.sha256()
for _ in range(REPS):                                # repeatedly: later passes reuse the pool's workspace and the side stream
    pred = model(x, training=True)
    loss = losses.shiftCompensatedL1Loss(hr, mask, pred)
    model.flat.grad = None
    loss.backward()
    torch.cuda.synchronize()
    h.update(pred.detach().cpu().numpy().tobytes()); h.update(model.flat.grad.cpu().numpy().tobytes())
print("DIGEST", h.hexdigest())
if len(sys.argv) > 4:                                # the last pass' values, for comparisons with a tolerance
    import numpy as np
    np.savez(sys.argv[4], pred=pred.detach().cpu().numpy(), grad=model.flat.grad.cpu().numpy())
""" % ROOT


def _run(extra_env, B=5, T=9, reps=2, dump=None):
    env = dict(os.environ)
    env.pop("PROBAV_NO_SIDE_STREAM", None)
    env.update(extra_env)
    out = subprocess.run([sys.executable, "-c", SCRIPT, str(B), str(T), str(reps)] + ([dump] if dump else []), env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return [l for l in out.stdout.splitlines() if l.startswith("DIGEST")][0]


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,reps", [(5, 9, 2), (128, 9, 6), (128, 13, 4)], ids=["b5-t9", "b128-t9", "b128-t13"])
def test_side_stream_changes_no_bit(B, T, reps):
    """Default mode 2 (slab sums, residual path AND the backward-filter kernels at the lowest priority beside the chain, reading
    `gblk` / `gred` late) against no side stream at all.  A late-read race would be timing dependent: besides the small case, the
    headline size (BASELINE.json config 2, batch 128) and config 3 (T = 13), several passes each."""
    assert _run({}, B, T, reps) == _run({"PROBAV_NO_SIDE_STREAM": "1"}, B, T, reps)


@pytest.mark.gpu
def test_training_step_replays_from_a_captured_graph():
    """include/probav_hip.h promises calls that only enqueue (graph-capturable): forward + loss + backward captured once (the fork to the
    side stream and its join are recorded with it), replayed, and compared bit for bit with the eager launches."""
    import torch
    sys.path.insert(0, ROOT)
    from probav_amd import synth
    from probav_amd.loss import Losses
    from probav_amd.modelsTF import WDSRConv3D
    dev = torch.device("cuda:0")
    model = WDSRConv3D("g", "NIR", synth.NIR_MEAN, synth.NIR_STD, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True)
    model.load_variables(synth.synth_params(seed=21, perturb=True))
    model = model.to(dev)
    x, hr, mask = (torch.as_tensor(a).to(dev) for a in synth.synth_batch(3, seed=22))
    losses = Losses(targetShape=(48, 48, 1))

    def step():
        pred = model(x, training=True)
        loss = losses.shiftCompensatedL1Loss(hr, mask, pred)
        model.flat.grad = None
        loss.backward()
        return loss

    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False) if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch") else None
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):                               # warm-up on the capture stream: engine, side stream, allocator pools
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    model.flat.grad = None
    with torch.cuda.graph(g):
        loss_g = step()
    grad_g = model.flat.grad
    g.replay()
    torch.cuda.synchronize()
    lg, gg = float(loss_g.detach()), grad_g.clone()
    le = float(step().detach())
    torch.cuda.synchronize()
    assert lg == le
    assert torch.equal(gg, model.flat.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T", [(5, 9), (3, 13)], ids=["b5-t9", "b3-t13"])
def test_one_wave_per_simd_pointwise_backward_agrees_with_the_general_form(B, T, tmp_path):
    """The fused pointwise backward of round 5 (pw_bwd_w4_kernel: one wave per SIMD, a wave owns a tile and all eight hidden chunks) evaluates the products
    of the general eight-wave form (pw_bwd_x6_kernel<H3>, PROBAV_GEN1=pwb) per tile and chunk, with the same recomputed hidden tile and so the same ReLU gates; what differs is
    the order of the fp32 additions across chunks (dX: one accumulator chain instead of eight partials) and across tiles (filter gradients: a wave's run
    instead of a workgroup's).  Whole network: identical predictions (the forward pass is untouched), the flat gradient vector to 1e-5 of its max-norm."""
    import numpy as np
    a, b = str(tmp_path / "w4.npz"), str(tmp_path / "gen1.npz")
    _run({}, B, T, 1, a)
    _run({"PROBAV_GEN1": "pwb"}, B, T, 1, b)
    A, Bv = np.load(a), np.load(b)
    assert np.array_equal(A["pred"], Bv["pred"])
    assert np.abs(A["grad"] - Bv["grad"]).max() <= 1e-5 * np.abs(Bv["grad"]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("B,T", [(5, 9), (3, 13)], ids=["b5-t9", "b3-t13"])
def test_one_wave_per_simd_pointwise_forward_agrees_with_the_eight_wave_kernel(B, T, tmp_path):
    """The fused pointwise forward of round 5 (pw_fwd_w4_kernel: one wave per SIMD, the weight fragments in registers) sums the 32 input channels of a hidden value in two
    k-blocks of v_mfma_f32_32x32x16_f16 -- the order of the reverse pass' recompute --, pw_fwd_h3k_kernel (PROBAV_GEN1=pwf) in one 16x16x32 instruction: the same scaled fp16
    pieces, another order of the fp32 additions.  Whole network: predictions to 1e-5 of their max-norm; the gradient to 1e-3 (a forward pass that differs in the last bits
    flips gates of hidden values at zero, see the convolution test below).  Bit-for-bit agreement with the 32x32x16 arrangement itself: tools/pf4bench.hip and
    test_forward_and_reverse_pass_decide_the_same_relu_gates."""
    import numpy as np
    a, b = str(tmp_path / "w4.npz"), str(tmp_path / "h3k.npz")
    _run({}, B, T, 1, a)
    _run({"PROBAV_GEN1": "pwf"}, B, T, 1, b)
    A, Bv = np.load(a), np.load(b)
    assert not np.array_equal(A["pred"], Bv["pred"])                      # (the switch did select another kernel)
    assert np.abs(A["pred"] - Bv["pred"]).max() <= 1e-5 * np.abs(Bv["pred"]).max()
    assert np.abs(A["grad"] - Bv["grad"]).max() <= 1e-3 * np.abs(Bv["grad"]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("B,T", [(5, 9), (2, 7)], ids=["b5-t9", "b2-t7"])
def test_one_wave_per_simd_convolution_agrees_with_the_general_form(B, T, tmp_path):
    """The residual blocks' 3x3x3 convolution of round 5 (conv3_w4_kernel: one wave per SIMD, the filter's first pieces in registers, a tile's 45 / 54 k-blocks ONE
    accumulation chain that starts from the bias) evaluates the products of the eight-wave piece-ring kernel (conv3_pp_kernel, PROBAV_GEN1=conv: two waves' partial sums
    added, the bias added behind the scaling) -- the same scaled fp16 pieces, another order of the fp32 additions.  Whole network: the predictions agree to 1e-5 of their max-norm.  The reverse
    pass is held to 1e-3 of the gradient's max-norm only: a forward pass that differs in the last bits flips the ReLU gate of hidden values that sit at zero (12 blocks x 256
    channels x every voxel of the batch), and a flipped gate adds or drops a whole term of dX and dW1 -- with the forward pass untouched (the pointwise test above) the
    same comparison holds 1e-5.  (Each kernel against the fp32 oracle: tests/test_gpu_parity.py.)"""
    import numpy as np
    a, b = str(tmp_path / "w4.npz"), str(tmp_path / "gen1.npz")
    _run({}, B, T, 1, a)
    _run({"PROBAV_GEN1": "conv"}, B, T, 1, b)
    A, Bv = np.load(a), np.load(b)
    assert not np.array_equal(A["grad"], Bv["grad"])                      # (the switch did select another kernel)
    assert np.abs(A["pred"] - Bv["pred"]).max() <= 1e-5 * np.abs(Bv["pred"]).max()
    assert np.abs(A["grad"] - Bv["grad"]).max() <= 1e-3 * np.abs(Bv["grad"]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("B,T", [(5, 9), (2, 7), (128, 9), (3, 13)], ids=["b5-t9", "b2-t7", "b128-t9", "b3-t13"])
def test_one_wave_per_simd_backward_filter_agrees_with_the_general_form(B, T, tmp_path):
    """The backward-filter of the residual blocks and of the reducers in round 5 (conv3_wgrad_w4_kernel: one wave per SIMD, 64-byte piece planes of the input ring, dY cut
    once per workgroup, a tile's k-blocks ONE accumulation chain per row; the reducers' layer with mirrored pads, no depth pads and dY masked by the layer's output) evaluates
    the products of the general form (conv3_wgrad_x6_kernel<25 | 32, H3>, PROBAV_GEN1=wg: two k-block parities added at the end) -- the same scaled fp16 pieces with the same
    two tensor-wide scales, another order of the fp32 additions inside a workgroup's slab.  The forward pass and every other gradient are untouched: identical predictions, the
    flat gradient vector to 1e-5 of its max-norm.  (T = 9: reducer rows of 10 / 7 / 4 k-blocks, the last two unpadded; T = 7: 7 / 5; T = 13: column halves for normConv and the
    first reducer.  Since the end of round 5 the one-wave-per-SIMD kernel's second pieces are lifted by 2^11 and its cross products have their own accumulator: the same products, summed
    more exactly.)"""
    import numpy as np
    a, b = str(tmp_path / "w4.npz"), str(tmp_path / "gen1.npz")
    _run({}, B, T, 1, a)
    _run({"PROBAV_GEN1": "wg"}, B, T, 1, b)
    A, Bv = np.load(a), np.load(b)
    assert np.array_equal(A["pred"], Bv["pred"])
    assert not np.array_equal(A["grad"], Bv["grad"])                      # (the switch did select another kernel)
    assert np.abs(A["grad"] - Bv["grad"]).max() <= 1e-5 * np.abs(Bv["grad"]).max()
