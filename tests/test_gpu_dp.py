"""The data-parallel step on the hardware a box has: a WORLD-SIZE-1 RCCL group (PROBAV_FORCE_DP=1 / bench.py --force-dp) runs every
collective of the N > 1 path for real -- RCCL initialised with the device id, the all-reduce on the engine's flat 2.14 MB gradient
behind the engine's side-stream join, the private workspace pool beside RCCL's buffers, the trainer's one-collective step
(gradient + loss / metric in one bucket: trainClass.GradBucket) -- and, a mean over one rank being the identity, must not change a bit.
Reference semantics: debug/trainClassMultiGPU0.py:67-84 (per-replica batch), :153 (gradient all-reduce), :162-178 (strategy.reduce(MEAN)).
Each run is a child process: a process group is process-global state."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env(**extra):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "PROBAV_FORCE_DP"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    env.update(extra)
    return env


BENCH_FLAGS = ["--steps", "3", "--warmup", "2", "--batch", "8", "--no-cpu-baseline", "--no-other-configs", "--no-fp32-mfma-leg",
               "--no-kernel-events", "--digest"]


def _bench_line(cmd, env):
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-1500:]
    return json.loads(lines[0])


def test_bench_step_under_a_world_size_1_rccl_group_is_bitwise_the_plain_step(dev):
    plain = _bench_line([sys.executable, "bench.py"] + BENCH_FLAGS, _env())
    assert plain["config"]["world_size"] == 1 and plain["config"]["backend"] is None and not plain["config"]["forced_dp"]
    # (1) the way the driver starts ranks: torch.distributed.run, one process, RCCL, the all-reduce forced
    launched = _bench_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                            "--master-port", str(_free_port()), "bench.py"] + BENCH_FLAGS, _env(PROBAV_FORCE_DP="1"))
    # (2) the flag, without a launcher
    flagged = _bench_line([sys.executable, "bench.py", "--force-dp"] + BENCH_FLAGS, _env())
    for line in (launched, flagged):
        c = line["config"]
        assert c["world_size"] == 1 and c["backend"] == "nccl (RCCL)" and c["forced_dp"] is True and "all-reduce" in c["workload"]
        assert line["n_gpus"] == 1 and line["digest"] == plain["digest"], "the RCCL all-reduce over one rank changed the gradient"
        # the self-diagnosis of the first real N > 1 run, exercised on the one GPU there is: per-rank step, the all-reduce's own time, identical replicas
        d = line["dp"]
        assert d["world_size"] == 1 and len(d["per_rank_step_ms_median"]) == 1 and d["replicas_bitwise_identical"] is True
        assert 0.0 < d["allreduce_ms_per_step_median_per_rank"][0] < d["per_rank_step_ms_median"][0] and d["allreduce_payload_bytes"] == 535267 * 4
    assert "dp" not in plain


TRAINER = r"""
import hashlib, os, sys
sys.path.insert(0, %r)
import numpy as np
import torch
import torch.distributed as dist
from probav_amd import synth
from probav_amd.loss import Losses
from probav_amd.modelsTF import WDSRConv3D
from probav_amd.trainClass import GradBucket, HipNadam, ModelTrainer, dp_state, make_optimizer
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
forced = os.environ.get("PROBAV_FORCE_DP") == "1"
if forced:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dp_state() == (forced, 1)
model = WDSRConv3D("dp", "NIR", synth.NIR_MEAN, synth.NIR_STD, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True)
model.load_variables(synth.synth_params(seed=41, perturb=True))
model = model.to(dev)
if forced:
    dist.broadcast(model.flat.data, src=0)            # what a multi-rank job does after loading: a write behind the version counter ...
    model.invalidate_weight_cache()                   # ... so the cache is told
lo = Losses(targetShape=(48, 48, 1))
opt = make_optimizer("nadam", model, 5e-4)
assert isinstance(opt, HipNadam) and opt.model is model          # the fused optimizer + weight-cache path
tr = ModelTrainer(model, lo.shiftCompensatedL1Loss, lo.shiftCompensatedcPSNR, opt, sys.argv[1] + "/ck", sys.argv[1] + "/lg", multiGPU=True)
assert tr._dp() == forced
calls = []
if forced:
    real = dist.all_reduce
    def counting(t, *a, **k):
        calls.append(int(t.numel()))
        return real(t, *a, **k)
    dist.all_reduce = counting
h = hashlib.sha256()
for k in range(3):
    x, hr, mask = (torch.as_tensor(a).to(dev) for a in synth.synth_batch(6, seed=50 + k))
    tr.trainStep(x, hr, mask)
    assert (model.weight_cache() is not None)         # the fused optimizer left the next step's weights behind
    torch.cuda.synchronize()
    h.update(model.flat.detach().cpu().numpy().tobytes()); h.update(model.flat.grad.detach().cpu().numpy().tobytes())
x, hr, mask = (torch.as_tensor(a).to(dev) for a in synth.synth_batch(4, seed=60))
tr.testStep(x, hr, mask)
if forced:
    n = model.flat.numel()
    assert calls == [n + 2] * 3 + [2], calls          # ONE collective per training step (gradient + 2 scalars), one 2-float reduce per test step
    dist.barrier()
    dist.destroy_process_group()
print("LOSS %%.9g %%.9g %%.9g %%.9g" %% (tr.trainLoss.result(), tr.trainPSNR.result(), tr.testLoss.result(), tr.testPSNR.result()))
print("DIGEST", h.hexdigest())
""" % ROOT


def _trainer(tmp, **extra):
    os.makedirs(tmp, exist_ok=True)
    out = subprocess.run([sys.executable, "-c", TRAINER, tmp], env=_env(MASTER_PORT=str(_free_port()), **extra), capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    dig = [l for l in out.stdout.splitlines() if l.startswith("DIGEST")][0]
    vals = [float(v) for v in [l for l in out.stdout.splitlines() if l.startswith("LOSS")][0].split()[1:]]
    return dig, vals


def test_trainer_with_the_real_model_under_rccl_matches_the_plain_run_bit_for_bit(dev, tmp_path):
    """ModelTrainer.trainStep x 3 with the real WDSRModel, HipNadam fused with the weight cache, multiGPU=True: parameters and gradients
    after every step are bitwise those of the run without a process group; the logged means agree to fp32 rounding (the bucket carries
    the replica means as two fp32 values)."""
    d_plain, v_plain = _trainer(str(tmp_path / "plain"))
    d_dp, v_dp = _trainer(str(tmp_path / "dp"), PROBAV_FORCE_DP="1")
    assert d_plain == d_dp
    for a, b in zip(v_plain, v_dp):
        assert abs(a - b) <= 2e-7 * abs(a), (v_plain, v_dp)
