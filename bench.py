#!/usr/bin/env python3
"""bench.py -- LR-patches/sec, fwd+bwd, WDSR-B r12 t9, 16x16 patches (+6 px border), batch 128 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` with N > 1 starts its own ranks: the parent process (which never touches the GPU and never imports torch) runs
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...` as a CHILD,
forwards its output (rank 0 prints the JSON line) and exits with its status.  Launched under torch.distributed.run already
(RANK / WORLD_SIZE in the environment), the script is one rank of that job.

One "step" = one pass of the hot path over one batch of synthetic patches already resident in HBM:
model forward (weight-norm, head, 12 WDSR-B blocks, reducers, pixel shuffle), shift-compensated L1 loss
forward, and the full backward to all 132 parameter gradients (BASELINE.json metric) --
plus, for N > 1, the one gradient all-reduce that data parallelism adds (RCCL over xGMI).
The optimizer update and the cPSNR metric are not part of "fwd+bwd"; `--full-step` times them too and
reports the result under "full_step" without changing `value`.

Rank 0 prints ONE JSON line.  `value` comes from the wall clock around exactly K steps (barrier + synchronize on both
sides, max over ranks); `step_ms` (median, p10, p90) from HIP events recorded on the launch stream at every step boundary.
`roofline` is measured live: the engine brackets the launches of the dominant kernel class with HIP events on the launch
stream; `achieved` = SURVEY.md §8d's algorithmic MACs of that class / its time.
`cpu_baseline` times the CPU oracle (torch-CPU restatement of the TF path; the TF reference itself cannot run here) on the
host cores at the reference's CPU-runnable config (batch 8).  `other_configs` carries BASELINE.json's config 3 (T = 13)
and config 4 (full-frame inference) from short extra runs.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CLASSES = ["weight_norm", "small", "conv3x3x3_fwd", "conv3x3x3_bwd_data", "conv3x3x3_wgrad",
           "conv1x1x1_fwd", "conv1x1x1_bwd_data", "conv1x1x1_wgrad",
           "conv3x3x3_fwd_x6", "conv3x3x3_bwd_data_x6", "conv3x3x3_wgrad_x6", "conv1x1x1_fwd_x6", "conv1x1x1_bwd_data_x6"]
PEAK_F32_TFLOPS = 157.3            # MI355X_MICROARCH.md: fp32 vector == fp32 MFMA peak
PEAK_BF16_TFLOPS = 2500.0          # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak (~2.5 PF)
SPLIT_PRODUCTS = {3: 6, 4: 3}      # 16-bit MFMA products issued per fp32 product: x6 kernels (bf16 pieces) / H3 kernels (scaled fp16 pieces)
PEAK_HBM_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E peak
ALGO_MB_PER_PATCH = 402.414        # layer-boundary byte model, fwd + bwd (un-fused)
PLAN_MB_PER_PATCH = 81.0           # bytes the fused plan moves (256-channel tensor never reaches HBM)
ALGO_GFLOP_PER_PATCH = 12.436      # SURVEY.md §8d / BASELINE.md §2: fwd + bwd, p16t9c85r12
PW_BWD_ALGO_MAC, PW_BWD_ISSUED_MAC = 29184, 37376     # per voxel: §8d (bwd of expConv + decConv) / incl. the recompute of the hidden tile
IMPL_NAMES = {0: "direct", 1: "mfma-rowtile", 2: "mfma-strip", 3: "x6-split-bf16", 4: "h3-split-fp16"}
DTYPES = {3: "f32 (products as 6 bf16-piece MFMA products, f32 accumulate)", 4: "f32 (products as 3 scaled-fp16-piece MFMA products, f32 accumulate)"}


def hbm_profile_path():
    for name in ("r06_hbm_traffic.json", "r05_hbm_traffic.json", "r04_hbm_traffic.json", "r03_hbm_traffic.json", "r02_hbm_traffic.json", "r01_hbm_traffic.json"):
        p = os.path.join(ROOT, "profiles", name)
        if os.path.exists(p):
            return p
    return None


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=128, help="patches per GPU (BASELINE.json: 128)")
    ap.add_argument("--frames", type=int, default=9, help="numImgLR: 9 (headline), 13 or 7")
    ap.add_argument("--impl", type=int, default=4, help="4 = H3 kernels: fp32 products as three products of scaled fp16 piece pairs on the 16-bit MFMA pipe (default), "
                    "3 = x6 kernels: six bf16-piece products, "
                    "2 = native fp32 MFMA + strip convolution, 1 = fp32 MFMA row-tile kernels, 0 = generic direct kernels")
    ap.add_argument("--no-fp32-mfma-leg", action="store_true", help="skip the short extra run on the native fp32-MFMA kernels (impl 2)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short T=13 training and full-frame inference legs")
    ap.add_argument("--full-step", action="store_true", help="also time loss+metric+Nadam (reported separately)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spin-up", type=int, default=2, help="untimed steps between the interpreter's collector run and the synchronize that opens every timed region")
    ap.add_argument("--no-power", action="store_true", help="do not start the child process that samples board power and sclk (always give this under rocprofv3)")
    ap.add_argument("--no-trainer-loop", action="store_true", help="skip the ModelTrainer.fitTrainData leg of other_configs")
    ap.add_argument("--trainer-steps", type=int, default=240, help="steps of the ModelTrainer.fitTrainData leg (>= 200 for a stable median)")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--side-stream-mode", type=int, default=-1, choices=(-1, 0, 1, 2),
                    help="probav_engine_side_stream: 2 = slab sums, residual path AND the backward-filter kernels on the engine's low-priority "
                         "side stream; 1 = backward-filter kernels on the launch stream (per-kernel profiles: rocprofv3 / PMC passes); 0 = no side stream; "
                         "-1 (default) = WDSRModel.tune_side_stream picks 1 or 2 in a few untimed steps behind the warm-up (which is faster depends on the box)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the ranks (nccl = RCCL; gloo only with --dry-run)")
    ap.add_argument("--force-dp", action="store_true", help="with ONE rank: still initialise a world-size-1 RCCL process group and run the data-parallel "
                    "step's gradient all-reduce for real (PROBAV_FORCE_DP=1): the N > 1 code path on the one GPU a box has")
    ap.add_argument("--digest", action="store_true", help="add sha256 of (loss, flat gradient) of the last timed step to the line (bitwise comparisons between modes)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous / collective check without a GPU: every rank all-reduces a "
                    "gradient-sized buffer over --backend and rank 0 prints a JSON line (CPU test of the N > 1 path)")
    return ap.parse_args(argv)


# -----------------------------------------------------------------------------------------------------------------------------
# parent: start one process per GPU.  Nothing here may touch the GPU (no torch import): the ranks are children, never an exec
# -----------------------------------------------------------------------------------------------------------------------------
def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["PROBAV_BENCH_CHILD"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, env=env)
    try:
        rc = proc.wait()
    except KeyboardInterrupt:
        proc.terminate()
        rc = proc.wait()
    return rc


# -----------------------------------------------------------------------------------------------------------------------------
def cpu_baseline(seconds=20.0):
    """The oracle's torch-CPU restatement, fp32, all host cores, BASELINE.json config 1 (batch 8)."""
    import torch
    from oracle import wdsr_torch as ot
    from probav_amd import synth
    params = ot.to_torch_params(synth.synth_params(seed=1234), dtype=torch.float32)
    x, hr, mask = (torch.as_tensor(a) for a in synth.synth_batch(8, seed=1234))
    # oneDNN on a batch of 8 small patches does not scale to hundreds of threads (one step took 54 s with
    # 256 threads on the GPU host): pick the fastest thread count from a short calibration, and report it.
    avail = os.cpu_count() or 1
    best, cores = None, 1
    for nthr in sorted({min(avail, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(nthr)
        ot.train_step_grads(x, hr, mask, params, synth.NIR_MEAN, synth.NIR_STD)      # warm-up at this width
        t0 = time.perf_counter()
        ot.train_step_grads(x, hr, mask, params, synth.NIR_MEAN, synth.NIR_STD)
        dt1 = time.perf_counter() - t0
        if best is None or dt1 < best:
            best, cores = dt1, nthr
        if dt1 > 15.0:
            break
    torch.set_num_threads(cores)
    n, t0 = 0, time.perf_counter()
    while True:
        ot.train_step_grads(x, hr, mask, params, synth.NIR_MEAN, synth.NIR_STD)
        n += 1
        if time.perf_counter() - t0 >= seconds or n >= 40:
            break
    dt = time.perf_counter() - t0
    return {"value": round(8 * n / dt, 3), "unit": "patches/s", "cores": cores, "threads": cores, "host_cores": avail, "kind": "port",
            "sample": "oracle/wdsr_torch.py fp32, batch 8 (cfg p16t9c85r12 on CPU), fwd + L1 loss + bwd, %d steps in %.1f s; `cores` = `threads` = "
                      "the torch thread count used (the fastest of 8 / 16 / 32 / 64 in a short calibration: oneDNN does not scale further at "
                      "this size), `host_cores` = os.cpu_count() of the box" % (n, dt)}


def percentiles(ms):
    s = sorted(ms)
    q = lambda f: s[min(len(s) - 1, max(0, int(round(f * (len(s) - 1)))))]
    return {"median": round(q(0.5), 4), "p10": round(q(0.1), 4), "p90": round(q(0.9), 4), "min": round(s[0], 4), "max": round(s[-1], 4), "n": len(s),
            "argmax": max(range(len(ms)), key=lambda i: ms[i])}       # which step of the timed region was the slowest (0 = the first)


# -----------------------------------------------------------------------------------------------------------------------------
# Board power and shader clock beside the timed region.  A CHILD process that never touches the GPU (no torch, no HIP) reads the
# amdgpu hwmon files (power1_input in microwatts, freq1_input = sclk in Hz) every ~10 ms and prints "time power sclk" lines; the bench keeps the
# samples that fall inside the timed region's wall-clock window.  Not started under rocprofv3 (--no-power, and skipped by itself when a profiler's
# library is preloaded): the profiler's tool library would initialise in the child as well.
_SAMPLER_SRC = r"""
import glob, sys, time
paths = sys.argv[1:]
def rd(p):
    try:
        with open(p) as fh:
            return fh.read().strip()
    except OSError:
        return "nan"
while True:
    t = time.time()
    sys.stdout.write(" ".join([repr(t)] + [rd(p + "/power1_input") + " " + rd(p + "/freq1_input") for p in paths]) + "\n")
    sys.stdout.flush()
    time.sleep(0.008)
"""


class PowerSampler:
    def __init__(self, pci_bus_id=None):
        import glob
        self.proc, self.paths, self.pci = None, [], pci_bus_id
        if pci_bus_id:
            self.paths = sorted(glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % pci_bus_id))
        if not self.paths:                      # the device's PCI address is unknown: every board's sensors, the one whose power moves with the run is ours
            self.paths = sorted(p for p in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*") if os.path.exists(p + "/power1_input"))
            self.pci = None
        self.paths = self.paths[:16]

    def start(self):
        if not self.paths or any(k in os.environ.get("LD_PRELOAD", "") for k in ("rocprof", "roctracer")) or os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD"):
            return False
        try:
            self.proc = subprocess.Popen([sys.executable, "-c", _SAMPLER_SRC] + self.paths, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                         env={k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD",)}, text=True)
        except OSError:
            self.proc = None
        return self.proc is not None

    def stop(self, windows):
        """windows: {name: (t0, t1)} in time.time() -> {name: {board_power_W, sclk_GHz, ...}} of the samples inside each window."""
        if self.proc is None:
            return None
        self.proc.terminate()
        try:
            text, _ = self.proc.communicate(timeout=5)
        except subprocess.TimeoutExpired:
            self.proc.kill()
            text, _ = self.proc.communicate()
        rows = []
        for line in text.splitlines():
            f = line.split()
            if len(f) == 1 + 2 * len(self.paths):
                try:
                    rows.append([float(v) for v in f])
                except ValueError:
                    pass
        if not rows:
            return None
        col = 0
        if len(self.paths) > 1:                 # the board whose power rose the most above its first samples
            base = rows[:5]
            rise = [max(r[1 + 2 * c] for r in rows) - sum(r[1 + 2 * c] for r in base) / len(base) for c in range(len(self.paths))]
            col = max(range(len(self.paths)), key=lambda c: rise[c] if rise[c] == rise[c] else -1)
        out = {"sensor": self.paths[col], "pci_bus_id": self.pci, "sample_period_ms": round((rows[-1][0] - rows[0][0]) / max(1, len(rows) - 1) * 1e3, 2),
               "note": "amdgpu hwmon power1_input (socket power) and freq1_input (sclk) read by a child process that never touches the GPU; mean / "
                       "min / max of the samples inside each wall-clock window"}
        for name, (t0, t1) in windows.items():
            sel = [r for r in rows if t0 <= r[0] <= t1 and r[1 + 2 * col] == r[1 + 2 * col]]
            if not sel:
                out[name] = None
                continue
            pw = [r[1 + 2 * col] * 1e-6 for r in sel]
            ck = [r[2 + 2 * col] * 1e-9 for r in sel if r[2 + 2 * col] == r[2 + 2 * col]]
            out[name] = {"samples": len(sel), "board_power_W": {"mean": round(sum(pw) / len(pw), 1), "min": round(min(pw), 1), "max": round(max(pw), 1)},
                         "sclk_GHz": {"mean": round(sum(ck) / len(ck), 3), "min": round(min(ck), 3), "max": round(max(ck), 3)} if ck else None}
        return out


def dry_run(args, world, rank):
    """The N > 1 plumbing without a GPU: rendezvous, the gradient-sized all-reduce of trainClass.allreduce_mean_, barrier, max-over-ranks."""
    import torch
    import torch.distributed as dist
    from probav_amd.arch import layer_table
    from probav_amd.trainClass import allreduce_mean_
    be = "gloo" if args.backend == "nccl" else args.backend          # (the dry run moves CPU tensors: RCCL has nothing to say about those)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(be, rank=rank, world_size=world)
    _, total = layer_table()
    g = torch.full((total,), float(rank + 1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g.fill_(float(rank + 1))
        allreduce_mean_(g)
    if world > 1:
        dist.barrier()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    ok = bool(torch.allclose(g, torch.full_like(g, (world + 1) / 2.0)))
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "world_size": dist.get_world_size() if world > 1 else 1, "backend": be,
                          "steps": args.steps, "allreduce_floats": total, "allreduce_ok": ok, "ms_per_step": round(float(tmax) / max(1, args.steps) * 1e3, 4)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if args.dry_run:
        return dry_run(args, world, rank)
    if args.backend != "nccl":
        raise SystemExit("--backend %s: the measured path runs on HIP devices over RCCL (backend nccl); gloo is for --dry-run" % args.backend)

    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    # the library is built BEFORE the process group exists (a cold hipcc build takes minutes: inside the group it would sit in the other
    # ranks' collective watchdog window); build() serialises the ranks on a file lock, the first one in compiles, the others find it done
    ge.build()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dp = bool(args.force_dp) or os.environ.get("PROBAV_FORCE_DP") == "1"
    dp = world > 1 or force_dp
    if dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ["PROBAV_FORCE_DP"] = "1"
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dist.barrier()

    from probav_amd import _lib, synth, testClass
    from probav_amd.loss import Losses
    from probav_amd.modelsTF import WDSRConv3D
    from probav_amd.trainClass import allreduce_mean_, make_optimizer

    B = args.batch
    losses = Losses(targetShape=(48, 48, 1))
    L = _lib.lib()

    def sync():
        if dp:
            dist.barrier()
        torch.cuda.synchronize()

    def make(T):
        model = WDSRConv3D("bench", "NIR", synth.NIR_MEAN, synth.NIR_STD, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, T, 16, True)
        model.load_variables(synth.synth_params(seed=1234, numImgLR=T))            # same random-init weights on every rank
        model = model.to(dev)
        model.set_impl(args.impl)
        data = tuple(torch.as_tensor(a).to(dev) for a in synth.synth_batch(B, seed=1234 + rank, numImgLR=T))
        return model, data

    def stepper(model, data, opt=None):
        x, hr, mask = data

        seed = [None]

        def step(full=False, comm_events=None):
            pred = model(x, training=True)
            loss = losses.shiftCompensatedL1Loss(hr, mask, pred)
            model.flat.grad = None
            if seed[0] is None:
                seed[0] = torch.ones_like(loss)                # d loss / d loss, as in trainClass.trainStep: one fill launch per step less
            loss.backward(seed[0])
            if dp:
                if comm_events is not None:                # (diagnostic leg only: the headline region carries no events inside a step)
                    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ea.record()
                allreduce_mean_(model.flat.grad)
                if comm_events is not None:
                    eb.record()
                    comm_events.append((ea, eb))
            if full:
                opt.step()
                losses.shiftCompensatedcPSNR(hr, mask, pred.detach())
            return loss
        return step

    def timed(step, k, *a):
        """Wall clock around exactly k steps, max over ranks; HIP events at every step boundary (torch's current stream IS the launch stream)."""
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
        # the host runs one step ahead of the device at most (134 launches): a generation-2 sweep of the interpreter's collector in the timed region is a
        # 10-20 ms hole in the launch stream (seen once: one 26-ms step in 100).  The trainer does the same around its loop (trainClass.fitTrainData).
        import gc
        gc.collect()
        gc_was = gc.isenabled()
        gc.disable()
        try:
            # the collector run above leaves the device idle for tens of milliseconds and its shader clock gated (sclk 95 MHz in amdgpu's deep-sleep state); a region that
            # opens on a gated device pays the ramp in its first step (+1.2 ms at batch 128).  --spin-up untimed steps stand between the collector and the opening synchronize
            # (untimed warm-up like the W before them; the timed region stays exactly k steps between two synchronizes)
            for _ in range(args.spin_up):
                step(*a)
            sync()
            t0 = time.perf_counter()
            evs[0].record()
            for i in range(k):
                loss = step(*a)
                evs[i + 1].record()
            sync()
            dt = time.perf_counter() - t0
        finally:
            if gc_was:
                gc.enable()
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if dp:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax), [evs[i].elapsed_time(evs[i + 1]) for i in range(k)], loss

    T = args.frames
    model, data = make(T)
    opt = make_optimizer("nadam", model, 5e-4)
    step = stepper(model, data, opt)
    h = model._handle()
    side_probe = None
    if args.side_stream_mode >= 0:
        _lib.check(L.probav_engine_side_stream(h, args.side_stream_mode), "probav_engine_side_stream")
    sampler, windows = None, {}
    if rank == 0 and not args.no_power:
        props = torch.cuda.get_device_properties(dev)
        try:
            pci = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
        except AttributeError:
            pci = None
        sampler = PowerSampler(pci)
        if not sampler.start():
            sampler = None
    for _ in range(args.warmup):
        step()
    sync()
    if args.side_stream_mode < 0:       # untimed, behind the W warm-up steps: the trainer does the same at the start of training (trainClass.ModelTrainer)
        args.side_stream_mode, side_probe = model.tune_side_stream(step)
        side_probe = {str(k): round(v, 4) for k, v in side_probe.items()}
        for _ in range(3):              # the probe ends by switching modes: the first step in the kept mode (new fork / join pattern, slabs of the other
            step()                      # mode's layout cold) is not a step of the steady state -- three untimed ones stand between it and the timed region
        sync()
    use_events = not args.no_kernel_events
    n = len(CLASSES)

    def read_profile():
        ms, macs, cnt = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_int64 * n)()
        _lib.check(L.probav_engine_profile_read(h, n, ms, macs, cnt), "probav_engine_profile_read")
        _lib.check(L.probav_engine_profile(h, 0, 0))
        return {c: {"ms": ms[i], "macs": macs[i], "launches": int(cnt[i])} for i, c in enumerate(CLASSES)}

    # THE timed region: exactly K steps, nothing else on the stream but the K + 1 step-boundary events.
    w0 = time.time()
    dt, step_ms, loss = timed(step, args.steps)
    windows["headline_timed_region"] = (w0, time.time())
    digest = None
    if args.digest:
        import hashlib
        hsh = hashlib.sha256()
        hsh.update(loss.detach().cpu().numpy().tobytes())
        hsh.update(model.flat.grad.detach().cpu().numpy().tobytes())
        digest = hsh.hexdigest()
    # Data-parallel self-diagnosis (N > 1, or --force-dp), behind the headline region: one run yields the curve's inputs AND what explains them --
    # every rank's own median step, the gradient all-reduce's own time per step (HIP events around the collective on the launch stream: its
    # wait for the slowest rank is inside), and a proof that the replicas hold the same bits: the all-reduced gradient and the parameters
    # digest identically on every rank (reference semantics: debug/trainClassMultiGPU0.py:67-84,153,162-178 -- one replica per GPU, the
    # gradients reduced once per step, the variables mirrored).
    dp_diag = None
    if dp:
        comm = []
        kd = min(10, args.steps)
        _, ms_d, _ = timed(step, kd, False, comm)
        torch.cuda.synchronize()
        comm_ms = [a.elapsed_time(b) for a, b in comm]

        def digest64(t):                                   # a position-weighted 64-bit fold of the bit patterns, computed on the device
            v = t.detach().contiguous().view(torch.int32).to(torch.int64)
            w = torch.arange(1, v.numel() + 1, device=v.device, dtype=torch.int64)
            return int(((v * (w % 65521 + 1)).sum()).item())
        mine = torch.tensor([sorted(step_ms)[len(step_ms) // 2], sorted(comm_ms)[len(comm_ms) // 2], float(digest64(model.flat) % (1 << 52)),
                             float(digest64(model.flat.grad) % (1 << 52))], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rows = [[float(v) for v in r.tolist()] for r in allr]
        same_params = all(r[2] == rows[0][2] for r in rows)
        same_grads = all(r[3] == rows[0][3] for r in rows)
        if not (same_params and same_grads):
            raise RuntimeError("data-parallel replicas diverged: parameter digests equal: %s, all-reduced gradient digests equal: %s" % (same_params, same_grads))
        dp_diag = {"world_size": dist.get_world_size(), "backend": "nccl (RCCL)", "per_rank_step_ms_median": [round(r[0], 4) for r in rows],
                   "allreduce_ms_per_step_median_per_rank": [round(r[1], 4) for r in rows],
                   "allreduce_share_of_step": round(max(r[1] for r in rows) / max(r[0] for r in rows), 4),
                   "allreduce_payload_bytes": int(model.flat.numel() * 4), "replicas_bitwise_identical": True,
                   "note": "a short run of steps behind the headline region; the all-reduce's time is HIP events around the collective on the launch "
                           "stream and includes its wait for the slowest rank; digests: parameters and the all-reduced gradient, compared over all ranks"}
    # Roofline leg, right behind it: the same steps with the engine's HIP events around kernel launches.  Events around EVERY launch cost
    # ~6 % of the step (launch ramps no longer overlap), so two steps bracketed in full give the per-class table and name the dominant
    # class, and a second timed run of steps brackets only that class's launches (its average launch time is the roofline's `achieved`).
    prof_all, prof, dom, psteps, rsteps = None, None, None, 2, max(5, min(args.steps, 30))
    if use_events:
        # per-kernel timings are taken with the backward-filter kernels back on the caller's stream (side-stream mode 1): in the headline
        # configuration (mode 2) they run on the low-priority side stream BESIDE the other kernels, whose events would then time both
        _lib.check(L.probav_engine_side_stream(h, min(1, args.side_stream_mode)), "probav_engine_side_stream")
        for _ in range(2):
            step()
        _lib.check(L.probav_engine_profile_classes(h, 0xFFFFFFFF), "probav_engine_profile_classes")
        _lib.check(L.probav_engine_profile(h, 1, 512 * psteps), "probav_engine_profile")
        for _ in range(psteps):
            step()
        sync()
        prof_all = read_profile()
        dom = max((c for c in prof_all if prof_all[c]["macs"] > 0), key=lambda c: prof_all[c]["ms"])
        _lib.check(L.probav_engine_profile_classes(h, 1 << CLASSES.index(dom)), "probav_engine_profile_classes")
        _lib.check(L.probav_engine_profile(h, 1, 64 * rsteps), "probav_engine_profile")
        dtr, _, _ = timed(step, rsteps)
        prof = read_profile()                                                  # the dominant class over rsteps timed steps
        _lib.check(L.probav_engine_profile_classes(h, 0xFFFFFFFF), "probav_engine_profile_classes")
        _lib.check(L.probav_engine_side_stream(h, args.side_stream_mode), "probav_engine_side_stream")

    full = None
    if args.full_step:
        for _ in range(2):
            step(True)
        dtf, _, _ = timed(step, args.steps, True)
        full = dtf / args.steps * 1e3

    # what the matrix pipe of THIS device sustains (the boxes of a pool differ by ~8 % under fp16-MFMA load): a few milliseconds of
    # nothing but dependent fp16 MFMAs on every CU, right behind the timed steps (the device is warm)
    mfma_probe = None
    if rank == 0:
        seed = (torch.rand(128 * 8, device=dev) - 0.5).half()
        sink = torch.empty(256 * 256, device=dev)
        iters, launches = 2000, 8
        shapes = {}
        w0 = time.time()
        for shape, name in ((0, "32x32x16"), (1, "16x16x32")):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            _lib.check(L.probav_mfma_probe_shape(_lib.ptr(seed), _lib.ptr(sink), iters, 4, shape, _lib.current_stream()), "probav_mfma_probe_shape")
            e0.record()
            _lib.check(L.probav_mfma_probe_shape(_lib.ptr(seed), _lib.ptr(sink), iters, launches, shape, _lib.current_stream()), "probav_mfma_probe_shape")
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            shapes[name] = {"tflops": round(launches * 256 * 4 * iters * 16 * 32768.0 / (ms * 1e-3) / 1e12, 1), "ms": round(ms, 3)}
        windows["sustained_mfma_probe"] = (w0, time.time())
        mfma_probe = {"tflops": shapes["32x32x16"]["tflops"], "ms": shapes["32x32x16"]["ms"], "by_shape": shapes,
                      "note": "dependent fp16 MFMA chains, one wave per SIMD on every CU, random operands: the fp16 matrix rate this device "
                              "sustains (data sheet: 2 500 dense) with v_mfma_f32_32x32x16_f16 (the shape the H3 kernels issue, three per fp32 "
                              "product) and with v_mfma_f32_16x16x32_f16 (same cycles per FLOP; the chip holds a higher clock on it when the "
                              "matrix pipe alone is what limits the power)"}

    fp32_leg = None
    if args.impl >= 3 and not args.no_fp32_mfma_leg:
        model.set_impl(2)
        for _ in range(2):
            step()
        k2 = max(3, args.steps // 10)
        w0 = time.time()
        t2, _, _ = timed(step, k2)
        windows["fp32_mfma_leg"] = (w0, time.time())
        fp32_leg = {"value": round(world * B * k2 / t2, 2), "unit": "patches/s", "ms_per_step": round(t2 / k2 * 1e3, 4), "steps": k2,
                    "note": "same step on the native fp32-MFMA kernels (--impl 2), for reference"}
        model.set_impl(args.impl)
    power = None
    if sampler is not None:
        if args.steps * (dt / max(1, args.steps)) < 0.5:        # a short headline region holds few 10-ms samples: a second, untimed stretch of the same steps for the sensors
            w0 = time.time()
            timed(step, max(args.steps, int(0.8 / (dt / args.steps))))
            windows["same_steps_untimed_0.8s"] = (w0, time.time())
        power = sampler.stop(windows)

    other = None
    if world == 1 and T == 9 and not args.no_other_configs:
        other = {}
        # BASELINE.json config 3, realised as the reference's only longer-T network (T = 13, ConvReduceAndUpscalev3)
        del step, opt
        model._ws.clear()
        m13, d13 = make(13)
        m13.set_side_stream_mode(args.side_stream_mode)
        s13 = stepper(m13, d13)
        for _ in range(3):
            s13()
        k3 = max(5, args.steps // 5)
        t3, ms3, _ = timed(s13, k3)
        st3 = percentiles(ms3)
        other["config3_t13_training"] = {"value": round(B * k3 / t3, 2), "unit": "patches/s", "ms_per_step": round(t3 / k3 * 1e3, 4), "steps": k3, "step_ms": st3,
                                         # (a leg of few steps: one stalled step moves its mean by 15 % -- the median step beside it)
                                         "value_at_median_step": round(B / (st3["median"] * 1e-3), 2) if isinstance(st3, dict) and st3.get("median") else None,
                                         "workload": "numImgLR 13 (reducer v3), %d patches of [22,22,13,1], fwd + shift-L1 + bwd" % B}
        m13._ws.clear()
        del m13, d13, s13
        # BASELINE.json config 4: 32 image sets of 9 registered 128x128 frames -> 64 patches each -> forward, clip, round, 8x8 stitch to 384x384
        import numpy as np
        rng = np.random.default_rng(7)
        frames = torch.as_tensor(np.clip(rng.normal(synth.NIR_MEAN, synth.NIR_STD, (32, 9, 128, 128)), 0, 16383).astype(np.float32)).to(dev)
        inf = {}
        # reference_micro_batch_16: the drop-in default for test.py:125's batch_size=16 -- the micro-batches are coalesced into launch sets (same
        # pixels bit for bit: tests/test_gpu_parity.py::test_config4_*); launch_sets_of_16: every micro-batch launched on its own, as the reference's loop does
        for name, mb, lb, reps in (("batched_2048", 2048, None, 6), ("reference_micro_batch_16", 16, None, 6), ("launch_sets_of_16", 16, 16, 3)):
            run = lambda: testClass.resolve_images(model, testClass.unfold_frames(frames), micro_batch=mb, launch_batch=lb)
            for _ in range(2):                      # the first pass sizes the workspace pool for this batch, the second finds it warm
                run()
            torch.cuda.synchronize()
            each = []
            for _ in range(reps):
                t0 = time.perf_counter()
                img = run()
                torch.cuda.synchronize()
                each.append(time.perf_counter() - t0)
            d4 = sum(each) / reps
            inf[name] = {"micro_batch": mb, "launch_batch": lb if lb else max(mb, testClass.LAUNCH_BATCH), "images_per_s": round(32 / d4, 2), "patches_per_s": round(32 * 64 / d4, 1), "ms_per_32_images": round(d4 * 1e3, 3),
                         "ms_each": [round(e * 1e3, 2) for e in each]}
        assert img.shape == (32, 384, 384)
        other["config4_inference"] = {"workload": "32 image sets x 9 frames of 128x128 resident in HBM -> unfold to 2048 patches of [22,22,9,1] -> forward -> "
                                                  "clip[0,2^16] + round-half-even -> stitch to 32 x [384,384] (test.py path)", **inf}
        if not args.no_trainer_loop:
            # The reference's own entry point (train.py -> ModelTrainer.fitTrainData, models/trainClass.py:61-122): host arrays -> shuffle / repeat / batch -> pinned
            # prefetch -> trainStep (forward, shift-L1, backward, Nadam fused with the next step's weight norm, cPSNR, running means, the per-step log line) on
            # synthetic host data; beside it the bare full step (the same device work launched by this file's loop, data resident).  VERDICT r5 #6.
            import logging
            import tempfile
            from probav_amd.trainClass import ModelTrainer
            model._ws.clear()
            mt, dtr_ = make(9)
            mt.set_side_stream_mode(args.side_stream_mode)
            opt_t = make_optimizer("nadam", mt, 5e-4)
            s_full = stepper(mt, dtr_, opt_t)
            for _ in range(3):
                s_full(True)
            kf = 40
            tf_, msf, _ = timed(s_full, kf, True)
            full_med = percentiles(msf)["median"]
            nsamp = 4 * B
            xh, hh, mh = synth.synth_batch(nsamp, seed=99)
            lvl = logging.root.manager.disable
            logging.disable(logging.CRITICAL)
            try:
                with tempfile.TemporaryDirectory() as d:
                    tr = ModelTrainer(model=mt, loss=losses.shiftCompensatedL1Loss, metric=losses.shiftCompensatedcPSNR, optimizer=opt_t, ckptDir=d, logDir=d,
                                      evalStep=10 ** 9)
                    tr.tune_side_stream = False                       # (the mode is the headline run's; the trainer's own probe would spend 36 of these steps on it)
                    val = (xh[:B], hh[:B], mh[:B])
                    tr.fitTrainData(xh[:2 * B], (hh[:2 * B], mh[:2 * B]), B, 4, val)      # warm-up: 8 steps
                    evs_t, inner = [], tr.trainStep

                    def stamped(*a):
                        e = torch.cuda.Event(enable_timing=True)
                        e.record()
                        evs_t.append(e)
                        return inner(*a)
                    tr.trainStep = stamped
                    want = max(8, args.trainer_steps)
                    epochs_t = (want * B + nsamp - 1) // nsamp
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    s0 = tr.step
                    tr.fitTrainData(xh, (hh, mh), B, epochs_t, val)
                    torch.cuda.synchronize()
                    wall = time.perf_counter() - t0
                    kt = tr.step - s0
            finally:
                logging.disable(lvl)
            ms_t = [evs_t[i].elapsed_time(evs_t[i + 1]) for i in range(len(evs_t) - 1)]
            st_t = percentiles(ms_t)
            other["trainer_loop"] = {"steps": kt, "ms_per_step_median": st_t["median"], "ms_per_step_wall_mean": round(wall / kt * 1e3, 4), "step_ms": st_t,
                                     "patches_per_s_at_median_step": round(B / (st_t["median"] * 1e-3), 1),
                                     "bare_full_step_ms_median": full_med, "bare_full_step_steps": kf,
                                     "trainer_over_bare_full_step": round(st_t["median"] / full_med, 4),
                                     "workload": "ModelTrainer.fitTrainData on %d synthetic host patches (numpy), batch %d: shuffle/repeat/batch + pinned prefetch + "
                                                 "trainStep (fwd + shift-L1 + bwd + fused Nadam/weight-norm + cPSNR + running means + log line); step_ms = HIP events "
                                                 "at the head of every trainStep; bare_full_step = the same device work from bench.py's loop on resident data" % (nsamp, B)}
            mt._ws.clear()
            del mt, dtr_, s_full, opt_t

    if rank == 0:
        value = world * B * args.steps / dt
        out = {
            "metric": "LR-patches/sec fwd+bwd (WDSR-B r12 t%d, 16x16, bs%d)" % (T, B),
            "value": round(value, 2), "unit": "patches/s", "n_gpus": world,
            # (early in the line so that a truncated tail still shows them: the same step on the native fp32-MFMA kernels, and the headline at the MEDIAN timed step --
            #  the first step behind the synchronize that opens the timed region refills the launch pipeline and is ~1.2 ms longer than the others, so the mean of a
            #  20-step region sits ~1 % under the steady state)
            "fp32_mfma_path_value": fp32_leg["value"] if fp32_leg is not None else None,
            "value_at_median_step": round(world * B / (percentiles(step_ms)["median"] * 1e-3), 2),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPES.get(args.impl, "f32"), "data": "synthetic",
            "step_ms": dict(percentiles(step_ms), note="rank 0, HIP events on the launch stream at every step boundary of the timed steps"),
            "config": {"workload": "cfg p16t%dc85r12: %d patches/GPU of [22,22,%d,1] -> [48,48,1], 12 WDSR-B blocks, 32 filters; "
                                   "model fwd + shift-L1 loss + bwd to all parameter gradients%s" %
                                   (T, B, T, "; 1 flat-gradient all-reduce/step (RCCL)" if dp else ""),
                       "global_batch": world * B, "parallelism": "dp%d" % world,
                       "world_size": dist.get_world_size() if dp else 1, "backend": "nccl (RCCL)" if dp else None,
                       "forced_dp": bool(force_dp and world == 1),
                       "impl": IMPL_NAMES[args.impl],
                       "arithmetic": ("fp32 in, fp32 out, fp32 accumulate; every fp32 product is evaluated as three exact products of fp16 piece pairs "
                                      "(a = a0 + a1 to 2^-24, power-of-two operand scaling) on the fp16 MFMA pipe (H3 kernels, error of the order of "
                                      "fp32 rounding: see tests/test_gpu_parity.py)" if args.impl == 4 else
                                      "fp32 in, fp32 out, fp32 accumulate; every fp32 product is evaluated as six exact bf16-piece products on the "
                                      "bf16 MFMA pipe (x6 kernels, error of the order of fp32 rounding: see tests/test_gpu_parity.py)" if args.impl == 3
                                      else "native fp32 MFMA / VALU"),
                       "loss": float(loss.detach()), "kernel_events": use_events,
                       "side_stream_mode": args.side_stream_mode,
                       "side_stream_probe_ms": side_probe},      # None: the mode was given; else {mode: median ms per step} of WDSRModel.tune_side_stream
            "algorithmic_tflops_whole_step": round(value * ALGO_GFLOP_PER_PATCH / 1e3, 3) if T == 9 else None,
            "reference_derived": {"value": 215, "unit": "patches/s", "hardware": "GTX 1080 Ti",
                                  "note": "derived from the reference's TensorBoard logs (BASELINE.md), not a published figure"},
        }
        if dp_diag is not None:
            out["dp"] = dp_diag
        hbm_profile = hbm_profile_path()
        if T == 9:
            # the HBM view SURVEY.md §8d asks for next to the compute roofline: the un-fused layer-boundary byte model, the bytes the
            # fused plan declares, and (when profiles/ holds the PMC passes of this workload) the bytes rocprofv3 counted
            gbps = lambda mb: value / world * mb / 1e3
            hv = {"peak_GBps": PEAK_HBM_GBPS, "per_gpu": True,
                  "layer_boundary_model": {"MB_per_patch": ALGO_MB_PER_PATCH, "GBps": round(gbps(ALGO_MB_PER_PATCH), 1),
                                           "frac": round(gbps(ALGO_MB_PER_PATCH) / PEAK_HBM_GBPS, 4)},
                  "fused_plan": {"MB_per_patch": PLAN_MB_PER_PATCH, "GBps": round(gbps(PLAN_MB_PER_PATCH), 1),
                                 "frac": round(gbps(PLAN_MB_PER_PATCH) / PEAK_HBM_GBPS, 4)}}
            if hbm_profile and B == 128 and args.impl == 4:
                with open(hbm_profile) as fh:
                    bps = json.load(fh)["bytes_per_step"]
                hv["counted_by_rocprof"] = {"GB_per_step": round(bps / 1e9, 2), "GBps": round(bps / 1e9 / (dt / args.steps), 1),
                                            "source": os.path.relpath(hbm_profile, ROOT)}
            out["hbm_view"] = hv
        if digest is not None:
            out["digest"] = digest
        if mfma_probe is not None:
            out["sustained_mfma"] = mfma_probe
            # the pool's boxes differ by several percent in what their matrix pipe sustains: the headline per sustained TFLOP/s compares builds across boxes
            out["normalised_value"] = {"value": round(value / world / mfma_probe["tflops"], 4), "unit": "patches/s per sustained fp16-MFMA TFLOP/s (32x32x16), per GPU"}
        if power is not None:
            out["power"] = power
        if fp32_leg is not None:
            out["fp32_mfma_path"] = fp32_leg
            # the pool's boxes fall into two classes under fp16-MFMA load (~8 % apart on the H3 kernels, < 1 % apart on the fp32-MFMA path):
            # the ratio of the two step times tells a reader which class produced this line
            ratio = (dt / args.steps * 1e3) / fp32_leg["ms_per_step"]
            out["box_class"] = {"h3_step_over_fp32_step": round(ratio, 4), "class": "faster" if ratio < 0.405 else "slower",
                                "note": "same build on both classes; docs/notebook_r1-r5.md section 5 lists the per-class numbers"}
        if full is not None:
            out["full_step"] = {"ms_per_step": round(full, 4), "patches_per_s": round(world * B / full * 1e3, 2),
                                "includes": "fwd + L1 loss + bwd + Nadam update + cPSNR metric"}
        if other:
            out["other_configs"] = other
        if prof:
            per = {c: {"ms_per_step": round(v["ms"] / psteps, 4), "launches_per_step": v["launches"] / psteps,
                       "tflops": round(2 * v["macs"] / (v["ms"] * 1e-3) / 1e12, 3) if v["ms"] > 0 and v["macs"] > 0 else None}
                   for c, v in prof_all.items()}
            ach = 2 * prof[dom]["macs"] / (prof[dom]["ms"] * 1e-3) / 1e12
            x6 = dom.endswith("_x6")              # a split-operand class (x6 or H3 kernels, by --impl)
            nprod = SPLIT_PRODUCTS.get(args.impl, 6)
            peak = PEAK_BF16_TFLOPS / nprod if x6 else PEAK_F32_TFLOPS
            out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
                               "frac": round(ach / peak, 4), "traffic": None,
                               "avg_launch_ms": round(prof[dom]["ms"] / max(1, prof[dom]["launches"]), 4),
                               "algorithmic_gflop_per_launch": round(2 * prof[dom]["macs"] / max(1, prof[dom]["launches"]) / 1e9, 3),
                               "timed_steps": rsteps, "ms_per_step_while_bracketed": round(dtr / rsteps * 1e3, 4),
                               "note": "rank 0, HIP events on the launch stream around every launch of this class during a second timed run of steps "
                                       "right behind the headline one (which carries no kernel events; the other classes are bracketed in two "
                                       "further steps: kernel_classes); during this leg the backward-filter kernels run on the launch stream "
                                       "(probav_engine_side_stream mode 1) -- in the headline run they sit on the engine's low-priority side stream "
                                       "beside this class's launches, and events around those would time both (ms_per_step_while_bracketed shows what "
                                       "that concurrency is worth); "
                                       "achieved = SURVEY.md §8d's algorithmic fp32 FLOP of the class / its time" + (
                                           "; this class runs split-operand kernels, which issue %d 16-bit MFMA products per fp32 product, so its "
                                           "ceiling is the dense bf16/fp16 MFMA peak (%.0f TFLOP/s) / %d" % (nprod, PEAK_BF16_TFLOPS, nprod) if x6 else
                                           "; peak = dense fp32 MFMA")}
            if dom.startswith("conv1x1x1_bwd_data"):
                out["roofline"]["issued_incl_recompute_tflops"] = round(ach * PW_BWD_ISSUED_MAC / PW_BWD_ALGO_MAC, 3)
                out["roofline"]["recompute_note"] = ("the fused pointwise backward also recomputes the 256-channel hidden tile (8 192 MAC/voxel on top of the "
                                                     "29 184 algorithmic ones); the recompute is NOT counted in `achieved`")
            out["kernel_classes_note"] = ("HIP events around every launch of %d untimed steps after the warm-up, backward-filter kernels on the launch "
                                          "stream (side-stream mode 1); launches on the side stream (slab sums, residual path) are not bracketed" % psteps)
            out["kernel_classes"] = per
            if hbm_profile and T == 9 and B == 128 and args.impl == 4:
                with open(hbm_profile) as fh:
                    hp = json.load(fh)
                if dom in hp.get("per_class", {}):
                    out["roofline"]["traffic"] = hp["per_class"][dom]["bytes_per_launch"]
                    out["roofline"]["traffic_note"] = ("HBM bytes per launch of %s from %s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                                       "passes of this command; FETCH_SIZE doubled per the gfx950 note), not collected in this run"
                                                       % (hp["per_class"][dom]["kernel"], os.path.relpath(hbm_profile, ROOT)))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dp:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not under_launcher:
        return launch_ranks(args, argv)                # parent: no GPU call, no torch import; the ranks are child processes
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
