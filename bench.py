#!/usr/bin/env python3
"""bench.py -- LR-patches/sec, fwd+bwd, WDSR-B r12 t9, 16x16 patches (+6 px border), batch 128 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic patches already resident in HBM:
model forward (weight-norm, head, 12 WDSR-B blocks, reducers, pixel shuffle), shift-compensated L1 loss
forward, and the full backward to all 132 parameter gradients (BASELINE.json metric) --
plus, for N > 1, the one gradient all-reduce that data parallelism adds (RCCL over xGMI).
The optimizer update and the cPSNR metric are not part of "fwd+bwd"; `--full-step` times them too and
reports the result under "full_step" without changing `value`.

Rank 0 prints ONE JSON line.  `roofline` is measured live: the engine brackets every kernel launch of
the timed steps with HIP events on the launch stream and reports per-class time and algorithmic MACs;
the dominant class is priced against the fp32 MFMA/VALU peak of MI355X (157.3 TFLOP/s).
`cpu_baseline` times the CPU oracle (torch-CPU restatement of the TF path; the TF reference itself
cannot run here) on the host cores at the reference's CPU-runnable config (batch 8).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CLASSES = ["weight_norm", "small", "conv3x3x3_fwd", "conv3x3x3_bwd_data", "conv3x3x3_wgrad",
           "conv1x1x1_fwd", "conv1x1x1_bwd_data", "conv1x1x1_wgrad",
           "conv3x3x3_fwd_x6", "conv3x3x3_bwd_data_x6", "conv3x3x3_wgrad_x6", "conv1x1x1_fwd_x6", "conv1x1x1_bwd_data_x6"]
PEAK_F32_TFLOPS = 157.3            # MI355X_MICROARCH.md: fp32 vector == fp32 MFMA peak
PEAK_BF16_TFLOPS = 2500.0          # MI355X_MICROARCH.md: dense bf16 MFMA peak (~2.5 PF)
SPLIT_PRODUCTS = {3: 6, 4: 3}      # 16-bit MFMA products issued per fp32 product: x6 kernels (bf16 pieces) / H3 kernels (scaled fp16 pieces)
PEAK_HBM_GBPS = 8000.0              # MI355X_MICROARCH.md: HBM3E peak
ALGO_MB_PER_PATCH = 402.414        # layer-boundary byte model, fwd + bwd (un-fused)
PLAN_MB_PER_PATCH = 81.0           # bytes the fused plan moves (256-channel tensor never reaches HBM)
HBM_PROFILE = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")   # rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this command
ALGO_GFLOP_PER_PATCH = 12.436      # SURVEY.md §8d / BASELINE.md §2: fwd + bwd, p16t9c85r12


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
    return {"value": round(8 * n / dt, 3), "unit": "patches/s", "cores": cores, "kind": "port",
            "sample": "oracle/wdsr_torch.py fp32, batch 8 (cfg p16t9c85r12 on CPU), fwd + L1 loss + bwd, %d steps in %.1f s" % (n, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="patches per GPU (BASELINE.json: 128)")
    ap.add_argument("--frames", type=int, default=9, help="numImgLR: 9 (headline), 13 or 7")
    ap.add_argument("--impl", type=int, default=4, help="4 = H3 kernels: fp32 products as three products of scaled fp16 piece pairs on the 16-bit MFMA pipe (default), "
                    "3 = x6 kernels: six bf16-piece products, "
                    "2 = native fp32 MFMA + strip convolution, 1 = fp32 MFMA row-tile kernels, 0 = generic direct kernels")
    ap.add_argument("--no-fp32-mfma-leg", action="store_true", help="skip the short extra run on the native fp32-MFMA kernels (impl 2)")
    ap.add_argument("--full-step", action="store_true", help="also time loss+metric+Nadam (reported separately)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
        args.gpus = world
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()

    from probav_amd import _lib, synth
    from probav_amd.loss import Losses
    from probav_amd.modelsTF import WDSRConv3D
    from probav_amd.trainClass import allreduce_mean_, make_optimizer

    T, B = args.frames, args.batch
    model = WDSRConv3D("bench", "NIR", synth.NIR_MEAN, synth.NIR_STD, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, T, 16, True)
    model.load_variables(synth.synth_params(seed=1234, numImgLR=T))            # same random-init weights on every rank
    model = model.to(dev)
    model.set_impl(args.impl)
    losses = Losses(targetShape=(48, 48, 1))
    x, hr, mask = (torch.as_tensor(a).to(dev) for a in synth.synth_batch(B, seed=1234 + rank, numImgLR=T))
    opt = make_optimizer("nadam", model, 5e-4)
    L, h = _lib.lib(), model._handle()

    def step(full=False):
        pred = model(x, training=True)
        loss = losses.shiftCompensatedL1Loss(hr, mask, pred)
        model.flat.grad = None
        loss.backward()
        if world > 1:
            allreduce_mean_(model.flat.grad)
        if full:
            opt.step()
            losses.shiftCompensatedcPSNR(hr, mask, pred.detach())
        return loss

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    use_events = not args.no_kernel_events
    n = len(CLASSES)

    def read_profile():
        ms, macs, cnt = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_int64 * n)()
        _lib.check(L.probav_engine_profile_read(h, n, ms, macs, cnt), "probav_engine_profile_read")
        _lib.check(L.probav_engine_profile(h, 0, 0))
        return {c: {"ms": ms[i], "macs": macs[i], "launches": int(cnt[i])} for i, c in enumerate(CLASSES)}

    # HIP events around EVERY launch cost ~6 % of the step (launch ramps no longer overlap), so the timed region brackets only the
    # launches of the dominant kernel class; which class that is, and the per-class table, come from two untimed steps bracketed in full.
    prof_all, dom = None, None
    if use_events:
        psteps = 2
        _lib.check(L.probav_engine_profile_classes(h, 0xFFFFFFFF), "probav_engine_profile_classes")
        _lib.check(L.probav_engine_profile(h, 1, 512 * psteps), "probav_engine_profile")
        for _ in range(psteps):
            step()
        sync()
        prof_all = read_profile()
        dom = max((c for c in prof_all if prof_all[c]["macs"] > 0), key=lambda c: prof_all[c]["ms"])
        _lib.check(L.probav_engine_profile_classes(h, 1 << CLASSES.index(dom)), "probav_engine_profile_classes")
        _lib.check(L.probav_engine_profile(h, 1, 64 * args.steps), "probav_engine_profile")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    dt = time.perf_counter() - t0
    prof = None
    if use_events:
        prof = read_profile()                                                  # the dominant class over the timed steps
        _lib.check(L.probav_engine_profile_classes(h, 0xFFFFFFFF), "probav_engine_profile_classes")
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    full = None
    if args.full_step:
        for _ in range(2):
            step(True)
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(True)
        sync()
        full = (time.perf_counter() - t1) / args.steps * 1e3

    fp32_leg = None
    if args.impl >= 3 and not args.no_fp32_mfma_leg:
        model.set_impl(2)
        for _ in range(2):
            step()
        sync()
        k2 = max(3, args.steps // 4)
        t1 = time.perf_counter()
        for _ in range(k2):
            step()
        sync()
        t2 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        fp32_leg = {"value": round(world * B * k2 / float(t2), 2), "unit": "patches/s", "ms_per_step": round(float(t2) / k2 * 1e3, 4), "steps": k2,
                    "note": "same step on the native fp32-MFMA kernels (--impl 2), for reference"}
        model.set_impl(args.impl)

    if rank == 0:
        value = world * B * args.steps / dt
        out = {
            "metric": "LR-patches/sec fwd+bwd (WDSR-B r12 t%d, 16x16, bs%d)" % (T, B),
            "value": round(value, 2), "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg p16t%dc85r12: %d patches/GPU of [22,22,%d,1] -> [48,48,1], 12 WDSR-B blocks, 32 filters; "
                                   "model fwd + shift-L1 loss + bwd to all parameter gradients%s" %
                                   (T, B, T, "; 1 flat-gradient all-reduce/step (RCCL)" if world > 1 else ""),
                       "global_batch": world * B, "parallelism": "dp%d" % world, "impl": {0: "direct", 1: "mfma-rowtile", 2: "mfma-strip", 3: "x6-split-bf16", 4: "h3-split-fp16"}[args.impl],
                       "arithmetic": ("fp32 in, fp32 out, fp32 accumulate; every fp32 product is evaluated as three exact products of fp16 piece pairs "
                                      "(a = a0 + a1 to 2^-24, per-tensor power-of-two scaling) on the fp16 MFMA pipe (H3 kernels, error of the order of "
                                      "fp32 rounding: see tests/test_gpu_parity.py)" if args.impl == 4 else
                                      "fp32 in, fp32 out, fp32 accumulate; every fp32 product is evaluated as six exact bf16-piece products on the "
                                      "bf16 MFMA pipe (x6 kernels, error of the order of fp32 rounding: see tests/test_gpu_parity.py)" if args.impl == 3
                                      else "native fp32 MFMA / VALU"),
                       "loss": float(loss.detach()), "kernel_events": use_events},
            "algorithmic_tflops_whole_step": round(value * ALGO_GFLOP_PER_PATCH / 1e3, 3) if T == 9 else None,
            "reference_derived": {"value": 215, "unit": "patches/s", "hardware": "GTX 1080 Ti",
                                  "note": "derived from the reference's TensorBoard logs (BASELINE.md), not a published figure"},
        }
        if T == 9:
            # the HBM view SURVEY.md §8d asks for next to the compute roofline: the un-fused layer-boundary byte model, the bytes the
            # fused plan declares, and (when profiles/ holds the PMC passes of this workload) the bytes rocprofv3 counted
            gbps = lambda mb: value * mb / 1e3
            hv = {"peak_GBps": PEAK_HBM_GBPS,
                  "layer_boundary_model": {"MB_per_patch": ALGO_MB_PER_PATCH, "GBps": round(gbps(ALGO_MB_PER_PATCH), 1),
                                           "frac": round(gbps(ALGO_MB_PER_PATCH) / PEAK_HBM_GBPS, 4)},
                  "fused_plan": {"MB_per_patch": PLAN_MB_PER_PATCH, "GBps": round(gbps(PLAN_MB_PER_PATCH), 1),
                                 "frac": round(gbps(PLAN_MB_PER_PATCH) / PEAK_HBM_GBPS, 4)}}
            if os.path.exists(HBM_PROFILE) and B == 128 and args.impl == 4:
                with open(HBM_PROFILE) as fh:
                    bps = json.load(fh)["bytes_per_step"]
                hv["counted_by_rocprof"] = {"GB_per_step": round(bps / 1e9, 2), "GBps": round(bps / 1e9 / (dt / args.steps) / world, 1) if world == 1 else None,
                                            "source": os.path.relpath(HBM_PROFILE, ROOT)}
            out["hbm_view"] = hv
        if fp32_leg is not None:
            out["fp32_mfma_path"] = fp32_leg
        if full is not None:
            out["full_step"] = {"ms_per_step": round(full, 4), "patches_per_s": round(world * B / full * 1e3, 2),
                                "includes": "fwd + L1 loss + bwd + Nadam update + cPSNR metric"}
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
                               "note": "rank 0, HIP events on the launch stream around every launch of this class during the timed steps "
                                       "(the other classes are bracketed only in two untimed steps: kernel_classes); "
                                       "achieved = algorithmic fp32 FLOP/s" + (
                                           "; this class runs split-operand kernels, which issue %d 16-bit MFMA products per fp32 product, so its "
                                           "ceiling is the dense bf16/fp16 MFMA peak (%.0f TFLOP/s) / %d" % (nprod, PEAK_BF16_TFLOPS, nprod) if x6 else
                                           "; peak = dense fp32 MFMA")}
            out["kernel_classes_note"] = "HIP events around every launch of %d untimed steps after the warm-up" % psteps
            out["kernel_classes"] = per
            if os.path.exists(HBM_PROFILE) and T == 9 and B == 128 and args.impl == 4:
                with open(HBM_PROFILE) as fh:
                    hp = json.load(fh)
                if dom in hp.get("per_class", {}):
                    out["roofline"]["traffic"] = hp["per_class"][dom]["bytes_per_launch"]
                    out["roofline"]["traffic_note"] = ("HBM bytes per launch of %s from %s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                                       "passes of this command; FETCH_SIZE doubled per the gfx950 note), not collected in this run"
                                                       % (hp["per_class"][dom]["kernel"], os.path.relpath(HBM_PROFILE, ROOT)))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
