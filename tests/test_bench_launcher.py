"""bench.py --gpus N must start its own ranks: the parent spawns `torch.distributed.run` as a child before anything
touches a GPU and forwards rank 0's JSON line.  Covered here on CPU with the gloo backend and `--dry-run` (rendezvous + the
gradient-sized all-reduce of the data-parallel step; no kernels)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    e = dict(os.environ)
    e.pop("RANK", None); e.pop("WORLD_SIZE", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_gpus2_self_launches_its_ranks():
    out = _run(["--gpus", "2", "--dry-run", "--backend", "gloo", "--steps", "3"])
    assert out["dry_run"] and out["n_gpus"] == 2 and out["world_size"] == 2 and out["allreduce_ok"]
    assert out["allreduce_floats"] == 535267


def test_bench_single_rank_dry_run_needs_no_launcher():
    out = _run(["--dry-run", "--backend", "gloo", "--steps", "1"])
    assert out["n_gpus"] == 1 and out["world_size"] == 1 and out["allreduce_ok"]


def test_bench_parent_does_not_import_torch():
    """The launching parent must stay GPU-free: it may not even import torch before the ranks exist."""
    code = ("import sys, bench\n"
            "bench.launch_ranks = lambda a, v: (print('torch' in sys.modules), 0)[1]\n"
            "sys.exit(bench.main(['--gpus', '4']))\n")
    e = dict(os.environ)
    e.pop("RANK", None); e.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=e, cwd=ROOT)
    assert p.returncode == 0 and p.stdout.strip() == "False", p.stdout + p.stderr


def test_power_sampler_reads_hwmon_files_from_a_child_process(tmp_path):
    """bench.py's board-power / shader-clock sampler: a child process that reads amdgpu's hwmon files, windows cut by wall clock.
    Here on two fake sensors: the one whose power moves is taken for the device, its samples inside the window are averaged."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    paths = []
    for k, (uw, hz) in enumerate(((255000000, 95000000), (250000000, 500000000))):
        d = tmp_path / ("hwmon%d" % k)
        d.mkdir()
        (d / "power1_input").write_text("%d\n" % uw)
        (d / "freq1_input").write_text("%d\n" % hz)
        paths.append(str(d))
    s = bench.PowerSampler(None)
    s.paths, s.pci = paths, None
    assert s.start()
    time.sleep(0.15)
    t0 = time.time()
    (tmp_path / "hwmon1" / "power1_input").write_text("1000000000\n")            # the load arrives on the second board: 1 kW at 1.6 GHz
    (tmp_path / "hwmon1" / "freq1_input").write_text("1600000000\n")
    time.sleep(0.25)
    t1 = time.time()
    out = s.stop({"run": (t0 + 0.05, t1), "nothing": (t1 + 10, t1 + 11)})
    assert out["sensor"] == paths[1] and out["nothing"] is None
    assert out["run"]["samples"] >= 5 and out["run"]["board_power_W"]["mean"] == 1000.0 and out["run"]["sclk_GHz"]["max"] == 1.6
