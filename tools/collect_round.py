#!/usr/bin/env python3
"""Copy the summaries tools/measure_round.sh left under gpurun_out/round_<tag>/ into profiles/<tag>_* (tracked).

    python tools/collect_round.py r03"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", "round_" + tag)
dst = os.path.join(ROOT, "profiles")
for name, out in (("kernel_stats.csv", "bench_kernel_stats.csv"), ("hbm_traffic.json", "hbm_traffic.json"),
                  ("bench_line.json", "bench_line.json"), ("bench_line_fullstep.json", "bench_line_fullstep.json")):
    shutil.copy(os.path.join(src, name), os.path.join(dst, "%s_%s" % (tag, out)))
if os.path.exists(os.path.join(src, "step_timeline.txt")):
    shutil.copy(os.path.join(src, "step_timeline.txt"), os.path.join(dst, tag + "_step_timeline.txt"))
with open(os.path.join(dst, tag + "_pmc_summary.txt"), "w") as fh:
    fh.write("# rocprofv3 --pmc (two passes, each with --kernel-trace only) over python3 bench.py --steps 3 --warmup 1 (default --impl 4: H3 kernels), "
             "summed per kernel by tools/pmc_summary.py\n# X/WAVE = counter / SQ_WAVE_CYCLES; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs)\n## pass A\n")
    fh.write(open(os.path.join(src, "pmc_a.txt")).read())
    fh.write("## pass B\n")
    fh.write(open(os.path.join(src, "pmc_b.txt")).read())
print("collected", sorted(f for f in os.listdir(dst) if f.startswith(tag)))
