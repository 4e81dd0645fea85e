#!/usr/bin/env python3
"""Diagnostic (not part of the product): the timeline of ONE training step out of a rocprofv3 --kernel-trace csv.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o p -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline ...
    python3 tools/timeline.py gpurun_out/trace/p_kernel_trace.csv [step index, default: the last complete one]

A step is the span from one `head_kernel` launch to the next.  Printed: the span, the time at least one kernel was running, the busy time
per queue, the idle gaps of the whole device (largest first, with the kernels on either side), and the launch queue's own gaps.
"""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.replace("void ", "").replace("probav::", "")
    return n.split("(")[0][:48]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"])) for r in rows), key=lambda k: k[0])
    heads = [i for i, k in enumerate(ks) if k[3].startswith("head_kernel")]
    if len(heads) < 2:
        sys.exit("fewer than two head_kernel launches in the trace")
    si = int(sys.argv[2]) if len(sys.argv) > 2 else len(heads) - 2
    step = ks[heads[si]:heads[si + 1]]
    t0, t1 = step[0][0], ks[heads[si + 1]][0]
    print("step %d: %d kernels, span %.3f ms" % (si, len(step), (t1 - t0) / 1e6))
    perq = defaultdict(list)
    for k in step:
        perq[k[2]].append(k)
    for q, lst in sorted(perq.items(), key=lambda kv: -len(kv[1])):
        print("  queue %d: %3d kernels, busy %.3f ms, first %s" % (q, len(lst), sum(e - s for s, e, _, _ in lst) / 1e6, lst[0][3]))
    # union of the intervals
    gaps, cur_end, last = [], step[0][0], step[0]
    busy = 0
    for k in step:
        if k[0] > cur_end:
            gaps.append((k[0] - cur_end, last[3], k[3], (cur_end - t0) / 1e6))
            busy += 0
        if k[1] > cur_end:
            busy += k[1] - max(k[0], cur_end)
            cur_end, last = k[1], k
    print("  at least one kernel running: %.3f ms; device idle inside the step: %.3f ms in %d gaps; after the last kernel: %.3f ms"
          % (busy / 1e6, sum(g[0] for g in gaps) / 1e6, len(gaps), (t1 - cur_end) / 1e6))
    print("  largest device-idle gaps:")
    for g in sorted(gaps, reverse=True)[:12]:
        print("    %6.1f us at %.3f ms  between %-40s and %s" % (g[0] / 1e3, g[3], g[1], g[2]))
    mainq = max(perq.items(), key=lambda kv: len(kv[1]))[0]
    lst = perq[mainq]
    qg = [(lst[i + 1][0] - lst[i][1], lst[i][3], lst[i + 1][3]) for i in range(len(lst) - 1)]
    print("  launch queue %d: sum of its own gaps %.3f ms (median %.1f us); concurrent time (two or more kernels running) %.3f ms"
          % (mainq, sum(g[0] for g in qg if g[0] > 0) / 1e6, sorted(g[0] for g in qg)[len(qg) // 2] / 1e3,
             (sum(e - s for s, e, _, _ in step) - busy) / 1e6))
    hist = defaultdict(lambda: [0, 0])
    for g in qg:
        hist[(g[1], g[2])][0] += 1
        hist[(g[1], g[2])][1] += g[0]
    print("  launch-queue gaps by kernel pair (count, total us):")
    for (a, b), (n, t) in sorted(hist.items(), key=lambda kv: -kv[1][1])[:14]:
        print("    %3d x %7.1f us  %-40s -> %s" % (n, t / 1e3, a, b))


if __name__ == "__main__" and not (len(sys.argv) > 3 and sys.argv[3] == "dump"):
    main()


def dump(path, si=None):
    """every kernel of one step in start order: queue, start (us from the step's first launch), duration (us), name, grid"""
    rows = list(csv.DictReader(open(path)))
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["LDS_Block_Size"])) for r in rows), key=lambda k: k[0])
    heads = [i for i, k in enumerate(ks) if k[3].startswith("head_kernel")]
    si = len(heads) - 2 if si is None else si
    step = ks[heads[si]:heads[si + 1]]
    for s, e, q, n, g, lds in step:
        print("q%d %9.1f %8.1f  %-46s grid %5d lds %6d" % (q, (s - step[0][0]) / 1e3, (e - s) / 1e3, n, g, lds))


if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] == "dump":
    dump(sys.argv[1], int(sys.argv[2]) if sys.argv[2] != "-" else None)
