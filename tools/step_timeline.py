#!/usr/bin/env python3
"""One step of a rocprofv3 --kernel-trace of bench.py as a timeline (start, end, duration in us relative to the step's first kernel; queue; kernel): which kernels run beside
which on the engine's two streams.      python tools/step_timeline.py gpurun_out/<trace dir>  >  profiles/<tag>_step_timeline.txt"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'head_kernel' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
qs = sorted({r['Queue_Id'] for r in rows[a:b]})
print('# last whole step of the trace: %d launches, %.1f us from head_kernel to the next step\'s head_kernel; queue %s = the launch stream, the other = the engine\'s side stream' % (b - a, (int(rows[b]['Start_Timestamp']) - t0) / 1e3, rows[a]['Queue_Id']))
print('#   start      end      us  queue  kernel')
for r in rows[a:b]:
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    n = re.sub(r'^void ', '', r['Kernel_Name']).replace('probav::', '').split('(')[0]
    print('%9.1f %8.1f %7.1f  q%s  %s' % (s, e, e - s, r['Queue_Id'], n[:70]))
