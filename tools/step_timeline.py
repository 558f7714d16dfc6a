#!/usr/bin/env python3
"""One training step on the rocprofv3 time line: every launch of the LAST timed step in start order with its offset from the step's
first launch, its duration, the stream-agnostic idle gap in front of it (no kernel running on the device) and what overlaps it.
    python tools/step_timeline.py [gpurun_out/qt/trace]   (after tools/quick_trace.sh)
"""
import csv
import glob
import sys

root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/qt/trace'
f = glob.glob(root + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
nm = lambda r: r['Kernel_Name'].replace('void ', '')
adam = [i for i, r in enumerate(rows) if nm(r).startswith('adam_segs')]     # one Adam launch per step since round 6
ends = adam
if not adam:    # older libraries: a step ends with three adam_step launches
    adam = [i for i, r in enumerate(rows) if nm(r).startswith('adam_step')]
    ends = adam[2::3]
# take the step before the last one (the last may be followed by leg teardown)
lo, hi = ends[-3] + 1, ends[-2] + 1
step = rows[lo:hi]
t0 = int(step[0]['Start_Timestamp'])
busy_end = t0
idle = 0.0
print(f"# {len(step)} launches, span {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
print("# offset_us  dur_us  idle_before_us  queue  kernel")
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = max(0, s - busy_end) / 1e3
    idle += gap
    k = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')
    k = k[:k.find('(')] if '(' in k else k
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f}  q{r.get('Queue_Id', '?'):>3}  {k[:120]}")
    busy_end = max(busy_end, e)
print(f"# device idle inside the step: {idle:.1f} us")
