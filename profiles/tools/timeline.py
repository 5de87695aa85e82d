#!/usr/bin/env python3
"""Timeline of one coarse step of an AMR run from a rocprofv3 CSV kernel trace: every kernel with its start (us after the step's first kernel),
duration, queue and how many other kernels were running when it started — what overlapped what.
usage: timeline.py DIR [STEP_INDEX] [MAX_LINES]"""
import csv
import glob
import os
import re
import sys

d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 10
maxl = int(sys.argv[3]) if len(sys.argv) > 3 else 400
kf = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
ks = []
for r in csv.DictReader(open(kf)):
    name = r["Kernel_Name"]
    m = re.search(r"(k_[A-Za-z0-9_]+(?:<[0-9, a-z]+>)?|__amd_rocclr_\w+)", name)
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else name[:50], r.get("Queue_Id", "?"), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
ks.sort()
# a coarse step starts with the big level-0 x sweep of stage 1 (k_sweep_x<3, 1 ... on the largest grid)
big = max(k[4] for k in ks if "k_sweep_x" in k[2])
starts = [i for i, k in enumerate(ks) if "k_sweep_x<3, 1" in k[2] and k[4] == big]
i0, i1 = starts[which], starts[which + 1]
# back up to the ghost fill in front of that sweep
while i0 > 0 and ks[i0][0] - ks[i0 - 1][1] < 30e3 and "k_sweep" not in ks[i0 - 1][2]:
    i0 -= 1
step = ks[i0:i1]
t0 = step[0][0]
print(f"coarse step {which}: {len(step)} kernels, span {(max(k[1] for k in step) - t0) / 1e3:.1f} us, sum of durations {sum(k[1] - k[0] for k in step) / 1e3:.1f} us")
busy, end = 0.0, t0
for s, e, *_ in step:
    if e > end:
        busy += (e - max(s, end)) / 1e3
        end = e
print(f"GPU busy (union of kernel intervals) {busy:.1f} us")
for n, (s, e, name, q, g) in enumerate(step[:maxl]):
    conc = sum(1 for (s2, e2, *_r) in step if s2 < s < e2)
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} q{q:>3} conc={conc} grid={g:<9d} {name}")
