#!/usr/bin/env python3
"""Where does the GPU idle?  From the rocprofv3 CSV traces of one run (kernel_trace.csv + hip_api_trace.csv): every interval > MIN_US between the end
of one kernel and the start of the next is attributed to the kernel that FOLLOWS it and to the blocking HIP call (hipMemcpy / hipStreamSynchronize /
hipDeviceSynchronize / hipMalloc / hipFree) the host thread last returned from before that kernel was launched.
usage: gpu_gaps.py DIR [MIN_US]"""
import collections
import csv
import glob
import os
import re
import sys

d = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
kf = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
af = glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True)[0]
ks = []
for r in csv.DictReader(open(kf)):
    m = re.search(r"(k_[A-Za-z0-9_]+(?:<\d)?|qk_[a-z_]+|customBcKernel|__amd_rocclr_\w+)", r["Kernel_Name"])
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:40], int(r.get("Correlation_Id", 0) or 0)))
ks.sort()
# the evolve: from the first to the last marching sweep (start-up allocations and initial conditions left out)
marches = [i for i, k in enumerate(ks) if "k_sweep_march" in k[2]]
if marches:
    ks = ks[marches[0]:marches[-1] + 1]
api = []
for r in csv.DictReader(open(af)):
    api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], int(r.get("Correlation_Id", 0) or 0)))
api.sort()
blocking = [a for a in api if a[2] in ("hipMemcpy", "hipStreamSynchronize", "hipDeviceSynchronize", "hipMalloc", "hipFree", "hipHostMalloc", "hipMemcpyAsync")]
launch_of = {a[3]: a for a in api if a[2].startswith("hipLaunchKernel") or a[2].startswith("hipModuleLaunch") or a[2].startswith("hipExtLaunch")}
gaps = collections.Counter()
gapn = collections.Counter()
total_gap = 0
import bisect
bstarts = [b[1] for b in blocking]
t_first, t_last = ks[0][0], max(k[1] for k in ks)
busy_end = ks[0][1]
for (s, e, name, cid) in ks[1:]:
    if s > busy_end:
        g = (s - busy_end) / 1e3
        if g >= min_us:
            la = launch_of.get(cid)
            cause = "?"
            if la is not None:
                i = bisect.bisect_right(bstarts, la[0]) - 1
                if i >= 0 and la[0] - blocking[i][1] < 200e3:  # the last blocking call returned < 200 us before this launch
                    cause = blocking[i][2]
                else:
                    cause = "host busy (no blocking call)"
            key = (cause, name)
            gaps[key] += g
            gapn[key] += 1
            total_gap += g
    busy_end = max(busy_end, e)
print(f"kernels {len(ks)}, span {(t_last - t_first) / 1e6:.1f} ms, idle in gaps >= {min_us} us: {total_gap / 1e3:.1f} ms")
for (cause, name), g in gaps.most_common(30):
    print(f"{g / 1e3:8.2f} ms  {gapn[(cause, name)]:5d} x  {cause:32s} -> {name}")
bycause = collections.Counter()
for (cause, name), g in gaps.items():
    bycause[cause] += g
print("by cause:", {k: round(v / 1e3, 1) for k, v in bycause.most_common()})
