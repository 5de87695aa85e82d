#!/usr/bin/env python3
"""per-kernel register / scratch / LDS / occupancy table of one .hip file of quokka_amd/csrc (hipcc -Rpass-analysis=kernel-resource-usage, gfx950)
usage: resource_usage.py qk_hydro_fused.hip [--scratch-only]"""
import os
import re
import subprocess
import sys

here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "quokka_amd", "csrc")
f = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", *os.environ.get("QK_RU_FLAGS", "").split(), "-I../../include", "-I.",
       "-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", "/tmp/ru.o"]
out = subprocess.run(cmd, cwd=here, capture_output=True, text=True).stderr
rows, cur = [], {}
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k.split(" ")[0]] = v
for r in rows:
    if "--scratch-only" in sys.argv and r.get("ScratchSize", "0") == "0":
        continue
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    print(f"VGPR {r.get('VGPRs', '?'):>4} AGPR {r.get('AGPRs', '?'):>3} scratch {r.get('ScratchSize', '?'):>5} LDS {r.get('LDS', '?'):>6} occ {r.get('Occupancy', '?'):>2}  {name[:150]}")
