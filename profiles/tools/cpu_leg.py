"""The CPU leg of bench.py alone (cpu_baseline: the fused form of the oracle) at several thread counts, with the phase profile of the step
(ORACLE_PROF): python profiles/tools/cpu_leg.py [ncell] [steps] [threads ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
threads = [int(t) for t in sys.argv[3:]] or [0]
code = ("import json, sys; sys.path.insert(0, %r); import bench; print('CPULEG', json.dumps(bench.cpu_baseline(%d, %d)))" % (ROOT, ncell, steps))
for t in threads:
    env = dict(os.environ, ORACLE_PROF="1")
    if t > 0:
        env["OMP_NUM_THREADS"] = str(t)
    else:
        env.pop("OMP_NUM_THREADS", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    for line in (r.stdout + r.stderr).splitlines():
        if line.startswith("CPULEG"):
            d = json.loads(line[7:])
            print(f"threads {d['cores']:3d}: {d['value']:.3f} M cell-updates/s, {d['value_per_granted_cpu']:.3f} per granted CPU ({d['cpus_granted']:g}); {d['sample']}")
        elif "oracle prof" in line or "Error" in line or "error" in line:
            print("   ", line)
