import sys, time, json, os
sys.path.insert(0, os.getcwd())
import torch
from quokka_amd.multifab import Context
from quokka_amd.simulation import sedov_problem
ctx = Context(0)
t0 = time.time()
sim = sedov_problem(ctx, 512, max_grid_size=128)
print("init s", time.time() - t0, "boxes", sim.lev.nboxes, flush=True)
def run(n):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        assert sim.step()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
run(2)
a = run(5)
print("unsplit ms/step", a, "Mcell/s", 512**3 / a / 1e3)
# emulate the 8-rank brick: boxes touching the +x/+y/+z brick faces wait for remote strips
late = [b for b, (lo, hi) in enumerate(sim.my_boxes) if hi[0] == 511 or hi[1] == 511 or hi[2] == 511]
for b in late:
    sim.ghost.set_box_remote(b, True)
g = sim.overlap_groups()
print("groups", None if g is None else (len(g[0][1]), len(g[1][1])))
run(1)
b = run(5)
print("split ms/step", b, "Mcell/s", 512**3 / b / 1e3)
print("mem GB", torch.cuda.max_memory_allocated() / 1e9)
