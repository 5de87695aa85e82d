import os, sys, re, numpy as np, pathlib, tempfile
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import test_host_mirror_gpu as T
from quokka_amd.simulation import chop_domain, distribute_boxes
N, mgs, nranks, steps, bf = (int(os.environ.get(k, d)) for k, d in (("N", 64), ("MGS", 16), ("RANKS", 8), ("STEPS", 30), ("BF", 8)))
args = ["geometry.prob_lo=0 0 0", "geometry.prob_hi=1.2 1.2 1.2", "geometry.is_periodic=0 0 0", f"amr.n_cell={N} {N} {N}", f"amr.max_grid_size={mgs}",
        "amr.max_level=2", f"amr.blocking_factor={bf}", "amr.n_error_buf=3", "do_reflux=1", f"max_timesteps={steps}", f"qk.refine_grid_layout_target={nranks}"]
tmp = pathlib.Path(tempfile.mkdtemp())
for s in "ab": os.makedirs(tmp / s)
(one,), o1 = T.run_ranks("ref_HydroBlast3D", args, tmp / "a", 1, 0)
parts, on = T.run_ranks("ref_HydroBlast3D", args + ["qk.level0_distribution=bricks"], tmp / "b", nranks, 0)
zone = re.compile(r"Zone-updates on level (\d): (\d+) \((\d+) grids\)")
print(zone.findall(o1[0])); print(zone.findall(on[0]))
print(re.findall(r"Boxes of level \d per rank:.*", on[0]))
boxes = chop_domain([N] * 3, [mgs] * 3); owner = distribute_boxes(boxes, nranks, [N] * 3, [mgs] * 3)
many = T._level0_state_by_box(parts, owner, len(boxes), mgs ** 3, mgs)
one = one.reshape(many.shape)
print([float(np.abs(many[:, n] - one[:, n]).max() / np.abs(one[:, n]).max()) for n in range(6)])
print([l for l in on[0].split("\n") if "conservation is" in l or "figure-of-merit" in l])
