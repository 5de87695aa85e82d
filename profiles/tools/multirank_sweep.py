"""Sweep of hierarchy configurations of the C++ host on several ranks sharing one GPU: for each (problem, deck overrides, ranks, scheme) one rank building the same
grids against N ranks; the final plotfiles are compared level by level (grids equal; largest difference per variable relative to the variable's level-0 scale).
scheme "per_level": qk.distribute_levels=1 (reference: one rank with qk.refine_grid_layout_target=N); "ancestor": qk.distribute_levels=0 (reference: one rank
with qk.cluster_within_parent=1).  usage: multirank_sweep.py [name filter]"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from quokka_amd import plotfile as pf  # noqa: E402

HOST = os.path.join(ROOT, "quokka_amd", "host")
CASES = [  # name, executable, deck, overrides, ranks, coarse steps
    ("blast2d", "ref_HydroBlast2D", None, ["geometry.prob_lo=0 0 0", "geometry.prob_hi=1 1 1", "geometry.is_periodic=0 0 0", "amr.n_cell=128 128 8", "amr.max_grid_size=16",
                                           "amr.blocking_factor=8", "amr.n_error_buf=3", "do_reflux=1", "do_tracers=0", "amr.max_level=2"], 4, 10),
    ("shadow", "ref_RadShadow", "shadow.in", ["amr.max_grid_size=16"], 4, 8),
    ("beam8", "ref_RadBeam", "beam.in", ["amr.max_grid_size=8"], 4, 8),
    ("shell_amr", "ref_RadhydroShell", "radhydro_shell_amr.in", ["amr.n_cell=32 32 32", "amr.max_level=1", "amr.max_grid_size=8", "amr.blocking_factor=8", "max_timesteps=4", "plotfile_interval=2"], 4, 4),
    ("shocktube_cma", "ref_HydroShocktubeCMA", "shocktube_cma.in", ["amr.max_grid_size=32"], 2, 40),
    ("uniform_xy", "ref_HydroBlast3D", None, ["geometry.prob_lo=0 0 0", "geometry.prob_hi=1.2 1.2 1.2", "geometry.is_periodic=0 0 0", "amr.n_cell=128 128 128", "amr.max_grid_size=64",
                                             "amr.max_level=0", "hydro.rk2_carry_rhs=1", "qk.min_overlap_cells=1", "max_timesteps=10"], 4, 10),  # fused XY sweep + early / late exchange
    ("periodic3d", "ref_HydroBlast3D", None, ["geometry.prob_lo=0 0 0", "geometry.prob_hi=1.2 1.2 1.2", "geometry.is_periodic=1 1 1", "amr.n_cell=32 32 32", "amr.max_grid_size=8",
                                             "amr.max_level=2", "amr.blocking_factor=8", "amr.n_error_buf=3", "do_reflux=1"], 4, 10),  # the blast in the corner: every level wraps through three faces
    ("four_levels", "ref_HydroBlast3D", None, ["geometry.prob_lo=0 0 0", "geometry.prob_hi=1.2 1.2 1.2", "geometry.is_periodic=0 0 0", "amr.n_cell=32 32 32", "amr.max_grid_size=16",
                                              "amr.max_level=3", "amr.blocking_factor=8", "amr.n_error_buf=2", "do_reflux=1"], 4, 6),  # a shadow of a level that has a shadow itself
    ("blast3d", "ref_HydroBlast3D", None, ["geometry.prob_lo=0 0 0", "geometry.prob_hi=1.2 1.2 1.2", "geometry.is_periodic=0 0 0", "amr.n_cell=32 32 32", "amr.max_grid_size=8",
                                          "amr.max_level=2", "amr.blocking_factor=8", "amr.n_error_buf=3", "do_reflux=1"], 8, 8),
]
_tag = [0]


def run(exe, args, cwd, nranks, steps):
    _tag[0] += 1
    import shutil
    shutil.copy(os.path.join(ROOT, "tests", "golden", "dust_shell_initial_conditions.txt"), os.path.join(cwd, "initial_conditions.txt"))  # (RadhydroShell reads it)
    procs = []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_RANK=str(r), MASTER_PORT=str(28000 + _tag[0]), MASTER_ADDR="127.0.0.1", QK_COMM_BACKEND="shm",
                   QK_COMM_TIMEOUT="120", QK_MAX_COARSE_STEPS=str(steps), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([os.path.join(HOST, "bin", exe)] + args, env=env, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    plots = re.findall(r"Writing plotfile (\S+)", outs[0])
    return [p.returncode for p in procs], outs, (os.path.join(cwd, plots[-1]) if plots else None)


def compare(a, b):
    A, B = pf.read_plotfile(a), pf.read_plotfile(b)
    if A.finest_level != B.finest_level:
        return f"finest level {A.finest_level} vs {B.finest_level}"
    scale = {v: max([float(np.nanmax(np.abs(f[i]))) for f in A.levels[0].fabs] + [1e-300]) for i, v in enumerate(A.varnames)}
    worst, nan = {}, False
    for l, (la, lb) in enumerate(zip(A.levels, B.levels)):
        if la.boxes != lb.boxes:
            return f"grids of level {l} differ ({len(la.boxes)} vs {len(lb.boxes)} boxes)"
        for fa, fb in zip(la.fabs, lb.fabs):
            nan = nan or bool(np.isnan(fa).any() or np.isnan(fb).any())
            for n, v in enumerate(A.varnames):
                worst[v] = max(worst.get(v, 0.0), float(np.nanmax(np.abs(fa[n] - fb[n]))) / scale[v])
    w = max(worst, key=worst.get)
    return f"levels {A.finest_level + 1}, boxes {[len(l.boxes) for l in A.levels]}, worst relative difference {worst[w]:.2e} ({w})" + (" NaN!" if nan else "")


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, exe, deck, over, nranks, steps in CASES:
        if only and only not in name:
            continue
        base = ([os.path.join(HOST, "decks", deck)] if deck else []) + ([] if any(o.startswith("plotfile_interval") for o in over) else ["plotfile_interval=100000"])
        base += ["checkpoint_interval=-1", "qk.dump_state=s.bin", "qk.level0_distribution=bricks"] + over
        for scheme in ("per_level", "ancestor"):
            one_args = [f"qk.refine_grid_layout_target={nranks}"] if scheme == "per_level" else ["qk.cluster_within_parent=1"]
            many_args = ["qk.distribute_levels=1"] if scheme == "per_level" else ["qk.distribute_levels=0"]
            with tempfile.TemporaryDirectory() as d1, tempfile.TemporaryDirectory() as dn:
                rc1, o1, p1 = run(exe, base + one_args, d1, 1, steps)
                rcn, on, pn = run(exe, base + many_args, dn, nranks, steps)
                if all(r in (0, 1) for r in rc1 + rcn) and not (p1 and pn):  # no plotfile (the problem sets its own interval): the level-0 dumps instead
                    from quokka_amd.simulation import chop_domain, distribute_boxes
                    kv = dict(o.split("=", 1) for o in over if "=" in o)
                    n_cell = [int(x) for x in kv["amr.n_cell"].split()]
                    mgs = int(kv["amr.max_grid_size"])
                    boxes = chop_domain(n_cell, [mgs] * 3)
                    owner = distribute_boxes(boxes, nranks, n_cell, [mgs] * 3)
                    one = np.fromfile(os.path.join(d1, "s.bin"))
                    parts = [np.fromfile(os.path.join(dn, f"s.bin.rank{r}")) for r in range(nranks)]
                    per = one.size // len(boxes)
                    cur, many = [0] * nranks, []
                    for r in owner:
                        many.append(parts[r][cur[r]:cur[r] + per])
                        cur[r] += per
                    many = np.concatenate(many).reshape(len(boxes), -1, mgs ** 3)
                    one = one.reshape(many.shape)
                    rel = [float(np.abs(many[:, n] - one[:, n]).max() / max(np.abs(one[:, n]).max(), 1e-300)) for n in range(one.shape[1])]
                    zone = re.findall(r"Zone-updates on level \d: \d+ \((\d+) grids\)", on[0])
                    same = zone == re.findall(r"Zone-updates on level \d: \d+ \((\d+) grids\)", o1[0])
                    print(f"{name:14s} {scheme:9s} {nranks} ranks: level-0 dump, grids per level {zone} (same as one rank: {same}), worst relative difference {max(rel):.2e}"
                          + (" NaN!" if np.isnan(many).any() else ""), flush=True)
                    continue
                ok = all(r in (0, 1) for r in rc1 + rcn) and p1 and pn
                if not ok:
                    bad = next((o for r, o in zip(rc1 + rcn, o1 + on) if r not in (0, 1)), "")
                    why = [ln for ln in bad.splitlines() if "Abort" in ln or "qkhost" in ln or "rror" in ln][:2]
                    print(f"{name:14s} {scheme:9s} {nranks} ranks: FAILED rc {rc1} {rcn}: {why}", flush=True)
                    continue
                print(f"{name:14s} {scheme:9s} {nranks} ranks: {compare(p1, pn)}", flush=True)


if __name__ == "__main__":
    main()
