"""Where does the composite energy of an AMR Sedov run change?  Sums the composite energy before / after every regrid and every
coarse step: python profiles/tools/amr_energy_audit.py [N] [nsteps]"""
import os
import sys

sys.path.insert(0, os.getcwd())
from quokka_amd.amr_simulation import AmrSimulation, sedov_amr_problem
from quokka_amd.multifab import Context

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mgs = int(sys.argv[3]) if len(sys.argv) > 3 else 32
bf = int(sys.argv[4]) if len(sys.argv) > 4 else 8
amr = sedov_amr_problem(Context(0), N, 2, max_grid_size=mgs, blocking_factor=bf)
E0 = amr.composite_sum(4)
acc = {"regrid": 0.0, "advance": 0.0}
orig_regrid = AmrSimulation.regrid
depth = [0]


def audited_regrid(self, lev):
    if depth[0] > 0:
        return orig_regrid(self, lev)
    depth[0] += 1
    before = self.composite_sum(4)
    nb0 = [len(L.all_boxes) for L in self.levels]
    orig_regrid(self, lev)
    d = self.composite_sum(4) - before
    acc["regrid"] += d
    if abs(d) > 1e-12 * abs(E0):
        print(f"  regrid({lev}) at coarse step {self.istep[0]}: {d / E0:.3e}; boxes {nb0} -> {[len(L.all_boxes) for L in self.levels]}; "
              f"level boxes now {[L.all_boxes for L in self.levels[1:]]}", flush=True)
    depth[0] -= 1


AmrSimulation.regrid = audited_regrid
for it in range(nsteps):
    before = amr.composite_sum(4)
    r0 = acc["regrid"]
    c0 = dict(amr.levels[0].counters)
    amr.step()
    d_adv = (amr.composite_sum(4) - before) - (acc["regrid"] - r0)
    acc["advance"] += d_adv
    if abs(d_adv) > 1e-13 * abs(E0):
        print(f"  step {it}: advance {d_adv / E0:.3e}, boxes {[L.all_boxes for L in amr.levels[1:]]}, counters {[dict(L.counters) for L in amr.levels]}", flush=True)
E1 = amr.composite_sum(4)
print(f"N = {N}, {nsteps} coarse steps, levels {[len(L.all_boxes) for L in amr.levels]}: total relative change {(E1 - E0) / E0:.3e}; "
      f"from regrids {acc['regrid'] / E0:.3e}, from advances (fluxes + reflux + average-down) {acc['advance'] / E0:.3e}")
