"""RadhydroShell physics check (the one the reference has for BASELINE config 3: extern/dust_shell/analyze.py:50-57): density-weighted
mean |v| / a0 of the shell against the thin-shell solution  M(R) = sqrt(2) M0 sqrt(1 - 1/R),
T(R) = (sqrt(R (R - 1)) + ln(sqrt(R) + sqrt(R - 1))) / (M0 sqrt(2)),  M0 = sqrt(L kappa0 / (4 pi r0 c)) / a0.
usage: python profiles/tools/shell_velocity.py [N] [nsamples]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from quokka_amd.multifab import Context  # noqa: E402
from quokka_amd.radhydro import ShellConstants as S, shell_problem  # noqa: E402


def analytic_mach(T):
    M0 = np.sqrt(S.L_star * S.kappa0 / (4.0 * np.pi * S.r_0 * S.c)) / S.a0
    R = np.linspace(1.0, 3.0, 20001)
    Tr = (np.sqrt(R * (R - 1.0)) + np.log(np.sqrt(R) + np.sqrt(R - 1.0))) / (M0 * np.sqrt(2.0))
    return np.sqrt(2.0) * M0 * np.sqrt(1.0 - 1.0 / np.interp(T, Tr, R))


def mean_mach(sim):
    num = den = 0.0
    for b in range(sim.lev.nboxes):
        U = sim.state_new_cc_.valid(b)
        rho = U[0]
        num += float(((U[1] * U[1] + U[2] * U[2] + U[3] * U[3]).sqrt()).sum())  # rho |v| = |p|
        den += float(rho.sum())
    return num / den / S.a0


def run(N=64, nsamples=5, t_end=0.125):
    here = os.path.dirname(os.path.abspath(__file__))
    tab = np.loadtxt(os.path.join(here, "..", "..", "tests", "golden", "dust_shell_initial_conditions.txt"), skiprows=1)
    sim = shell_problem(Context(0), N, (tab[:, 0], tab[:, 2], tab[:, 3]), max_grid_size=min(N, 128), pow_mode=1)
    sim.maxTimesteps_ = 10 ** 9
    t0 = S.r_0 / S.a0
    out = []
    for n in range(1, nsamples + 1):
        sim.stopTime_ = t_end * t0 * n / nsamples
        assert sim.evolve()
        T = sim.tNew_ / t0
        out.append((T, mean_mach(sim), float(analytic_mach(T))))
        print(f"T = {T:.4f}  steps = {sim.istep:5d}  <|v|>/a0 = {out[-1][1]:.4f}  thin-shell solution = {out[-1][2]:.4f}  ratio = {out[-1][1] / out[-1][2]:.4f}", flush=True)
    return out


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 5)
