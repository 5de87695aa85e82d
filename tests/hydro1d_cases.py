"""The reference's 1-D hydro known-answer tests with tabulated solutions, as data: driver settings (with the
reference lines they come from), the committed solution tables (tests/golden/, copied data files of the reference's extern/), and
the error norm of QuokkaSimulation::computeAfterEvolve.  Shared by the oracle test (CPU) and the GPU parity test."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = {
    # src/problems/HydroLeblanc/test_hydro_leblanc.cpp:32 (gamma), :60-71 (ICs), :118-127 (BC states), :337-341 (driver), :371 (tolerance);
    # tests/leblanc.in (2000 cells on [0, 9]); extern/ppm1d/leblanc.dat columns (i, x, density, pressure, velocity), two header lines
    "leblanc": dict(spec=dict(gamma=5.0 / 3.0, profile=0, x_split=3.0, left=[1.0, 0.0, (2.0 / 3.0) * 1.0e-1], right=[1.0e-3, 0.0, (2.0 / 3.0) * 1.0e-10],
                              dirichlet=1, cfl=0.1, max_dt=1e-3, init_dt=1e-5, stop_time=6.0), max_timesteps=50000, nx=2000, hi=9.0,
                    table=("ppm1d_leblanc.dat", 2, (1, 2, 4, 3)), tol=0.002),
    # src/problems/HydroVacuum/test_hydro_vacuum.cpp:16, :40-48, :85-93, :233-236, :257; tests/vacuum.in (100 cells on [0, 1]);
    # extern/Toro/e1rpex.out columns (x, density, velocity, pressure, ...), no header
    "vacuum": dict(spec=dict(gamma=1.4, profile=0, x_split=0.5, left=[1.0, -2.0, 0.4], right=[1.0, 2.0, 0.4], dirichlet=1, cfl=0.8, max_dt=1e-3,
                             stop_time=0.15), max_timesteps=5000, nx=100, hi=1.0, table=("Toro_e1rpex.out", 0, (0, 1, 2, 3)), tol=0.015),
    # src/problems/HydroShuOsher/test_hydro_shuosher.cpp:15, :39-47, :88-96, :217-220, :241; tests/ShuOsher.in (400 cells on [0, 10]);
    # extern/ShuOsher_athena_3c_hllc_vl.txt columns (i, x1v, rho, press, vel1, ...), two header lines
    "shuosher": dict(spec=dict(gamma=1.4, profile=1, x_split=1.0, left=[3.857143, 2.629369, 10.33333], right=[1.0, 0.0, 1.0], dirichlet=1, cfl=0.2,
                               stop_time=1.8), max_timesteps=20000, nx=400, hi=10.0, table=("ShuOsher_athena_3c_hllc_vl.txt", 2, (1, 2, 4, 3)), tol=0.01),
    # src/problems/HydroHighMach/test_hydro_highmach.cpp:32, :57-62, :199; tests/HighMach.in (128 cells, periodic, cfl 0.4, stop_time 3,
    # max_grid_size 64); extern/highmach_reference.txt columns (x, density, velocity, pressure), one header line
    "highmach": dict(spec=dict(gamma=5.0 / 3.0, profile=2, dirichlet=0, cfl=0.4, stop_time=3.0), max_timesteps=100000, nx=128, hi=1.0, mgs=64,
                     table=("highmach_reference.txt", 1, (0, 1, 2, 3)), tol=0.26),
    # src/problems/HydroSMS/test_hydro_sms.cpp:17, :57-66 (states given as rho, m, E), :96-105, :237-240, :294; tests/SlowMovingShock.in
    # (100 cells on [0, 1]); exact solution :136-151: the shock at 0.5 + 0.1096 t between (3.86, -0.81, 10.3334) and (1, -3.44, 1)
    "sms": dict(spec=dict(gamma=1.4, profile=3, x_split=0.5, left=[3.86, -3.1266, 27.0913], right=[1.0, -3.44, 8.4168], dirichlet=1, cfl=0.2,
                          stop_time=1.0), max_timesteps=20000, nx=100, hi=1.0, exact=dict(vshock=0.1096, left=[3.86, -0.81, 10.3334], right=[1.0, -3.44, 1.0]),
                tol=0.005),
    # src/problems/HydroWave/test_hydro_wave.cpp:16 (gamma), :38-71 (eigenmode), :99-101 (driver), :124-141 (error: rms over the components
    # except the auxiliary internal energy of the mean |U(t=1) - U(0)|), err_tol 1e-8 for Nx = 100; tests/hydro_wave.in (100 cells, periodic)
    "wave": dict(spec=dict(gamma=5.0 / 3.0, profile=4, dirichlet=0, cfl=0.1, stop_time=1.0), max_timesteps=20000, nx=100, hi=1.0, tol=1.0e-8),
}


def wave_error(U0, U1):
    """test_hydro_wave.cpp:124-141"""
    return float(np.sqrt(sum(np.abs(U1[n] - U0[n]).mean() ** 2 for n in range(6) if n != 5)))


def reference_state(name, nx=None):
    """computeReferenceSolution of the four problems: the table interpolated to the cell centres, as conserved variables"""
    c = CASES[name]
    nx = nx or c["nx"]
    if "exact" in c:  # a moving discontinuity between two constant states
        e = c["exact"]
        xs = (np.arange(nx) + 0.5) * (c["hi"] / nx)
        left = xs < (c["spec"]["x_split"] + e["vshock"] * c["spec"]["stop_time"])
        rho, v, P = (np.where(left, e["left"][n], e["right"][n]) for n in range(3))
        g = c["spec"]["gamma"]
        U = np.zeros((6, nx))
        U[0], U[1] = rho, rho * v
        U[4] = P / (g - 1.0) + 0.5 * rho * (v * v)
        U[5] = P / (g - 1.0)
        return U
    fname, skip, (cx, crho, cv, cP) = c["table"]
    tab = np.loadtxt(os.path.join(GOLDEN, fname), skiprows=skip, comments=None if skip else "#")
    order = np.argsort(tab[:, cx], kind="stable")
    xs = (np.arange(nx) + 0.5) * (c["hi"] / nx)
    rho, v, P = (np.interp(xs, tab[order, cx], tab[order, col]) for col in (crho, cv, cP))
    g = c["spec"]["gamma"]
    U = np.zeros((6, nx))
    U[0], U[1] = rho, rho * v
    U[4] = P / (g - 1.0) + 0.5 * rho * (v * v)
    U[5] = P / (g - 1.0)
    return U


def error_norm(ref, sol):
    """relative rms L1 error norm over all components (reference src/QuokkaSimulation.hpp:620-644)"""
    err = np.sqrt(sum(np.abs(ref[n] - sol[n]).sum() ** 2 for n in range(ref.shape[0])))
    return float(err / np.sqrt(sum(np.abs(ref[n]).sum() ** 2 for n in range(ref.shape[0]))))


def oracle_sim(oracle, name):
    from oracle.pyoracle import HYDRO1D
    c = CASES[name]
    mgs = c.get("mgs", c["nx"])
    return oracle.sim(HYDRO1D, 1, [c["nx"], 1, 1], [0, 0, 0], [c["hi"], 1, 1], [0 if c["spec"]["dirichlet"] else 1, 1, 1], max_grid_size=[mgs, 1, 1],
                      max_timesteps=c["max_timesteps"], hydro1d=c["spec"])


def gather_x(sim):
    return np.concatenate([sim.valid(b)[:, 0, 0, :] for b in range(sim.nboxes)], axis=-1)
