"""The C++17 host mirror (quokka_amd/host) driven by the reference's OWN problem files, compiled unchanged in place (bin/ref_*, see
tests/test_reference_problems_gpu.py): QuokkaSimulation<problem_t>, HydroSystem<problem_t>, RadSystem<problem_t>, trait specialisations and
ParmParse decks run end-to-end through the C-ABI and reproduce the committed oracle states bit for bit.  (Until round 2 this module ran
adapted copies of eleven problem files; they are gone.)"""
import os
import shutil
import subprocess

import numpy as np
import pytest

from test_reference_problems_gpu import exe, extern_tree

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "quokka_amd", "host")


def run(name, args, tmp_path, allow_fail=False, cwd=None):
    dump = str(tmp_path / "state.bin")
    cmd = [exe(name)] + args + [f"qk.dump_state={dump}"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=cwd)
    assert allow_fail or p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    meta = [float(x) for x in open(dump + ".meta").read().split()]
    return np.fromfile(dump, dtype=np.float64), meta, p.stdout


def assert_state_matches(got, want, tol=1e-12):
    """bit for bit — or, where the unmodified problem file evaluates std::pow / std::exp in DEVICE code for its initial or boundary states
    (device libm and glibc differ by an ulp; the adapted copies this module used until round 2 evaluated them on the host), north_star's
    tolerance: relative L1 <= 1e-12 on every conserved component"""
    if np.array_equal(got, want):
        return 0.0
    worst = 0.0
    for n in range(want.shape[0]):
        scale = np.abs(want[n]).sum()
        if scale > 0.0:
            worst = max(worst, float(np.abs(got[n] - want[n]).sum() / scale))
        else:
            assert not got[n].any(), n
    assert worst <= tol, f"relative L1 {worst:.3e} > {tol:g}"
    return worst


def sod_tree(tmp_path):
    """HydroShocktube's computeReferenceSolution opens ../extern/ppm1d/output relative to its working directory"""
    return extern_tree(tmp_path, {"ppm1d/output": "ppm1d_sod_exact.txt"})


def test_sod_shocktube_executable(tmp_path):
    """BASELINE config 1 through the C++ mirror (1-D build), the reference's problem file unchanged: its Dirichlet boundary code runs as a
    kernel; on this UNREFINED grid the error norm is 0.00204 (the reference's 0.002 holds for its ctest configuration with one refined
    level: next test), so the exit status is not asserted here — the final state equals the committed oracle state in every bit."""
    data, meta, out = run("ref_HydroShocktube", [os.path.join(HOST, "decks", "shocktube.in")], tmp_path, allow_fail=True, cwd=sod_tree(tmp_path))
    gold = np.load(os.path.join(ROOT, "tests", "golden", "sod_1024_final.npy"))
    assert np.array_equal(data.reshape(6, 1024), gold)
    assert abs(meta[1] - 0.4) < 1e-12 and meta[5] < 0.0021
    assert "Performance figure-of-merit" in out


def test_sod_shocktube_amr_meets_the_reference_ctest_criterion(tmp_path):
    """The reference's own ctest configuration of the Sod tube (tests/shocktube.in: one refined level on the density gradient,
    subcycled, refluxed): relative L1 error of the level-0 state vs the exact solution <= 0.002, exit status 0."""
    data, meta, out = run("ref_HydroShocktube", [os.path.join(HOST, "decks", "shocktube_amr.in")], tmp_path, cwd=sod_tree(tmp_path))
    assert abs(meta[1] - 0.4) < 1e-12 and meta[5] <= 0.002, meta
    # the refined level must have done real work (its cells count in the figure of merit) and the coarse state differs from the unrefined run
    gold = np.load(os.path.join(ROOT, "tests", "golden", "sod_1024_final.npy"))
    assert not np.array_equal(data.reshape(6, 1024), gold)


@pytest.mark.parametrize("handoff", [1, 0])
def test_sedov_executable_matches_golden(tmp_path, handoff):
    """32^3 Sedov, 10 steps, through the C++ mirror (3-D build, fused path) == committed oracle state; handoff: the primitive hand-off between
    the stages of a step (qk_hydro_stage_args::prim_out / prim_in; the default of a plain hydro level) or the conserved intermediate state"""
    data, meta, out = run("ref_HydroBlast3D", ["geometry.prob_lo=0 0 0", "geometry.prob_hi=1.2 1.2 1.2", "geometry.is_periodic=0 0 0",
                                                 "amr.n_cell=32 32 32", "amr.max_grid_size=32", "max_timesteps=10", f"qk.prim_handoff={handoff}"],
                          tmp_path, allow_fail=True)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "sedov_32_step10.npy"))
    assert int(meta[0]) == 10
    assert np.array_equal(data.reshape(6, 32, 32, 32), gold)
    assert "Energy conservation is OK." in out
    assert f"prim_handoff={handoff} prim_handoff_dropped=0" in out


def test_sedov_amr_executable_matches_python_driver(tmp_path, ctx):
    """BASELINE config 5 (Sedov, amr.max_level = 2, subcycling + reflux) through the C++ mirror's level machinery (quokka_amr.hpp) at
    32^3 base resolution: the level-0 state after 8 coarse steps equals the Python driver's (quokka_amd/amr_simulation.py), which
    tests/test_amr_driver_gpu.py pins by properties — same kernels, same grid generation (library), same schedule."""
    from quokka_amd.amr_simulation import sedov_amr_problem
    N, nsteps = 32, 8
    data, meta, out = run("ref_HydroBlast3D", ["geometry.prob_lo=0 0 0", "geometry.prob_hi=1.2 1.2 1.2", "geometry.is_periodic=0 0 0", f"amr.n_cell={N} {N} {N}",
                                                 "amr.max_level=2", "amr.max_grid_size=32", "amr.blocking_factor=8", "amr.n_error_buf=3", "do_reflux=1",
                                                 f"max_timesteps={nsteps}", "plotfile_interval=100"], tmp_path, allow_fail=True, cwd=str(tmp_path))
    assert int(meta[0]) == nsteps and "Zone-updates on level 2" in out, out[-1500:]
    amr = sedov_amr_problem(ctx, N, 2, max_grid_size=32, blocking_factor=8)
    for _ in range(nsteps):
        amr.step()
    assert amr.tNew_ == meta[1], (amr.tNew_, meta[1])
    want = np.zeros((6, N, N, N))
    c = amr.levels[0]
    for (lo, hi), v in zip(c.my_boxes, c.gather_valid_local()):
        want[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = v
    assert np.array_equal(data.reshape(6, N, N, N), want), f"max abs diff {np.abs(data.reshape(6, N, N, N) - want).max()}"
    for l in range(3):
        assert f"Zone-updates on level {l}: {amr.cellUpdatesEachLevel_[l]} " in out
    # both drivers' plotfile writers: same grids on every level, same data in every cell, same header fields
    from quokka_amd import plotfile
    plotfile.WritePlotFile(amr, str(tmp_path / "py_plt00008"))
    diff = plotfile.compare_plotfiles(str(tmp_path / "plt00008"), str(tmp_path / "py_plt00008"))
    assert all(v == 0.0 for v in diff.values()), diff
    cpp, py = plotfile.read_plotfile(str(tmp_path / "plt00008")), plotfile.read_plotfile(str(tmp_path / "py_plt00008"))
    assert cpp.finest_level == 2 and (cpp.time, cpp.level_steps, cpp.dx, cpp.domains) == (py.time, py.level_steps, py.dx, py.domains)
    assert open(tmp_path / "plt00008" / "Header").read() == open(tmp_path / "py_plt00008" / "Header").read()


BLAST32 = ["geometry.prob_lo=0 0 0", "geometry.prob_hi=1.2 1.2 1.2", "geometry.is_periodic=0 0 0", "amr.n_cell=32 32 32", "amr.max_grid_size=16"]


@pytest.mark.parametrize("amr", [[], ["amr.max_level=2", "amr.blocking_factor=8", "amr.n_error_buf=1"]], ids=["unigrid", "amr"])
def test_checkpoint_restart_reproduces_the_plotfile(tmp_path, amr):
    """The reference's checkpoint_restart_test.sh (tests/checkpoint_restart_test.sh), made stricter: run to step 12 writing a checkpoint at
    step 6; restart from `last_chk`-style directory chk00006 and run to step 12 again; the second plt00012 must equal the first one
    (kept as plt00012.old.*, as AMReX renames it) in every bit of every level.  Also: the plotfile holds exactly the dumped state."""
    from quokka_amd import plotfile
    wd = str(tmp_path)
    common = BLAST32 + amr + ["plotfile_interval=100", "checkpoint_interval=6"]
    data, meta, out = run("ref_HydroBlast3D", common + ["max_timesteps=12"], tmp_path, allow_fail=True, cwd=wd)
    assert int(meta[0]) == 12 and "Writing checkpoint chk00006" in out and "Writing plotfile plt00012" in out
    assert os.path.islink(os.path.join(wd, "last_chk")) and os.readlink(os.path.join(wd, "last_chk")) == "chk00012"
    first = plotfile.read_plotfile(os.path.join(wd, "plt00012"))
    assert first.varnames == plotfile.HYDRO_NAMES and first.ndim == 3 and first.level_steps[0] == 12
    assert first.finest_level == (2 if amr else 0) and abs(first.time - meta[1]) < 1e-15
    lvl0 = first.levels[0]
    assert lvl0.nghost == 0 and len(lvl0.boxes) == 8
    state = data.reshape(8, 6, 16, 16, 16)
    for b in range(8):
        assert np.array_equal(lvl0.fabs[b], state[b])
        assert np.array_equal(lvl0.minima[b], state[b].reshape(6, -1).min(axis=1))
    h, lev = plotfile.read_checkpoint(os.path.join(wd, "chk00006"))
    assert h.istep[0] == 6 and h.finest_level == first.finest_level or amr  # (the hierarchy may have a different depth at step 6)
    assert lev[0].nghost == 4 and lev[0].fabs[0].shape == (6, 24, 24, 24)

    # VisMF::Header writes m_ngrow as one integer when it is isotropic and as an IntVect otherwise; a reader must take both:
    # the level-0 header of the checkpoint is rewritten in the IntVect form before the restart
    cell_h = os.path.join(wd, "chk00006", "Level_0", "Cell_H")
    lines = open(cell_h).read().split("\n")
    assert lines[3] == "4"
    lines[3] = "(4,4,4)"
    open(cell_h, "w").write("\n".join(lines))
    assert plotfile.read_checkpoint(os.path.join(wd, "chk00006"))[1][0].nghost == 4
    data2, meta2, out2 = run("ref_HydroBlast3D", common + ["max_timesteps=12", "restartfile=chk00006"], tmp_path, allow_fail=True, cwd=wd)
    assert int(meta2[0]) == 12 and meta2[1] == meta[1]
    olds = [d for d in os.listdir(wd) if d.startswith("plt00012.old.")]
    assert len(olds) == 1
    diff = plotfile.compare_plotfiles(os.path.join(wd, "plt00012"), os.path.join(wd, olds[0]))
    assert all(v == 0.0 for v in diff.values()), diff
    assert np.array_equal(data, data2)


def test_radiative_shock_executable_meets_the_reference_criterion_and_matches_oracle(tmp_path, oracle):
    """The reference's RadhydroShockCGS ctest through the C++ mirror (1-D build; the problem's opacity hooks kappa = k0 / rho are
    compiled device code, the Eddington approximation is a trait, Dirichlet states come from setCustomBoundaryConditions): exit status 0 == relative L1
    error of T_rad against Lowrie & Edwards' solution <= 0.005 after ~6000 hydro steps x 10 radiation substeps — and, with the shared
    T^4 evaluation, the final state equals the oracle's full run in every bit."""
    from oracle.pyoracle import RADSHOCK
    cwd = extern_tree(tmp_path, {"LowrieEdwards/shock.txt": "LowrieEdwards_shock.txt"})
    data, meta, out = run("ref_RadhydroShockCGS", [os.path.join(HOST, "decks", "radshock.in"), "radiation.pow_mode=1"], tmp_path, cwd=cwd)
    assert abs(meta[1] - 1.0e-9) < 1e-24, meta
    so = oracle.sim(RADSHOCK, 1, [512, 1, 1], [0, 0, 0], [0.01575, 1, 1], [0, 1, 1], max_grid_size=[512, 1, 1], rad_pow_mode=1)
    assert so.evolve()
    assert so.istep == int(meta[0]) and so.time == meta[1]
    want = so.valid(0).reshape(10, 512)
    got = data.reshape(10, 512)
    assert np.array_equal(got, want), [float(np.abs(got[n] - want[n]).sum() / max(np.abs(want[n]).sum(), 1e-300)) for n in range(10)]


def test_streaming_executable_meets_the_reference_criterion(tmp_path):
    """the reference's RadStreaming ctest through the C++ mirror (1-D build, hydro disabled): exit status 0 == error < 0.01"""
    data, meta, out = run("ref_RadStreaming", [os.path.join(HOST, "decks", "RadStreaming.in")], tmp_path)
    assert int(meta[0]) == 667 and meta[1] == 1.0, meta
    assert np.array_equal(data.reshape(10, 1000)[0], np.ones(1000))


def test_uniform_advecting_executable_matches_oracle(tmp_path, oracle):
    """the reference's RadhydroUniformAdvecting ctest through the C++ mirror (beta_order = 2, radiation CFL 8, periodic): exit status
    0 == T_gas within 1e-10 of T0; the final state equals the oracle's bit for bit."""
    from oracle.pyoracle import ADVECTING
    data, meta, out = run("ref_RadhydroUniformAdvecting", [os.path.join(HOST, "decks", "RadhydroUniformAdvecting.in"), "radiation.pow_mode=1"], tmp_path)
    assert int(meta[0]) == 125, meta
    so = oracle.sim(ADVECTING, 1, [64, 1, 1], [0, 0, 0], [64.0, 1, 1], [1, 1, 1], max_grid_size=[64, 1, 1], rad_pow_mode=1)
    assert so.evolve() and so.time == meta[1]
    assert np.array_equal(data.reshape(10, 64), so.valid(0).reshape(10, 64))


def test_marshak_executable_meets_the_reference_criterion_and_matches_oracle(tmp_path, oracle):
    """the reference's RadMarshak ctest through the C++ mirror: the problem's setCustomBoundaryConditions (the Marshak half-range
    condition) runs as device code for every ghost cell beyond the face; its T^4 material is recognised on probe points and served by the
    library's arithmetic (quokka_host.hpp eosTemperatureModel; anything else would be compiled, QK_HOOK_COMPILED); exit status 0 == radiation
    temperature within 2 per cent of Su & Olson's solution; the final state after 10135 steps equals the oracle's bit for bit."""
    from oracle.pyoracle import MARSHAK
    cwd = extern_tree(tmp_path, {"SuOlson/100pt_tau10p0.dat": "SuOlson_100pt_tau10p0.dat"})
    data, meta, out = run("ref_RadMarshak", [os.path.join(HOST, "decks", "Marshak.in"), "radiation.pow_mode=1"], tmp_path, cwd=cwd)
    assert int(meta[0]) == 10135 and abs(meta[1] - 10.0) < 1e-12, meta
    so = oracle.sim(MARSHAK, 1, [80, 1, 1], [0, 0, 0], [20.0, 1, 1], [0, 1, 1], max_grid_size=[80, 1, 1], rad_pow_mode=1)
    assert so.evolve() and so.time == meta[1]
    assert_state_matches(data.reshape(10, 80), so.valid(0).reshape(10, 80))  # E_inc = a_rad * std::pow(T_H, 4) inside the boundary kernel


def test_radiation_force_executable_meets_the_reference_criterion_and_matches_oracle(tmp_path, oracle):
    """the reference's RadForce ctest through the C++ mirror: isothermal EOS_Traits (gamma = 1, cs_isothermal), Planck opacity 0 with a
    flux-mean opacity, inflow face filled by the problem's setCustomBoundaryConditions on the device (the upper face stays with its BCRec); exit status 0 ==
    Mach number within 0.002 of the steady wind; the final state after 9520 steps equals the oracle's bit for bit."""
    from oracle.pyoracle import RADFORCE
    cwd = extern_tree(tmp_path, {"pressure_tube/optically_thin_wind.txt": "optically_thin_wind.txt"})
    data, meta, out = run("ref_RadForce", [os.path.join(HOST, "decks", "RadForce.in"), "radiation.pow_mode=1"], tmp_path, cwd=cwd)
    assert int(meta[0]) == 9520, meta
    so = oracle.sim(RADFORCE, 1, [128, 1, 1], [0, 0, 0], [1.0263747986171498e16, 1, 1], [0, 1, 1], max_grid_size=[128, 1, 1], rad_pow_mode=1)
    assert so.evolve() and so.time == meta[1]
    assert_state_matches(data.reshape(10, 128), so.valid(0).reshape(10, 128))


def test_marshak_asymptotic_executable_meets_the_reference_criterion(tmp_path, oracle):
    """the reference's RadMarshakAsymptotic ctest through the C++ mirror: the problem's opacity hooks (a temperature power law) are compiled
    device code inside the Newton-Raphson kernel (quokka_amd/host/qk_problem_kernels.hpp), the oracle uses the closed power-law form of the
    library — equal to rounding, no bit-level claim here (that is tests/test_radhydro_gpu.py's); exit status 0 == gas temperature within 9 per cent of the similarity solution after
    90847 steps, and the state agrees with the oracle's to 1e-6."""
    from oracle.pyoracle import MARSHAK_ASYMPTOTIC
    cwd = extern_tree(tmp_path, {"marshak_similarity.csv": "marshak_similarity.csv"})
    data, meta, out = run("ref_RadMarshakAsymptotic", [os.path.join(HOST, "decks", "MarshakAsymptotic.in")], tmp_path, cwd=cwd)
    assert int(meta[0]) == 90847, meta
    so = oracle.sim(MARSHAK_ASYMPTOTIC, 1, [60, 1, 1], [0, 0, 0], [0.66, 1, 1], [0, 1, 1], max_grid_size=[60, 1, 1])
    assert so.evolve()
    A, B = data.reshape(10, 60), so.valid(0).reshape(10, 60)
    for n in (4, 6):
        assert float(np.abs(A[n] - B[n]).sum() / np.abs(B[n]).sum()) < 1e-6


def test_marshak_asymptotic_executable_with_the_wavespeed_correction(tmp_path, oracle):
    """the reference's ctest RadMarshakAsymptoticCorr (tests/MarshakAsymptoticCorr.in = MarshakAsymptotic.in + marshak.use_wavespeed_correction = true)
    through the C++ mirror: QuokkaSimulation::use_wavespeed_correction_ set by the unchanged problem file, ComputeCellOpticalDepth instantiated in the
    problem's translation unit with its compiled opacity hook (qk_problem_kernels.hpp).  Exit status 0 == within 9 per cent of the similarity
    solution; the state agrees with the oracle's corrected run to 1e-6 (compiled std::pow against the closed power-law form) and differs from the
    uncorrected one."""
    from oracle.pyoracle import MARSHAK_ASYMPTOTIC
    cwd = extern_tree(tmp_path, {"marshak_similarity.csv": "marshak_similarity.csv"})
    data, meta, out = run("ref_RadMarshakAsymptotic", [os.path.join(HOST, "decks", "MarshakAsymptotic.in"), "marshak.use_wavespeed_correction=true"], tmp_path, cwd=cwd)
    so = oracle.sim(MARSHAK_ASYMPTOTIC, 1, [60, 1, 1], [0, 0, 0], [0.66, 1, 1], [0, 1, 1], max_grid_size=[60, 1, 1])
    so.set_wavespeed_correction(True)
    assert so.evolve() and int(meta[0]) == so.istep
    A, B = data.reshape(10, 60), so.valid(0).reshape(10, 60)
    for n in (4, 6):
        assert float(np.abs(A[n] - B[n]).sum() / np.abs(B[n]).sum()) < 1e-6
    plain = oracle.sim(MARSHAK_ASYMPTOTIC, 1, [60, 1, 1], [0, 0, 0], [0.66, 1, 1], [0, 1, 1], max_grid_size=[60, 1, 1])
    assert plain.evolve()
    assert float(np.abs(A[6] - plain.valid(0).reshape(10, 60)[6]).sum() / np.abs(A[6]).sum()) > 1e-6


def test_passive_scalar_executable_meets_the_reference_criteria(tmp_path):
    """the reference's PassiveScalar ctest (tests/PassiveScalar.in: one refined level on the density gradient, subcycled, refluxed;
    src/problems/PassiveScalar/test_scalars.cpp): scalar conserved to 1e-14 and relative rms L1 error <= 0.008 after four box
    crossings — exit status 0.  Seven components (6 + 1 passive scalar) through interpolation, flux registers and average-down."""
    data, meta, out = run("ref_PassiveScalar", [os.path.join(HOST, "decks", "PassiveScalar.in")], tmp_path)
    assert abs(meta[1] - 2.0) < 1e-12 and meta[5] <= 0.008, meta
    assert "Zone-updates on level 1" in out
    assert data.size == 7 * 128


def test_contact_wave_executable_error_is_exactly_zero(tmp_path):
    """the reference's HydroContact ctest (src/problems/HydroContact/test_hydro_contact.cpp:213-216) through the C++ mirror: after
    t = 2 the relative L1 error norm against the initial state must be 0.0 — not small: zero — and the exit status says so.  Two
    passive scalars (all zero) ride along as in the reference; every component of the final state equals the initial one."""
    data, meta, out = run("ref_HydroContact", [os.path.join(HOST, "decks", "contact_wave.in")], tmp_path)
    assert meta[5] == 0.0 and abs(meta[1] - 2.0) < 1e-12, meta
    U = data.reshape(8, 100)
    assert np.array_equal(U[0], np.where((np.arange(100) + 0.5) / 100 < 0.5, 1.4, 1.0))
    assert not U[1:4].any() and not U[6:].any() and np.array_equal(U[4], U[5])


_RENDEZVOUS = __import__("itertools").count()


def run_ranks(name, args, tmp_path, nranks, port, backend="shm"):
    """N processes of one executable sharing this GPU: the multi-rank layer of the host mirror (quokka_amd/host/qk_comm.hpp) with its test
    transport (QK_COMM_BACKEND=shm: buffers staged through the host; RCCL refuses two ranks on one device).  Returns the per-rank dumps."""
    dump = str(tmp_path / f"state_n{nranks}_{backend}.bin")
    procs = []
    port = 29000 + next(_RENDEZVOUS)  # (one tag per call: the files two runs leave under /dev/shm must never meet, whatever port the caller suggested)
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_RANK=str(r), MASTER_PORT=str(port), MASTER_ADDR="127.0.0.1",
                   QK_COMM_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([exe(name)] + args + [f"qk.dump_state={dump}"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
        # (exit status 1 = the problem's own kinetic-energy criterion, which needs the full run to t = 1; anything else is a crash)
        assert p.returncode in (0, 1), out[-2000:]
    if nranks == 1:
        return [np.fromfile(dump, dtype=np.float64)], outs
    return [np.fromfile(dump + f".rank{r}", dtype=np.float64) for r in range(nranks)], outs


@pytest.mark.parametrize("nranks", [2, 4, 8])
def test_sedov_on_several_ranks_equals_one_rank(tmp_path, nranks):
    """64^3 Sedov in eight 32^3 boxes, 12 steps (the blast crosses box and rank boundaries; FOFC does not fire, the fused path carries every
    stage): N ranks — boxes distributed in bricks, ghost strips packed / exchanged / unpacked, dt and counters all-reduced — must reproduce
    the one-rank state bit for bit, and rank 0 must report the same conservation check."""
    from quokka_amd.simulation import chop_domain, distribute_boxes
    args = ["geometry.prob_lo=0 0 0", "geometry.prob_hi=1.2 1.2 1.2", "geometry.is_periodic=0 0 0", "amr.n_cell=64 64 64", "amr.max_grid_size=32",
            "max_timesteps=12"]
    (one,), outs1 = run_ranks("ref_HydroBlast3D", args, tmp_path, 1, 29611)
    parts, outs = run_ranks("ref_HydroBlast3D", args, tmp_path, nranks, 29611 + nranks)
    boxes = chop_domain([64, 64, 64], [32, 32, 32])
    owner = distribute_boxes(boxes, nranks, [64, 64, 64], [32, 32, 32])
    one = one.reshape(8, 6, 32, 32, 32)
    cursor = [0] * nranks
    for b, r in enumerate(owner):
        chunk = parts[r][cursor[r]:cursor[r] + 6 * 32 ** 3].reshape(6, 32, 32, 32)
        cursor[r] += 6 * 32 ** 3
        assert np.array_equal(chunk, one[b]), (b, r, np.abs(chunk - one[b]).max())
    assert all(cursor[r] == parts[r].size for r in range(nranks))
    assert sorted(set(owner)) == list(range(nranks))
    assert "Energy conservation is OK." in outs[0] and "Energy conservation is OK." in outs1[0]


@pytest.mark.parametrize("nranks", [2, 4])
def test_overlapped_fill_schedule_of_the_cxx_host_equals_one_rank(tmp_path, nranks):
    """north_star's schedule in the C++17 host (quokka_host.hpp fillAndStage): the boxes whose ghost cells are all filled on this rank are advanced
    by a fused launch over a sub-level while the strips of the others travel (RCCL's stream in production; the shm test transport here keeps the
    ORDER of the schedule: pack -> send -> same-rank copies -> physical boundaries of the early group -> early launch -> receive -> unpack ->
    boundaries of the late group -> late launch).  64^3 in 64 boxes of 16^3, the split forced on (qk.min_overlap_cells = 1): the union of the
    ranks' boxes equals the one-rank run in every bit, also in the carried-rhs mode."""
    from quokka_amd.simulation import chop_domain, distribute_boxes
    for extra in ([], ["hydro.rk2_carry_rhs=1"]):
        args = ["geometry.prob_lo=0 0 0", "geometry.prob_hi=1.2 1.2 1.2", "geometry.is_periodic=0 0 0", "amr.n_cell=64 64 64", "amr.max_grid_size=16",
                "max_timesteps=10"] + extra
        (one,), outs1 = run_ranks("ref_HydroBlast3D", args, tmp_path, 1, 29651)
        parts, outs = run_ranks("ref_HydroBlast3D", args + ["qk.min_overlap_cells=1"], tmp_path, nranks, 29651 + nranks)
        assert all("overlapped ghost fill:" in o for o in outs), outs[0][-1500:]
        assert "overlapped ghost fill:" not in outs1[0]
        boxes = chop_domain([64, 64, 64], [16, 16, 16])
        owner = distribute_boxes(boxes, nranks, [64, 64, 64], [16, 16, 16])
        one = one.reshape(64, 6, 16, 16, 16)
        cursor = [0] * nranks
        for b, r in enumerate(owner):
            chunk = parts[r][cursor[r]:cursor[r] + 6 * 16 ** 3].reshape(6, 16, 16, 16)
            cursor[r] += 6 * 16 ** 3
            assert np.array_equal(chunk, one[b]), (extra, b, r, np.abs(chunk - one[b]).max())


@pytest.mark.parametrize("nranks,distribution,mgs", [(2, "interleaved", 8), (4, "interleaved", 8), (2, "bricks", 16)])
def test_amr_hierarchy_of_the_cxx_host_across_ranks_matches_one_rank(tmp_path, nranks, distribution, mgs):
    """BASELINE config 5's structure (Sedov, amr.max_level = 2, subcycling, reflux, regrid every 2 steps) in the C++17 host on several ranks
    (quokka_amr.hpp) in the scheme that keeps a refined box on the rank of its level-0 ancestor (qk.distribute_levels = 0; the default on several ranks is
    a box -> rank map per level: test_amr_levels_with_their_own_distribution_in_the_cxx_host): grids are clustered inside each level-0 box, tile flags are
    all-reduced, reflux increments cross ranks through SumBoundary.  32^3 base grid; with 8^3 level-0 boxes the refined region around the blast
    (which the problem puts in the corner cell of the octant) spans level-0 boxes of several ranks on both finer levels; with 16^3 boxes it
    stays inside one, so the other ranks hold EMPTY levels 1 and 2 and still take part in every collective step.  Against ONE rank building the
    same grids (qk.cluster_within_parent = 1): same time steps, same number of grids and zone updates on every level, level-0 state (which holds
    the average of every finer level) equal to rounding — the reflux additions are reassociated —, energy conserved to the problem's own 2e-15."""
    from quokka_amd.simulation import chop_domain, distribute_boxes, distribute_boxes_interleaved
    import re
    args = ["geometry.prob_lo=0 0 0", "geometry.prob_hi=1.2 1.2 1.2", "geometry.is_periodic=0 0 0", "amr.n_cell=32 32 32", f"amr.max_grid_size={mgs}",
            "amr.max_level=2", "amr.blocking_factor=8", "amr.n_error_buf=3", "do_reflux=1", "max_timesteps=8"]
    (one,), outs1 = run_ranks("ref_HydroBlast3D", args + ["qk.cluster_within_parent=1"], tmp_path, 1, 29651)
    parts, outs = run_ranks("ref_HydroBlast3D", args + [f"qk.level0_distribution={distribution}", "qk.distribute_levels=0"], tmp_path, nranks,
                            29651 + 3 * nranks + len(distribution))
    zone = re.compile(r"Zone-updates on level (\d): (\d+) \((\d+) grids\)")
    z1 = zone.findall(outs1[0])
    assert len(z1) == 3 and zone.findall(outs[0]) == z1, (zone.findall(outs[0]), z1)
    assert int(z1[1][2]) >= (8 if mgs == 8 else 1), z1  # (8 level-1 grids: one in each of the 2 x 2 x 2 level-0 boxes around the corner)
    boxes = chop_domain([32, 32, 32], [mgs] * 3)
    fn = distribute_boxes_interleaved if distribution == "interleaved" else distribute_boxes
    owner = fn(boxes, nranks, [32, 32, 32], [mgs] * 3)
    assert sorted(set(owner)) == list(range(nranks))
    if mgs == 8 and distribution == "interleaved":  # the level-0 boxes under the refined region belong to more than one rank
        under = {owner[ib + 4 * (jb + 4 * kb)] for ib in range(2) for jb in range(2) for kb in range(2)}
        assert len(under) > 1, under
    nb, n3 = len(boxes), mgs ** 3
    one = one.reshape(nb, 6, mgs, mgs, mgs)
    cursor = [0] * nranks
    worst = 0.0
    for b, r in enumerate(owner):
        chunk = parts[r][cursor[r]:cursor[r] + 6 * n3].reshape(6, mgs, mgs, mgs)
        cursor[r] += 6 * n3
        for n in range(6):
            worst = max(worst, float(np.abs(chunk[n] - one[b][n]).max() / np.abs(one[:, n]).max()))
    assert all(cursor[r] == parts[r].size for r in range(nranks))
    assert worst <= 1e-13, worst
    assert "Energy conservation is OK." in outs[0] and "Energy conservation is OK." in outs1[0]
    dump1, dumpn = str(tmp_path / "state_n1_shm.bin"), str(tmp_path / f"state_n{nranks}_shm.bin")
    assert open(dump1 + ".meta").read().split()[:3] == open(dumpn + ".rank0.meta").read().split()[:3]  # steps, time, dt


@pytest.mark.parametrize("nranks,scheme", [(2, "ancestor"), (4, "ancestor"), (4, "per_level")])
def test_plotfile_and_checkpoint_written_by_several_ranks(tmp_path, nranks, scheme):
    """AMRSimulation::WritePlotFile / WriteCheckpointFile of the C++17 host on several ranks (quokka_io.hpp: VisMF::Write with one data file per
    rank — Cell_D_00000 .. — and ONE header by rank 0 that lists the boxes of the whole level, each with the file of its owner and an offset
    that follows from the box list; per-fab minima / maxima reduced over the ranks; Header, metadata and the last_chk link by rank 0).  The
    three-level Sedov hierarchy of the test above: the plotfile and the checkpoint of N ranks hold the same grids as the one-rank files and the
    same data to rounding (the reflux additions are reassociated across ranks), the header tables are those of the data, every rank's file is
    referenced, and a restart from the N-rank checkpoint — on N ranks and on ONE rank — continues to the same final state as the uninterrupted
    N-rank run (bit for bit on N ranks).  scheme: where the refined boxes live — on the rank of their level-0 ancestor (qk.distribute_levels = 0, the one-rank
    reference clusters inside the level-0 boxes too) or by a box -> rank map per level (the default on several ranks; the one-rank reference chops its
    levels for N boxes)."""
    from quokka_amd import plotfile as pf
    many = ["qk.level0_distribution=interleaved", "qk.distribute_levels=0"] if scheme == "ancestor" else ["qk.level0_distribution=bricks"]
    single = ["qk.cluster_within_parent=1"] if scheme == "ancestor" else [f"qk.refine_grid_layout_target={nranks}"]
    base = ["geometry.prob_lo=0 0 0", "geometry.prob_hi=1.2 1.2 1.2", "geometry.is_periodic=0 0 0", "amr.n_cell=32 32 32", "amr.max_grid_size=8",
            "amr.max_level=2", "amr.blocking_factor=8", "amr.n_error_buf=3", "do_reflux=1"]
    out1, outn = tmp_path / "one", tmp_path / "many"
    os.makedirs(out1)
    os.makedirs(outn)
    io1 = [f"plotfile_prefix={out1}/plt", f"checkpoint_prefix={out1}/chk", "plotfile_interval=4", "checkpoint_interval=4", "max_timesteps=4"]
    ion = [f"plotfile_prefix={outn}/plt", f"checkpoint_prefix={outn}/chk", "plotfile_interval=4", "checkpoint_interval=4", "max_timesteps=4"]
    for sub in ("w", "r", "s"):
        os.makedirs(tmp_path / sub)
    run_ranks("ref_HydroBlast3D", base + io1 + single, tmp_path, 1, 29711 + 20 * len(scheme))
    run_ranks("ref_HydroBlast3D", base + ion + many, tmp_path, nranks, 29711 + nranks + 20 * len(scheme))
    # the plotfile
    A, B = pf.read_plotfile(str(out1 / "plt00004")), pf.read_plotfile(str(outn / "plt00004"))
    assert B.finest_level == 2 and A.finest_level == 2
    for la, lb in zip(A.levels, B.levels):
        assert la.boxes == lb.boxes
        for n, fab in enumerate(lb.fabs):  # the header's tables are those of the data
            assert np.array_equal(lb.minima[n], fab.reshape(fab.shape[0], -1).min(axis=1)) and np.array_equal(lb.maxima[n], fab.reshape(fab.shape[0], -1).max(axis=1))
    files = {ln.split()[1] for ln in open(outn / "plt00004" / "Level_0" / "Cell_H") if ln.startswith("FabOnDisk:")}
    assert files == {f"Cell_D_{r:05d}" for r in range(nranks)}
    diff = pf.compare_plotfiles(str(out1 / "plt00004"), str(outn / "plt00004"))
    scale = {v: max(float(np.abs(f[i]).max()) for f in A.levels[0].fabs) for i, v in enumerate(A.varnames)}
    assert all(diff[v] <= 1e-13 * max(scale[v], 1e-300) or diff[v] <= 1e-13 * scale["gasEnergy"] for v in A.varnames), diff
    # the checkpoint: global grids in the Header, ghost cells kept
    h1, c1 = pf.read_checkpoint(str(out1 / "chk00004"))
    hn, cn = pf.read_checkpoint(str(outn / "chk00004"))
    assert hn.finest_level == 2 and hn.grids == h1.grids and hn.istep == h1.istep and hn.dt == h1.dt
    assert all(a.boxes == b.boxes and a.nghost == b.nghost == 4 for a, b in zip(c1, cn))
    assert os.path.islink(outn / "last_chk")
    # restart from the N-rank checkpoint: N ranks and one rank, against the uninterrupted N-rank run
    more = base + ["plotfile_interval=-1", "checkpoint_interval=-1", "max_timesteps=8"] + many
    whole, _ = run_ranks("ref_HydroBlast3D", more, tmp_path / "w", nranks, 29731 + nranks + 20 * len(scheme))
    again, _ = run_ranks("ref_HydroBlast3D", more + [f"restartfile={outn}/chk00004"], tmp_path / "r", nranks, 29751 + nranks + 20 * len(scheme))
    for r in range(nranks):
        assert np.array_equal(whole[r], again[r]), r
    (single,), _ = run_ranks("ref_HydroBlast3D", base + ["plotfile_interval=-1", "checkpoint_interval=-1", "max_timesteps=8", f"restartfile={outn}/chk00004"] + single,
                             tmp_path / "s", 1, 29771 + 20 * len(scheme))
    from quokka_amd.simulation import chop_domain, distribute_boxes, distribute_boxes_interleaved
    boxes = chop_domain([32, 32, 32], [8, 8, 8])
    owner = (distribute_boxes_interleaved if scheme == "ancestor" else distribute_boxes)(boxes, nranks, [32, 32, 32], [8, 8, 8])
    single = single.reshape(len(boxes), 6, 8, 8, 8)
    cursor = [0] * nranks
    worst = 0.0
    for b, r in enumerate(owner):  # the level-0 state (which holds the averages of the finer levels) of the one-rank restart, to rounding
        chunk = whole[r][cursor[r]:cursor[r] + 6 * 512].reshape(6, 8, 8, 8)
        cursor[r] += 6 * 512
        for n in range(6):
            worst = max(worst, float(np.abs(chunk[n] - single[b][n]).max() / np.abs(single[:, n]).max()))
    assert all(cursor[r] == whole[r].size for r in range(nranks)) and worst <= 1e-13, worst


def test_sedov_on_several_gpus_over_rccl(tmp_path):
    """the production transport (ncclSend / ncclRecv on the library-owned stream, ncclAllReduce): needs one GPU per rank — skipped on a
    one-GPU box, where the test above covers everything but the transport itself"""
    import torch
    n = min(torch.cuda.device_count(), 8)
    n = 8 if n >= 8 else (4 if n >= 4 else (2 if n >= 2 else 1))
    if n < 2:
        pytest.skip("one GPU visible: RCCL needs one device per rank")
    from quokka_amd.simulation import chop_domain, distribute_boxes
    args = ["geometry.prob_lo=0 0 0", "geometry.prob_hi=1.2 1.2 1.2", "geometry.is_periodic=0 0 0", "amr.n_cell=64 64 64", "amr.max_grid_size=32",
            "max_timesteps=12"]
    (one,), _ = run_ranks("ref_HydroBlast3D", args, tmp_path, 1, 29631)
    parts, outs = run_ranks("ref_HydroBlast3D", args, tmp_path, n, 29631 + n, backend="rccl")
    owner = distribute_boxes(chop_domain([64, 64, 64], [32, 32, 32]), n, [64, 64, 64], [32, 32, 32])
    one = one.reshape(8, 6, 32, 32, 32)
    cursor = [0] * n
    for b, r in enumerate(owner):
        chunk = parts[r][cursor[r]:cursor[r] + 6 * 32 ** 3].reshape(6, 32, 32, 32)
        cursor[r] += 6 * 32 ** 3
        assert np.array_equal(chunk, one[b]), (b, r)


def _level0_state_by_box(parts, owner, nb, n3, mgs):
    """per-rank dumps (this rank's level-0 boxes in global order) -> one array indexed by global box"""
    out = np.empty((nb, 6, mgs, mgs, mgs))
    cursor = [0] * len(parts)
    for b, r in enumerate(owner):
        out[b] = parts[r][cursor[r]:cursor[r] + 6 * n3].reshape(6, mgs, mgs, mgs)
        cursor[r] += 6 * n3
    assert all(cursor[r] == parts[r].size for r in range(len(parts)))
    return out


@pytest.mark.parametrize("nranks", [2, 4, 8])
def test_amr_levels_with_their_own_distribution_in_the_cxx_host(tmp_path, ctx, nranks):
    """qk.distribute_levels = 1 (quokka_amr.hpp Shadow / RingIncrement, qk_pcopy.hpp, qk_grid_layout.hpp): every level of the three-level Sedov
    hierarchy has its own box -> rank map — grids clustered globally as with one rank, a level with fewer boxes than ranks chopped (AMReX's
    refine_grid_layout), boxes dealt along the space-filling curve to the least loaded ranks —, the parent's data reach the fine boxes, averaged-down
    and refluxed data the coarse boxes through ParallelCopy plans.  Against ONE rank chopping for the same box count, in the ordinary data path
    and in the distributed one (every plan then has same-rank items only): same grids, zone updates, time steps; level-0 state (which holds the
    averages of every finer level) to rounding — the two parts of a flux register are added one after the other —; energy conserved; the finer
    levels live on more than one rank."""
    import re
    from quokka_amd.simulation import chop_domain, distribute_boxes
    mgs = 16
    args = ["geometry.prob_lo=0 0 0", "geometry.prob_hi=1.2 1.2 1.2", "geometry.is_periodic=0 0 0", "amr.n_cell=32 32 32", f"amr.max_grid_size={mgs}",
            "amr.max_level=2", "amr.blocking_factor=8", "amr.n_error_buf=3", "do_reflux=1", "max_timesteps=8", f"qk.refine_grid_layout_target={nranks}",
            "plotfile_interval=100", f"plotfile_prefix={tmp_path}/c/plt"]
    for sub in ("a", "b", "c"):
        os.makedirs(tmp_path / sub)
    (one,), outs1 = run_ranks("ref_HydroBlast3D", args, tmp_path / "a", 1, 29811)
    (shadowed,), outs2 = run_ranks("ref_HydroBlast3D", args + ["qk.distribute_levels=1"], tmp_path / "b", 1, 29813)
    parts, outs = run_ranks("ref_HydroBlast3D", args + ["qk.distribute_levels=1", "qk.level0_distribution=bricks"], tmp_path / "c", nranks, 29815 + nranks)
    zone = re.compile(r"Zone-updates on level (\d): (\d+) \((\d+) grids\)")
    z1 = zone.findall(outs1[0])
    assert len(z1) == 3 and zone.findall(outs[0]) == z1 and zone.findall(outs2[0]) == z1, (zone.findall(outs[0]), zone.findall(outs2[0]), z1)
    assert int(z1[1][2]) >= 2 and int(z1[2][2]) >= 2, z1  # chopped (as far as the blocking factor allows: the refined region is 16^3 - 32^3 cells wide)
    per = {int(l): [int(x) for x in c.split()] for l, c in re.findall(r"Boxes of level (\d) per rank:((?: \d+)+)", outs[0])}
    assert set(per) == {0, 1, 2} and all(len(per[l]) == nranks for l in per)
    assert all(sum(1 for n in per[l] if n > 0) > 1 for l in (1, 2)), per
    boxes = chop_domain([32, 32, 32], [mgs] * 3)
    owner = distribute_boxes(boxes, nranks, [32, 32, 32], [mgs] * 3)
    nb, n3 = len(boxes), mgs ** 3
    one = one.reshape(nb, 6, mgs, mgs, mgs)
    shadowed = shadowed.reshape(nb, 6, mgs, mgs, mgs)
    many = _level0_state_by_box(parts, owner, nb, n3, mgs)
    for name, got in (("one rank, distributed data path", shadowed), (f"{nranks} ranks", many)):
        worst = max(float(np.abs(got[:, n] - one[:, n]).max() / np.abs(one[:, n]).max()) for n in range(6))
        assert worst <= 1e-13, (name, worst)
    # the distributed data path itself does not depend on the number of ranks: same plans, the increments of a coarse cell added in the order of the plan's groups
    assert np.array_equal(shadowed, many)
    for o in (outs1[0], outs2[0], outs[0]):
        assert "Energy conservation is OK." in o
    metas = [open(str(tmp_path / d / f) + ".meta").read().split()[:3] for d, f in (("a", "state_n1_shm.bin"), ("b", "state_n1_shm.bin"), ("c", f"state_n{nranks}_shm.bin.rank0"))]
    assert metas[0] == metas[1] == metas[2], metas  # steps, time, dt
    # the PYTHON host on one rank, chopping for the same count: the N-rank plotfile of the C++ host (one data file per rank; the last of the three runs above
    # wrote it) holds the same grids on every level and the same data to rounding — the two hosts build the same distributed hierarchy (chopGrids /
    # distributeSfc are pinned function by function in tests/test_amr_grids.py)
    from quokka_amd import plotfile
    from quokka_amd.amr_simulation import sedov_amr_problem
    amr = sedov_amr_problem(ctx, 32, 2, max_grid_size=mgs, blocking_factor=8, refine_grid_layout_target=nranks)
    for _ in range(8):
        amr.step()
    plotfile.WritePlotFile(amr, str(tmp_path / "py_plt00008"))
    cpp, py = plotfile.read_plotfile(str(tmp_path / "c" / "plt00008")), plotfile.read_plotfile(str(tmp_path / "py_plt00008"))
    assert cpp.finest_level == py.finest_level == 2 and (cpp.time, cpp.level_steps) == (py.time, py.level_steps)
    for l, (la, lb) in enumerate(zip(cpp.levels, py.levels)):
        assert la.boxes == lb.boxes, (l, la.boxes, lb.boxes)
    diff = plotfile.compare_plotfiles(str(tmp_path / "c" / "plt00008"), str(tmp_path / "py_plt00008"))
    scale = {v: max(float(np.abs(f[i]).max()) for f in py.levels[0].fabs) for i, v in enumerate(py.varnames)}
    assert all(diff[v] <= 1e-13 * max(scale[v], scale["gasEnergy"]) for v in py.varnames), diff
    if nranks == 4:
        # ... and the PYTHON host on the same number of ranks (gloo, the ranks sharing this GPU; tests/test_multirank_one_gpu.py): both hosts drive the same plans in the
        # same order — level 0 agrees in every bit, rank by rank
        import torch.multiprocessing as mp
        from test_multirank_one_gpu import collect, free_port, run_amr_worker
        mpctx = mp.get_context("spawn")
        q = mpctx.Queue()
        port = free_port()
        procs = [mpctx.Process(target=run_amr_worker, args=(r, nranks, port, 32, 8, q, "bricks", mgs)) for r in range(nranks)]
        for p in procs:
            p.start()
        results = sorted(collect(procs, q, nranks), key=lambda r: r[0])
        pyl0 = np.full((6, 32, 32, 32), np.nan)
        for rank, levels, tnew, drift, istep in results:
            all_boxes, own, mine, vals = levels[0]
            assert own == owner
            for (lo, hi), v in zip(mine, vals):
                pyl0[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = v
        cxx = np.full_like(pyl0, np.nan)
        for b, (lo, hi) in enumerate(boxes):
            cxx[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = many[b].reshape(6, mgs, mgs, mgs)
        assert not np.isnan(pyl0).any() and np.array_equal(cxx, pyl0), float(np.abs(cxx - pyl0).max())


@pytest.mark.parametrize("name,deck,extra,n_cell,mgs,steps", [("ref_Advection2D", "advection2d_amr.in", [], (64, 64), 16, 40),
                                                              ("ref_RadBeam", "beam.in", ["amr.max_grid_size=32"], (128, 128), 32, 8)],
                         ids=["advection_periodic_four_levels", "radiation_custom_boundaries"])
def test_distributed_levels_of_other_solvers_in_the_cxx_host(tmp_path, monkeypatch, name, deck, extra, n_cell, mgs, steps):
    """qk.distribute_levels = 1 beyond the hydro blast, with the reference's own decks on 2-D hierarchies: Advection2D (AdvectionSimulation levels,
    four levels, periodic: the ParallelCopy plans and the nesting across the periodic faces, both RK stages feeding the two-part register) and
    RadBeam (radiation only: the second register of a level with its own ring, the problem's custom boundary function evaluated on the coarse
    patch a child keeps on its rank).  One rank in the ordinary data path against one rank and four ranks in the distributed one, all chopping
    for four boxes: same grids, same time steps, level-0 state to rounding.  (Eight coarse steps of the beam: the two parts of a register are added
    to the state one after the other, and the sharp front of the streaming beam amplifies that rounding — 1e-16 after three coarse steps, 4e-9 after
    twenty, measured per level with profiles/tools/compare_plotfiles.py.)"""
    import re
    from quokka_amd.simulation import chop_domain, distribute_boxes
    monkeypatch.setenv("QK_MAX_COARSE_STEPS", str(steps))
    nranks = 4
    args = [os.path.join(HOST, "decks", deck), "plotfile_interval=-1", "checkpoint_interval=-1", f"qk.refine_grid_layout_target={nranks}"] + extra
    for sub in ("a", "b", "c"):
        os.makedirs(tmp_path / sub)
    port = 29851 + 10 * len(name)  # (the rendezvous files of two tests must not meet)
    (one,), outs1 = run_ranks(name, args, tmp_path / "a", 1, port)
    (shadowed,), outs2 = run_ranks(name, args + ["qk.distribute_levels=1"], tmp_path / "b", 1, port + 1)
    parts, outs = run_ranks(name, args + ["qk.distribute_levels=1", "qk.level0_distribution=bricks"], tmp_path / "c", nranks, port + 2)
    zone = re.compile(r"Zone-updates on level (\d): (\d+) \((\d+) grids\)")
    z1 = zone.findall(outs1[0])
    assert len(z1) >= 3 and all(int(u) > 0 for _, u, _ in z1) and zone.findall(outs[0]) == z1 and zone.findall(outs2[0]) == z1, (zone.findall(outs[0]), z1)
    per = {int(l): [int(x) for x in c.split()] for l, c in re.findall(r"Boxes of level (\d) per rank:((?: \d+)+)", outs[0])}
    assert all(sum(1 for n in per[l] if n > 0) > 1 for l in per), per  # every level lives on more than one rank
    boxes = chop_domain([n_cell[0], n_cell[1], 1], [mgs, mgs, mgs])
    owner = distribute_boxes(boxes, nranks, [n_cell[0], n_cell[1], 1], [mgs, mgs, mgs])
    nb, cells = len(boxes), mgs * mgs
    ncomp = one.size // (nb * cells)
    assert one.size == nb * ncomp * cells and sum(p.size for p in parts) == one.size
    many, cursor = [], [0] * nranks
    for r in owner:
        many.append(parts[r][cursor[r]:cursor[r] + ncomp * cells])
        cursor[r] += ncomp * cells
    assert all(cursor[r] == parts[r].size for r in range(nranks))
    one = one.reshape(nb, ncomp, cells)
    assert np.isfinite(one).all()
    for label, got in (("one rank, distributed data path", shadowed.reshape(nb, ncomp, cells)), ("four ranks", np.concatenate(many).reshape(nb, ncomp, cells))):
        for n in range(ncomp):
            scale = np.abs(one[:, n]).max()
            assert np.abs(got[:, n] - one[:, n]).max() <= 1e-12 * scale, (label, n, float(np.abs(got[:, n] - one[:, n]).max()), float(scale))
    metas = [open(str(tmp_path / d / f) + ".meta").read().split()[:3] for d, f in (("a", "state_n1_shm.bin"), ("b", "state_n1_shm.bin"), ("c", f"state_n{nranks}_shm.bin.rank0"))]
    assert metas[0] == metas[1] == metas[2], metas  # steps, time, dt
