"""The problem hooks as COMPILED device code (SURVEY §8(b) option (ii)): a problem file built against the C++ host gets the single-group
source-term kernel instantiated in its own translation unit with its ComputePlanckOpacity / ComputeEnergyMeanOpacity / ComputeFluxMeanOpacity
(+ emission, ISM and quokka::EOS hooks) called inside the Newton-Raphson iteration — nothing is sampled into closed sets.  Pin: an opacity
law no closed set holds, kappa = kappa0 rho^0.3 (T / T0)^-1.7, against the CPU oracle running the same expression as a std::function."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "quokka_amd", "host")


def test_general_opacity_law_matches_the_oracle(tmp_path, oracle):
    from oracle.pyoracle import GENERAL_OPACITY
    exe = os.path.join(HOST, "bin", "general_opacity")
    assert os.path.exists(exe), "quokka_amd/host/bin/general_opacity is built by __graft_entry__.build() (make -C quokka_amd/host)"
    dump = str(tmp_path / "state.bin")
    p = subprocess.run([exe, os.path.join(HOST, "decks", "general_opacity.in"), f"qk.dump_state={dump}"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    meta = [float(x) for x in open(dump + ".meta").read().split()]
    got = np.fromfile(dump, dtype=np.float64).reshape(10, 64)
    so = oracle.sim(GENERAL_OPACITY, 1, [64, 1, 1], [0, 0, 0], [64.0, 1, 1], [1, 1, 1], max_grid_size=[64, 1, 1])
    U0 = so.valid(0).reshape(10, 64).copy()
    assert so.evolve()
    want = so.valid(0).reshape(10, 64)
    assert int(meta[0]) == so.istep == 40 and abs(meta[1] - so.time) <= 1e-14 * so.time
    # the run does something: gas and radiation exchange energy (the opacity matters), the flow advects
    assert np.abs(want[5] - U0[5]).sum() > 0.05 * np.abs(U0[5]).sum() and np.abs(want[6] - U0[6]).sum() > 0.05 * np.abs(U0[6]).sum()
    worst = 0.0
    for n in (0, 1, 4, 5, 6, 7):  # (y / z momenta and fluxes are zero)
        worst = max(worst, float(np.abs(got[n] - want[n]).sum() / np.abs(want[n]).sum()))
    assert not got[2:4].any() and not got[8:].any()
    print(f"compiled opacity hooks vs oracle: worst relative L1 = {worst:.2e}")
    assert worst <= 1e-12  # std::pow / sin / cos: device libm vs glibc (an ulp), the tolerance of the parity contract
    c = so.rad_counters()
    assert f"{c['solves']} solves" in p.stdout or "solves" not in p.stdout


def test_library_entry_refuses_hooks_it_cannot_evaluate(ctx):
    """the C-ABI's own AddSourceTermsSingleGroup carries closed hook sets only: asked for a compiled hook it must say so, not guess"""
    import ctypes as C
    from quokka_amd import capi
    from quokka_amd.radhydro import ShellConstants as S
    from quokka_amd.multifab import Level, MultiFab
    lev = Level(ctx, 3, [([0, 0, 0], [7, 7, 7])])
    U, Q = MultiFab(lev, 10, 4, fill=1.0), MultiFab(lev, 1, 0, fill=0.0)
    cnt = __import__("torch").zeros(8, dtype=__import__("torch").int32, device=ctx.device)
    t = capi.traits(5. / 3., False, 3, mean_molecular_weight=capi.M_U, boltzmann_constant=capi.K_B)
    rt = capi.RadTraits(S.c, S.chat, S.a_rad, 0.0, 1, capi.HOOK_COMPILED, float("nan"), float("nan"), float("nan"), 0)
    rc = ctx.L.qk_rad_AddSourceTermsSingleGroup(lev.h, ctx.stream(), C.byref(rt), C.byref(t), U.ptr, Q.ptr, C.c_double(1.0), 1,
                                                C.c_void_p(cnt.data_ptr()), C.c_void_p(cnt[4:].data_ptr()))
    assert rc == capi.ERR_UNSUPPORTED
    assert b"compiled device code" in ctx.L.qk_last_error(ctx.h)
