"""Tabulated (Cloudy) cooling on the CPU: the oracle (oracle/cooling.hpp, the reference's src/cooling/TabulatedCooling.hpp + its math headers restated)
against properties the algorithm must have, the library's HDF5 table reader against an independent reader, and the HOST side of the functions the
kernels are made of (quokka_amd/csrc/qk_cooling_device.hpp is host + device code: the reference's problem files call these functions on the host
too, e.g. src/problems/ShockCloud/cloud.cpp:737-745) against the oracle, bit for bit.

The table is the reference's own data file (extern/cooling/isrf_1000Go_grains.h5, the file tests/ShockCloud_*.in name), committed as a fixture.
The oracle's parity against the reference BINARY is unpinned for this module (see the header of oracle/cooling.hpp)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from mini_hdf5 import cloudy_file_arrays
from oracle.pyoracle import OracleCloudy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "tests", "golden", "isrf_1000Go_grains.h5")
GAMMA = 5.0 / 3.0
M_H = 1.67262192369e-24 + 9.1093837015e-28
K_B = 1.380649e-16
X_H = 1.0 / (1.0 + 0.098 * 3.971)


@pytest.fixture(scope="module")
def arrays():
    return cloudy_file_arrays(TABLE)


@pytest.fixture(scope="module")
def orc(arrays):
    return OracleCloudy(arrays)


def sample(n, seed=1):
    r = np.random.default_rng(seed)
    return 10 ** r.uniform(-27.5, -19.5, n), 10 ** r.uniform(0.8, 9.2, n)  # beyond the table on every side


def test_table_axes_and_ranges_of_the_reference_data_file(arrays, orc):
    assert arrays["Cooling"].shape == (25, 161)
    assert np.allclose(arrays["Parameter1"], np.linspace(-6.0, 6.0, 25), rtol=0, atol=1e-12)  # log10 n_H
    assert arrays["Temperature"][0] == 10.0 and arrays["Temperature"][-1] == 1.0e9
    T_min, T_max, mmw_min, mmw_max = orc.ranges()
    assert (T_min, T_max) == (10.0, 1.0e9)
    assert (mmw_min, mmw_max) == (float(arrays["MMW"].min()), float(arrays["MMW"].max()))
    assert 0.59 < mmw_min < 0.63 and 1.2 < mmw_max < 1.3  # fully ionised ... neutral gas of ISM abundances
    # the prepared temperature axis is log10 T, the rates are FastMath::log10 of rate / (1.67e-24)^2
    assert np.array_equal(orc.prepared(1), np.log10(arrays["Temperature"]))
    m, e = np.frexp(arrays["Cooling"] / (1.67e-24 * 1.67e-24))
    assert np.array_equal(orc.prepared(2).reshape(161, 25), (0.301029995663981195 * (2 * (m - 1) + e)).T)


def test_library_reader_equals_the_independent_reader_and_the_oracle_s_preparation(arrays, orc):
    """qk_cloudy_tables_read (the library's own HDF5 reader + the transformations of CloudyDataReader.cpp) delivers the oracle's prepared arrays in
    every bit; the oracle was given the datasets by tests/mini_hdf5.py — two independent readers of the file format"""
    from quokka_amd import capi
    L = capi.lib()
    ctx = C.c_void_p()
    capi.check(None, L.qk_ctx_create(C.byref(ctx), capi.QK_DEVICE_HOST_PLANNING if hasattr(capi, "QK_DEVICE_HOST_PLANNING") else -1), "qk_ctx_create")
    t = capi.CloudyTables()
    capi.check(ctx, L.qk_cloudy_tables_read(ctx, TABLE.encode(), C.byref(t)), "qk_cloudy_tables_read")
    try:
        assert (t.n_nH, t.n_Tgas) == (25, 161)
        assert (t.T_min, t.T_max, t.mmw_min, t.mmw_max) == orc.ranges()
        take = lambda p, n: np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(n,)).copy()
        for which, (p, n) in enumerate([(t.log_nH, 25), (t.log_Tgas, 161), (t.cooling, 25 * 161), (t.heating, 25 * 161), (t.mean_mol_weight, 25 * 161)]):
            assert np.array_equal(take(p, n), orc.prepared(which)), which
    finally:
        L.qk_cloudy_tables_free(C.byref(t))
    # a file that is not HDF5 is refused with a message, not read
    assert L.qk_cloudy_tables_read(ctx, os.path.join(ROOT, "tests", "golden", "dust_shell_initial_conditions.txt").encode(), C.byref(t)) != 0
    assert b"HDF5" in L.qk_last_error(ctx)
    L.qk_ctx_destroy(ctx)


def test_truncated_and_corrupted_table_files_are_refused_not_read_out_of_bounds(tmp_path):
    """qk_hdf5_mini.hpp trusts no size it reads from the file: the table file cut short at many lengths, and with bytes of its header region
    overwritten (sizes of attribute / layout / dataspace messages, heap offsets), either still reads — the damage missed everything the reader
    looks at — or comes back as an error with a message.  (A reader that ran past its buffer would crash the test process.)"""
    from quokka_amd import capi
    L = capi.lib()
    ctx = C.c_void_p()
    capi.check(None, L.qk_ctx_create(C.byref(ctx), -1), "qk_ctx_create")
    data = open(TABLE, "rb").read()
    t = capi.CloudyTables()
    refused = 0
    cuts = sorted(set([0, 7, 8, 95, 96, 200, 511, 1024, 2048, 4096, len(data) // 2, len(data) - 4096, len(data) - 1] + list(range(600, 6000, 397))))
    for n in cuts:
        f = tmp_path / f"cut_{n}.h5"
        f.write_bytes(data[:n])
        rc = L.qk_cloudy_tables_read(ctx, str(f).encode(), C.byref(t))
        if rc == 0:
            L.qk_cloudy_tables_free(C.byref(t))
        else:
            refused += 1
            assert L.qk_last_error(ctx)
    assert refused >= len(cuts) - 2
    rng = np.random.default_rng(7)
    outcomes = [0, 0]
    for trial in range(150):  # header / heap / B-tree / object-header region: the first 8 KiB hold every size field the reader follows
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 4))):
            at = int(rng.integers(8, 8192))
            b[at] = int(rng.choice([0xFF, 0x7F, 0x00, int(rng.integers(0, 256))]))
        f = tmp_path / "corrupt.h5"
        f.write_bytes(bytes(b))
        rc = L.qk_cloudy_tables_read(ctx, str(f).encode(), C.byref(t))
        outcomes[1 if rc else 0] += 1
        if rc == 0:
            L.qk_cloudy_tables_free(C.byref(t))
    assert outcomes[1] > 0, outcomes
    L.qk_ctx_destroy(ctx)


def test_temperature_energy_round_trip_and_limits(orc):
    rho, T = sample(20000)
    E = orc.evaluate(orc.EGAS_FROM_TGAS, rho, T, GAMMA)
    # E = n k_B T / (gamma - 1) with n = rho / (m_H mu), mu within the table's range
    mu = rho * K_B * np.clip(T, None, None) / (M_H * E * (GAMMA - 1.0))
    assert mu.min() >= 0.614 and mu.max() <= 1.2883
    Tback = orc.evaluate(orc.TGAS_FROM_EGAS, rho, E, GAMMA)
    assert not np.isnan(Tback).any()
    inside = (T > 10.0) & (T < 1.0e9)
    # the bracket is closed to a relative width of 1e-5 and its midpoint returned; mu(T) is not monotonic everywhere, so where mu C = T has
    # several roots any of them is a valid answer: compare energies, not temperatures
    Eback = orc.evaluate(orc.EGAS_FROM_TGAS, rho[inside], Tback[inside], GAMMA)
    assert np.max(np.abs(Eback / E[inside] - 1.0)) < 3.0e-5
    assert np.all(Tback[T <= 10.0] == 10.0) and np.all(Tback[T >= 1.0e9] == 1.0e9)  # outside the table: its ends (TabulatedCooling.hpp:124-129)


def test_cooling_relaxes_to_thermal_equilibrium_and_equilibrium_is_a_fixed_point(orc):
    """n_H = 1 cm^-3 under the table's radiation field: hot gas cools, cold gas is heated, both end at the temperature where heating balances
    cooling; gas started there stays there"""
    rho = np.full(3, M_H / X_H)  # n_H = 1
    Tgrid = 10 ** np.linspace(1.2, 8.0, 4000)
    net = orc.evaluate(orc.NET_HEATING, np.full_like(Tgrid, rho[0]), Tgrid, GAMMA)
    crossings = np.where((net[:-1] > 0) & (net[1:] <= 0))[0]
    assert len(crossings) >= 1  # heating below, cooling above: a stable equilibrium
    T_eq = Tgrid[crossings[-1]]
    T0 = np.array([1.0e6, 30.0, T_eq])
    E0 = orc.evaluate(orc.EGAS_FROM_TGAS, rho, T0, GAMMA)
    U = np.zeros((6, 3))
    U[0], U[4], U[5] = rho, E0, E0
    t_cool = np.abs(E0 / orc.evaluate(orc.NET_HEATING, rho, T0, GAMMA))
    span = 5.0 * float(np.max(t_cool[:2]))  # (the explicit integrator resolves the relaxation time: ~10 substeps per cooling time once there)
    U1, ns = orc.compute_cooling(U, GAMMA, span, 10.0)
    assert ns.max() < 2000 and ns[2] < ns[1] < ns[0]  # converged; the cell that starts in equilibrium needs the fewest substeps
    T1 = orc.evaluate(orc.TGAS_FROM_EGAS, rho, U1[4], GAMMA)
    assert np.all(np.abs(T1 / T_eq - 1.0) < 0.01) and np.ptp(T1) / T_eq < 1.0e-3, (T1, T_eq)
    assert np.array_equal(U1[4] - E0, U1[5] - E0)  # gas energy and auxiliary internal energy take the same change
    # momentum is untouched and only the internal part of the energy cools
    U[1:4] = 1.0e-3 * rho * 1.0e5
    U[4] = E0 + 0.5 * (U[1] ** 2 + U[2] ** 2 + U[3] ** 2) / rho
    U2, _ = orc.compute_cooling(U, GAMMA, span, 10.0)
    assert np.array_equal(U2[:4], U[:4])
    assert np.allclose(U2[4] - U[4], U1[4] - E0, rtol=1e-9)


HOSTCHECK = r"""
#include "qk_cooling_device.hpp"
extern "C" {
struct T5 { const double *a, *b, *c, *d, *e; int n0, n1; double tmin, tmax, mmin, mmax; };
static qk::cool::Tables mk(const T5 *t)
{
	qk::cool::Tables r{};
	r.log_nH = t->a; r.log_T = t->b; r.cool = t->c; r.heat = t->d; r.mmw = t->e; r.n_nH = t->n0; r.n_T = t->n1;
	r.T_min = t->tmin; r.T_max = t->tmax; r.mmw_min = t->mmin; r.mmw_max = t->mmax;
	r.m_H = 1.67262192369e-24 + 9.1093837015e-28; r.k_B = 1.380649e-16; r.prepared = 0;
	return r;
}
void hc_eval(const T5 *t, double gamma, int what, long n, const double *rho, const double *val, double *out)
{
	auto tab = mk(t);
	for (long i = 0; i < n; ++i) {
		switch (what) {
		case 0: out[i] = qk::cool::tgasFromEgas(tab, rho[i], val[i], gamma); break;
		case 1: out[i] = qk::cool::egasFromTgas(tab, rho[i], val[i], gamma); break;
		case 2: out[i] = qk::cool::meanMolecularWeight(tab, rho[i], val[i], gamma); break;
		case 3: out[i] = qk::cool::coolingLength(tab, rho[i], val[i], gamma); break;
		default: out[i] = qk::cool::netHeating(tab, rho[i], val[i]);
		}
	}
}
void hc_cool(const T5 *t, double gamma, double dt, double Tfloor, long n, const double *rho, double *E, int *ns)
{
	auto tab = mk(t);
	for (long i = 0; i < n; ++i) {
		double abstol = 0.01 * qk::cool::egasFromTgas(tab, rho[i], Tfloor, gamma);
		ns[i] = qk::cool::integrateCooling(tab, rho[i], gamma, E[i], dt, 1e-4, abstol);
	}
}
}
"""


def test_host_side_of_the_kernel_functions_equals_the_oracle_bit_for_bit(orc, tmp_path):
    src = tmp_path / "hostcheck.cpp"
    src.write_text(HOSTCHECK)
    so = tmp_path / "libhostcheck.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "quokka_amd", "csrc"), str(src), "-o", str(so)])
    H = C.CDLL(str(so))
    DP = C.POINTER(C.c_double)

    class T5(C.Structure):
        _fields_ = [(k, DP) for k in "abcde"] + [("n0", C.c_int), ("n1", C.c_int)] + [(k, C.c_double) for k in ("tmin", "tmax", "mmin", "mmax")]

    prep = [orc.prepared(w) for w in range(5)]
    dp = lambda a: a.ctypes.data_as(DP)
    t5 = T5(*[dp(a) for a in prep], 25, 161, *orc.ranges())
    n = 60000
    rho, T = sample(n, seed=7)

    def host(what, val):
        out = np.empty(n)
        H.hc_eval(C.byref(t5), C.c_double(GAMMA), what, C.c_long(n), dp(rho), dp(np.ascontiguousarray(val)), dp(out))
        return out

    E = orc.evaluate(orc.EGAS_FROM_TGAS, rho, T, GAMMA)
    assert np.array_equal(host(1, T), E)
    assert np.array_equal(host(4, T), orc.evaluate(orc.NET_HEATING, rho, T, GAMMA))
    for what in (orc.TGAS_FROM_EGAS, orc.MMW, orc.COOLING_LENGTH):  # through Algorithm 748: the same brackets, the same midpoints
        assert np.array_equal(host(what, E), orc.evaluate(what, rho, E, GAMMA), equal_nan=True), what
    # the adaptive integration: the same substeps, the same energies
    m = 6000
    U = np.zeros((6, m))
    U[0], U[4], U[5] = rho[:m], E[:m], E[:m]
    dt = 3.15e7 * 1.0e4
    U1, ns = orc.compute_cooling(U, GAMMA, dt, 10.0)
    Eh, nh = E[:m].copy(), np.zeros(m, dtype=np.int32)
    H.hc_cool(C.byref(t5), C.c_double(GAMMA), C.c_double(dt), C.c_double(10.0), C.c_long(m), dp(np.ascontiguousarray(rho[:m])), dp(Eh), nh.ctypes.data_as(C.POINTER(C.c_int)))
    assert np.array_equal(nh, ns) and ns.max() > 100  # (some cells need hundreds of substeps)
    assert np.array_equal(E[:m] + (Eh - E[:m]), U1[5])
