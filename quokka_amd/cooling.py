"""Optically-thin cooling from Cloudy tables as a Strang-split source: the host side of quokka::TabulatedCooling (reference
src/cooling/TabulatedCooling.hpp, TabulatedCooling.cpp, CloudyDataReader.cpp) over the library's kernels (csrc/qk_cooling.hip).

    tables = CloudyTables(ctx, "isrf_1000Go_grains.h5")          # readCloudyData (cooling.hdf5_data_file of the deck)
    sim.add_strang_source(TabulatedCooling(sim, tables))         # cooling.enabled = 1, cooling.cooling_table_type = cloudy_cooling_tools
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import capi, comm

TGAS_FROM_EGAS, EGAS_FROM_TGAS, MMW, COOLING_LENGTH, NET_HEATING = range(5)
MAX_SUBSTEPS = 2000  # maxStepsODEIntegrate (src/math/ODEIntegrate.hpp:120)


class CloudyTables:
    """cloudy_tables (TabulatedCooling.hpp:54-73): the five arrays in device memory + the temperature / mean-molecular-weight ranges"""

    def __init__(self, ctx, path: str):
        host = capi.CloudyTables()
        ctx.check(ctx.L.qk_cloudy_tables_read(ctx.h, str(path).encode(), C.byref(host)), "qk_cloudy_tables_read")
        try:
            n0, n1 = host.n_nH, host.n_Tgas
            take = lambda p, n: np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(n,)).copy()
            self.host = {"log_nH": take(host.log_nH, n0), "log_Tgas": take(host.log_Tgas, n1), "cooling": take(host.cooling, n0 * n1),
                         "heating": take(host.heating, n0 * n1), "mean_mol_weight": take(host.mean_mol_weight, n0 * n1)}
            self.n_nH, self.n_Tgas = n0, n1
            self.T_min, self.T_max, self.mmw_min, self.mmw_max = host.T_min, host.T_max, host.mmw_min, host.mmw_max
        finally:
            ctx.L.qk_cloudy_tables_free(C.byref(host))
        self.dev = {k: torch.from_numpy(v).to(ctx.device) for k, v in self.host.items()}
        d = self.dev
        self.c = capi.CloudyTables(d["log_nH"].data_ptr(), d["log_Tgas"].data_ptr(), d["cooling"].data_ptr(), d["heating"].data_ptr(),
                                   d["mean_mol_weight"].data_ptr(), n0, n1, self.T_min, self.T_max, self.mmw_min, self.mmw_max)


class TabulatedCooling:
    """computeCooling<problem_t> as addStrangSplitSourcesWithBuiltin calls it (src/QuokkaSimulation.hpp:520-547): returns False when the
    integration failed in some cell (the hydro step is retried with a smaller dt)"""

    def __init__(self, sim, tables: CloudyTables, T_floor=None):
        self.sim, self.tables = sim, tables
        self.T_floor = float(sim.tempFloor_ if T_floor is None else T_floor)
        self.counters = torch.zeros(2, dtype=torch.int64, device=sim.ctx.device)
        self.last = (0.0, 0)  # (average, maximum) substeps per cell of the last call, as the reference prints them

    def __call__(self, state, time: float, dt: float) -> bool:
        s = self.sim
        c = s.ctx
        c.check(c.L.qk_clear_bytes(c.h, c.stream(), C.c_void_p(self.counters.data_ptr()), 16), "qk_clear_bytes")
        c.check(c.L.qk_cooling_tabulated(s.lev.h, c.stream(), C.byref(s.traits), state.ptr, C.byref(self.tables.c), float(dt), self.T_floor,
                                         C.c_void_p(self.counters.data_ptr())), "qk_cooling_tabulated")
        t = self.counters.clone()
        if s.nranks > 1:  # nsubstepsMF.max / sum are over all ranks
            mx, sm = t[0:1].clone(), t[1:2].clone()
            comm.all_reduce(mx, dist.ReduceOp.MAX)
            comm.all_reduce(sm, dist.ReduceOp.SUM)
            t = torch.cat([mx, sm])
        nmax, nsum = int(t[0].item()), int(t[1].item())
        self.last = (nsum / max(s.CountCells(), 1), nmax)
        return nmax < MAX_SUBSTEPS

    def evaluate(self, what: int, rho: torch.Tensor, value: torch.Tensor) -> torch.Tensor:
        """the per-cell functions of TabulatedCooling.hpp:82-220 over arrays (see qk_cooling_evaluate)"""
        c = self.sim.ctx
        rho = rho.contiguous().to(torch.float64)
        value = value.contiguous().to(torch.float64)
        out = torch.empty_like(rho)
        c.check(c.L.qk_cooling_evaluate(c.h, c.stream(), C.byref(self.tables.c), float(self.sim.traits.gamma), int(what), rho.numel(),
                                        C.c_void_p(rho.data_ptr()), C.c_void_p(value.data_ptr()), C.c_void_p(out.data_ptr())), "qk_cooling_evaluate")
        return out
