#!/usr/bin/env python3
"""Writes the forwarding headers that give the reference's include names a target in this tree (build artefacts under .shims/, not tracked):
the AMReX header names its problem files include -> amrex_mini.hpp; the reference's own header names -> quokka_host.hpp / compat headers.
usage: make_shims.py <outdir>"""
import os
import sys

AMREX = """AMReX.H AMReX_Algorithm.H AMReX_Arena.H AMReX_Array.H AMReX_Array4.H AMReX_BCRec.H AMReX_BC_TYPES.H AMReX_BLProfiler.H AMReX_BLassert.H AMReX_Box.H
AMReX_BoxArray.H AMReX_Config.H AMReX_CoordSys.H AMReX_DistributionMapping.H AMReX_Extension.H AMReX_FArrayBox.H AMReX_FabArray.H AMReX_FabArrayBase.H
AMReX_FabArrayUtility.H AMReX_Geometry.H AMReX_GpuAsyncArray.H AMReX_GpuContainers.H AMReX_GpuDevice.H AMReX_GpuQualifiers.H AMReX_IntVect.H AMReX_Loop.H
AMReX_MFParallelFor.H AMReX_MultiFab.H AMReX_MultiFabUtil.H AMReX_ParallelContext.H AMReX_ParallelDescriptor.H AMReX_ParmParse.H AMReX_PlotFileUtil.H AMReX_Print.H
AMReX_REAL.H AMReX_Reduce.H AMReX_SPACE.H AMReX_TableData.H AMReX_TagBox.H AMReX_ValLocPair.H AMReX_Vector.H AMReX_iMultiFab.H AMReX_GpuControl.H AMReX_Gpu.H
AMReX_Random.H AMReX_RandomEngine.H AMReX_ccse-mpi.H AMReX_Particles.H AMReX_AmrParticles.H AMReX_ParIter.H AMReX_ParticleInterpolators.H AMReX_ParticleReal.H AMReX_GpuLaunch.H AMReX_Utility.H AMReX_INT.H AMReX_Dim3.H AMReX_RealBox.H AMReX_Math.H""".split()
QUOKKA = ["QuokkaSimulation.hpp", "simulation.hpp", "SimulationData.hpp", "hydro/mhd_system.hpp", "physics_numVars.hpp", "physics_info.hpp", "hydro/hydro_system.hpp", "hydro/EOS.hpp", "hydro/HydroState.hpp",
          "radiation/radiation_system.hpp", "radiation/radiation_dust_system.hpp", "fundamental_constants.H", "hyperbolic_system.hpp", "grid.hpp", "math/math_impl.hpp",
          "cooling/TabulatedCooling.hpp", "cooling/GrackleLikeCooling.hpp"]
COMPAT = {"util/fextract.hpp": "compat/util_compat.hpp", "util/ArrayUtil.hpp": "compat/util_compat.hpp", "util/valarray.hpp": "compat/util_compat.hpp",
          "fmt/format.h": "compat/mini_fmt.hpp", "fmt/core.h": "compat/mini_fmt.hpp", "radiation/planck_integral.hpp": "compat/planck_integral.hpp",
          "hydro/NSCBC_inflow.hpp": "compat/nscbc.hpp", "hydro/NSCBC_outflow.hpp": "compat/nscbc.hpp",
          "math/ODEIntegrate.hpp": "compat/ode_integrate.hpp", "math/quadrature.hpp": "compat/quadrature.hpp",
          "turbulence/TurbDataReader.hpp": "compat/turb_data_reader.hpp", "particles/CICParticles.hpp": "compat/particles_decl.hpp",
          "eos.H": "compat/microphysics_stub.hpp", "extern_parameters.H": "compat/microphysics_stub.hpp"}
ADVECTION = ["linear_advection/AdvectionSimulation.hpp", "linear_advection/linear_advection.hpp"]
EMPTY = ["util/matplotlibcpp.h"]


def main():
    out = sys.argv[1]
    up = lambda name: "../" * (name.count("/") + 1)

    def write(name, body):
        path = os.path.join(out, name)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        if not os.path.exists(path) or open(path).read() != body:
            open(path, "w").write(body)

    for n in AMREX:
        write(n, f'#include "{up(n)}amrex_mini.hpp"\n')
    for n in QUOKKA:
        write(n, f'#include "{up(n)}quokka_host.hpp"\n#include "{up(n)}quokka_amr.hpp"\n')
    for n in ADVECTION:
        write(n, f'#include "{up(n)}quokka_advection.hpp"\n')
    for n, target in COMPAT.items():
        write(n, f'#include "{up(n)}{target}"\n')
    for n in EMPTY:
        write(n, "// (plots are made under HAVE_PYTHON only)\n")


if __name__ == "__main__":
    main()
