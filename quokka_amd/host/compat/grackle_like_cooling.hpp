// grackle_like_cooling.hpp — DECLARATIONS of quokka::GrackleLikeCooling (reference src/cooling/GrackleLikeCooling.hpp, GrackleDataReader.cpp), so that
// the problem files that name them compile and link unchanged.  The module itself is not built: its tables are Grackle's data files
// (extern/grackle_data_files, an empty submodule of the reference tree — no file to read, nothing to verify against), so readGrackleData refuses,
// and the per-cell functions, which can only be reached with tables, return NaN.  The Cloudy tables of the cloudy_cooling_tools ARE built:
// compat/tabulated_cooling.hpp.
#ifndef QK_HOST_COMPAT_GRACKLE_LIKE_COOLING_HPP_
#define QK_HOST_COMPAT_GRACKLE_LIKE_COOLING_HPP_

#include <limits>
#include <string>

#include "../quokka_rad_system.hpp"

namespace quokka::GrackleLikeCooling
{

constexpr double cloudy_H_mass_fraction = 1. / (1. + 0.1 * 3.971); // GrackleLikeCooling.hpp:36 (Grackle's default abundances: n_He / n_H = 0.1)

struct grackleGpuConstTables {
	amrex::Real T_min = 0, T_max = 0, mmw_min = 0, mmw_max = 0;
};

class grackle_tables
{
      public:
	amrex::Real T_min = 0, T_max = 0, mmw_min = 0, mmw_max = 0;
	[[nodiscard]] auto const_tables() const -> grackleGpuConstTables
	{
		amrex::Abort("grackle_tables::const_tables: Grackle-like cooling is not built (no Grackle data file in the reference tree); "
			     "cooling.cooling_table_type = cloudy_cooling_tools is");
		return {};
	}
};

AMREX_GPU_HOST_DEVICE AMREX_FORCE_INLINE auto cloudy_cooling_function(amrex::Real /*rho*/, amrex::Real /*T*/, grackleGpuConstTables const & /*tables*/) -> amrex::Real
{
	return std::numeric_limits<double>::quiet_NaN();
}
AMREX_GPU_HOST_DEVICE AMREX_FORCE_INLINE auto ComputeEgasFromTgas(double /*rho*/, double /*Tgas*/, double /*gamma*/, grackleGpuConstTables const & /*tables*/) -> amrex::Real
{
	return std::numeric_limits<double>::quiet_NaN();
}
AMREX_GPU_HOST_DEVICE AMREX_FORCE_INLINE auto ComputeTgasFromEgas(double /*rho*/, double /*Egas*/, double /*gamma*/, grackleGpuConstTables const & /*tables*/) -> amrex::Real
{
	return std::numeric_limits<double>::quiet_NaN();
}
template <typename problem_t>
auto computeCooling(amrex::MultiFab & /*mf*/, const amrex::Real /*dt_in*/, grackle_tables & /*tables*/, const amrex::Real /*T_floor*/) -> bool
{
	amrex::Abort("GrackleLikeCooling::computeCooling: not built (see compat/grackle_like_cooling.hpp)");
	return false;
}
inline void readGrackleData(std::string &grackle_hdf5_file, grackle_tables & /*tables*/)
{
	amrex::Abort("cooling.cooling_table_type = grackle (" + grackle_hdf5_file +
		     "): Grackle-like cooling (src/cooling/GrackleLikeCooling.hpp) is not built in quokka_amd/host — its data files are not in the reference "
		     "tree; cloudy_cooling_tools tables are read and integrated (compat/tabulated_cooling.hpp)");
}

} // namespace quokka::GrackleLikeCooling

#endif // QK_HOST_COMPAT_GRACKLE_LIKE_COOLING_HPP_
