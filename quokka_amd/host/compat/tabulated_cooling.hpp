// tabulated_cooling.hpp — quokka::TabulatedCooling of the reference (src/cooling/TabulatedCooling.hpp, TabulatedCooling.cpp, CloudyDataReader.cpp) over
// the library: the table file is read by qk_cloudy_tables_read (the library's own reader of the HDF5 format), the Strang-split source is
// qk_cooling_tabulated, and the per-cell functions a problem calls from its own kernels AND from host code (ComputeTgasFromEgas, ComputeMMW,
// ComputeCoolingLength, ComputeEgasFromTgas, cloudy_cooling_function) are the host + device functions of csrc/qk_cooling_device.hpp the kernel
// itself is made of.  Same names, arguments and results as the reference's.
#ifndef QK_HOST_COMPAT_TABULATED_COOLING_HPP_
#define QK_HOST_COMPAT_TABULATED_COOLING_HPP_

#include <string>

#include "../../csrc/qk_cooling_device.hpp"
#include "../quokka_rad_system.hpp"
#include "mini_fmt.hpp" // (the reference's header includes fmt/core.h: problems that include it use fmt::format without naming the header)

namespace quokka::TabulatedCooling
{

constexpr double cloudy_H_mass_fraction = qk::cool::H_mass_fraction; // TabulatedCooling.hpp:32

// what a kernel captures by value (TabulatedCooling.hpp:34-52).  The problem files only pass it on, so it holds the library's view of the tables
// twice: device pointers for device code, host pointers for the calls problems make from problem_main and computeAfterEvolve.
struct cloudyGpuConstTables {
	qk::cool::Tables dev{}, host{};
	amrex::Real T_min = 0, T_max = 0, mmw_min = 0, mmw_max = 0;
	AMREX_GPU_HOST_DEVICE [[nodiscard]] auto view() const -> qk::cool::Tables const &
	{
#if defined(__HIP_DEVICE_COMPILE__)
		return dev;
#else
		return host;
#endif
	}
};

// TabulatedCooling.hpp:54-73; filled by readCloudyData
class cloudy_tables
{
      public:
	amrex::Real T_min = 0, T_max = 0, mmw_min = 0, mmw_max = 0;

	cloudy_tables() = default;
	cloudy_tables(cloudy_tables const &) = delete;
	auto operator=(cloudy_tables const &) -> cloudy_tables & = delete;
	~cloudy_tables() { release(); }

	[[nodiscard]] auto const_tables() const -> cloudyGpuConstTables
	{
		if (host_.log_nH == nullptr) {
			amrex::Abort("cloudy_tables::const_tables: no table was read (cooling.enabled or cooling.read_tables_even_if_disabled, cooling.hdf5_data_file)");
		}
		cloudyGpuConstTables t;
		t.host = view(host_);
		t.dev = view(dev_);
		t.T_min = T_min;
		t.T_max = T_max;
		t.mmw_min = mmw_min;
		t.mmw_max = mmw_max;
		return t;
	}
	[[nodiscard]] auto device() const -> qk_cloudy_tables const * { return &dev_; }
	[[nodiscard]] auto empty() const -> bool { return host_.log_nH == nullptr; }

	void read(std::string const &file)
	{
		release();
		qkhost::check(qk_cloudy_tables_read(qkhost::Runtime::get().ctx, file.c_str(), &host_), "readCloudyData");
		dev_ = host_;
		size_t const n0 = static_cast<size_t>(host_.n_nH), n1 = static_cast<size_t>(host_.n_Tgas);
		dev_.log_nH = upload(host_.log_nH, n0);
		dev_.log_Tgas = upload(host_.log_Tgas, n1);
		dev_.cooling = upload(host_.cooling, n0 * n1);
		dev_.heating = upload(host_.heating, n0 * n1);
		dev_.mean_mol_weight = upload(host_.mean_mol_weight, n0 * n1);
		T_min = host_.T_min;
		T_max = host_.T_max;
		mmw_min = host_.mmw_min;
		mmw_max = host_.mmw_max;
	}

      private:
	qk_cloudy_tables host_{}, dev_{};

	static auto upload(const double *h, size_t n) -> const double *
	{
		double *d = nullptr;
		QK_HOST_HIP(hipMalloc(reinterpret_cast<void **>(&d), sizeof(double) * n));
		QK_HOST_HIP(hipMemcpy(d, h, sizeof(double) * n, hipMemcpyHostToDevice));
		return d;
	}
	static auto view(qk_cloudy_tables const &c) -> qk::cool::Tables
	{
		qk::cool::Tables t{};
		t.log_nH = c.log_nH;
		t.log_T = c.log_Tgas;
		t.cool = c.cooling;
		t.heat = c.heating;
		t.mmw = c.mean_mol_weight;
		t.n_nH = c.n_nH;
		t.n_T = c.n_Tgas;
		t.T_min = c.T_min;
		t.T_max = c.T_max;
		t.mmw_min = c.mmw_min;
		t.mmw_max = c.mmw_max;
		t.m_H = C::m_p + C::m_e;
		t.k_B = C::k_B;
		t.prepared = 0; // (the per-cell functions prepare a copy where they run: host pointers on the host, device pointers in a kernel)
		return t;
	}
	void release()
	{
		if (host_.log_nH != nullptr) {
			for (const double *p : {dev_.log_nH, dev_.log_Tgas, dev_.cooling, dev_.heating, dev_.mean_mol_weight}) {
				(void)hipFree(const_cast<double *>(p));
			}
			qk_cloudy_tables_free(&host_);
			dev_ = qk_cloudy_tables{};
		}
	}
};

// TabulatedCooling.hpp:82-220
AMREX_GPU_HOST_DEVICE AMREX_FORCE_INLINE auto cloudy_cooling_function(amrex::Real const rho, amrex::Real const T, cloudyGpuConstTables const &tables) -> amrex::Real
{
	return qk::cool::netHeating(tables.view(), rho, T);
}
AMREX_GPU_HOST_DEVICE AMREX_FORCE_INLINE auto ComputeEgasFromTgas(double rho, double Tgas, double gamma, cloudyGpuConstTables const &tables) -> amrex::Real
{
	return qk::cool::egasFromTgas(tables.view(), rho, Tgas, gamma);
}
AMREX_GPU_HOST_DEVICE AMREX_FORCE_INLINE auto ComputeTgasFromEgas(double rho, double Egas, double gamma, cloudyGpuConstTables const &tables) -> amrex::Real
{
	return qk::cool::tgasFromEgas(tables.view(), rho, Egas, gamma);
}
AMREX_GPU_HOST_DEVICE AMREX_FORCE_INLINE auto ComputeCoolingLength(double rho, double Egas, double gamma, cloudyGpuConstTables const &tables) -> amrex::Real
{
	return qk::cool::coolingLength(tables.view(), rho, Egas, gamma);
}
AMREX_GPU_HOST_DEVICE AMREX_FORCE_INLINE auto ComputeMMW(double rho, double Egas, double gamma, cloudyGpuConstTables const &tables) -> amrex::Real
{
	return qk::cool::meanMolecularWeight(tables.view(), rho, Egas, gamma);
}

// computeCooling<problem_t> (TabulatedCooling.hpp:258-317): false = the integration failed in some cell, the caller retries the hydro step
template <typename problem_t> auto computeCooling(amrex::MultiFab &mf, const amrex::Real dt_in, cloudy_tables &cloudyTables, const amrex::Real T_floor) -> bool
{
	static long long *d_counters = nullptr;
	if (d_counters == nullptr) {
		QK_HOST_HIP(hipMalloc(reinterpret_cast<void **>(&d_counters), 2 * sizeof(long long)));
	}
	auto const t = qkhost::traits<problem_t>();
	qk_ctx *ctx = qkhost::Runtime::get().ctx;
	qkhost::check(qk_clear_bytes(ctx, nullptr, d_counters, 2 * sizeof(long long)), "computeCooling");
	qkhost::check(qk_cooling_tabulated(qkhost::Runtime::get().lev, nullptr, &t, qkhost::tab(mf), cloudyTables.device(), dt_in, T_floor, d_counters), "computeCooling");
	long long h[2] = {0, 0};
	QK_HOST_HIP(hipMemcpy(h, d_counters, sizeof(h), hipMemcpyDeviceToHost));
	int const nmax = qkhost::Comm::get().allReduceMax(static_cast<int>(h[0]));
	double const nsum = qkhost::Comm::get().allReduceSum(static_cast<double>(h[1]));
	double local_cells = 0;
	for (amrex::MFIter it(mf); it.isValid(); ++it) {
		local_cells += static_cast<double>(it.validbox().numPts());
	}
	double const ncells = qkhost::Comm::get().allReduceSum(local_cells);
	amrex::Print() << "\tcooling substeps (per cell): avg " << nsum / ncells << ", max " << nmax << "\n";
	if (nmax >= qk::cool::maxSubsteps) {
		amrex::Print() << "\t[CloudyCooling] Reaction ODE failure! Retrying hydro update...\n";
		return false;
	}
	return true;
}

inline void readCloudyData(std::string &hdf5_file, cloudy_tables &cloudyTables) { cloudyTables.read(hdf5_file); }

} // namespace quokka::TabulatedCooling

#endif // QK_HOST_COMPAT_TABULATED_COOLING_HPP_
