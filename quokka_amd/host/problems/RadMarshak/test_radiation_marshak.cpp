// Marshak wave of Su & Olson (1996) in 1-D — problem generator written against the reference's surface (cf. reference
// src/problems/RadMarshak/test_radiation_marshak.cpp; deck tests/Marshak.in).  Radiation only, E_gas = alpha / 4 T^4, kappa = 1,
// Marshak half-range condition on the lower face.  Exit status = the reference's pass criterion: relative L1 error of the radiation
// temperature against the tabulated solution (extern/SuOlson/100pt_tau10p0.dat, path from the deck: `marshak.solution_file`)
// below 2 per cent.
#include <cmath>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "AMReX_BC_TYPES.H"
#include "AMReX_ParmParse.H"
#include "AMReX_Print.H"

#include "QuokkaSimulation.hpp"
#include "hydro/EOS.hpp"
#include "math/interpolate.hpp"
#include "radiation/radiation_system.hpp"

struct SuOlsonProblem {
};

constexpr double eps_SuOlson = 1.0;
constexpr double kappa = 1.0;
constexpr double rho0 = 1.0;
constexpr double T_hohlraum = 1.0;
constexpr double a_rad = 1.0;
constexpr double c = 1.0;
constexpr double alpha_SuOlson = 4.0 * a_rad / eps_SuOlson;
constexpr double T_initial = 1.0e-2;

template <> struct quokka::EOS_Traits<SuOlsonProblem> {
	static constexpr double mean_molecular_weight = 1.0;
	static constexpr double boltzmann_constant = 1.0;
	static constexpr double gamma = 5. / 3.;
	static constexpr double cs_isothermal = std::numeric_limits<double>::quiet_NaN();
};

template <> struct RadSystem_Traits<SuOlsonProblem> {
	static constexpr double c_light = c;
	static constexpr double c_hat = c;
	static constexpr double radiation_constant = a_rad;
	static constexpr double Erad_floor = 0.;
	static constexpr int beta_order = 0;
};

template <> struct Physics_Traits<SuOlsonProblem> {
	static constexpr bool is_hydro_enabled = false;
	static constexpr int numMassScalars = 0;
	static constexpr int numPassiveScalars = numMassScalars + 0;
	static constexpr bool is_radiation_enabled = true;
	static constexpr bool is_mhd_enabled = false;
	static constexpr int nGroups = 1;
};

template <> auto RadSystem<SuOlsonProblem>::ComputePlanckOpacity(const double /*rho*/, const double /*Tgas*/) -> amrex::Real { return kappa; }
template <> auto RadSystem<SuOlsonProblem>::ComputeFluxMeanOpacity(const double /*rho*/, const double /*Tgas*/) -> amrex::Real { return kappa; }

// the material of the exact solution: E_int = alpha / 4 T^4, heat capacity alpha T^3
template <> auto quokka::EOS<SuOlsonProblem>::ComputeTgasFromEint(const double /*rho*/, const double Egas, MassScalars const & /*massScalars*/) -> double
{
	return std::pow(4.0 * Egas / alpha_SuOlson, 1. / 4.);
}
template <> auto quokka::EOS<SuOlsonProblem>::ComputeEintFromTgas(const double /*rho*/, const double Tgas, MassScalars const & /*massScalars*/) -> double
{
	return (alpha_SuOlson / 4.0) * std::pow(Tgas, 4);
}
template <> auto quokka::EOS<SuOlsonProblem>::ComputeEintTempDerivative(const double /*rho*/, const double Tgas, MassScalars const & /*massScalars*/) -> double
{
	return alpha_SuOlson * std::pow(Tgas, 3);
}

template <>
void AMRSimulation<SuOlsonProblem>::setCustomBoundaryConditions(const amrex::IntVect &iv, amrex::Array4<amrex::Real> const &consVar, int /*dcomp*/,
								int /*numcomp*/, amrex::GeometryData const & /*geom*/, const amrex::Real /*time*/,
								const amrex::BCRec *bcr, int /*bcomp*/, int /*orig_comp*/)
{
	if (!((bcr->lo(0) == amrex::BCType::ext_dir) || (bcr->hi(0) == amrex::BCType::ext_dir))) {
		return;
	}
	auto const i = iv.toArray()[0];
	int const j = 0, k = 0;

	if (i < 0) {
		// Marshak condition: the incident half-range flux is that of a hohlraum at T_H; the ghost flux follows from the state
		// of the first cell inside the face
		const double E_inc = a_rad * std::pow(T_hohlraum, 4);
		const double E_0 = consVar(0, j, k, RadSystem<SuOlsonProblem>::radEnergy_index);
		const double F_0 = consVar(0, j, k, RadSystem<SuOlsonProblem>::x1RadFlux_index);
		const double F_bdry = 0.5 * c * E_inc - 0.5 * (c * E_0 + 2.0 * F_0);
		consVar(i, j, k, RadSystem<SuOlsonProblem>::radEnergy_index) = E_inc;
		consVar(i, j, k, RadSystem<SuOlsonProblem>::x1RadFlux_index) = F_bdry;
		consVar(i, j, k, RadSystem<SuOlsonProblem>::x2RadFlux_index) = 0.;
		consVar(i, j, k, RadSystem<SuOlsonProblem>::x3RadFlux_index) = 0.;
	} else {
		consVar(i, j, k, RadSystem<SuOlsonProblem>::radEnergy_index) = a_rad * std::pow(T_initial, 4);
		consVar(i, j, k, RadSystem<SuOlsonProblem>::x1RadFlux_index) = 0;
		consVar(i, j, k, RadSystem<SuOlsonProblem>::x2RadFlux_index) = 0;
		consVar(i, j, k, RadSystem<SuOlsonProblem>::x3RadFlux_index) = 0;
	}

	const double Egas = quokka::EOS<SuOlsonProblem>::ComputeEintFromTgas(rho0, T_initial);
	consVar(i, j, k, RadSystem<SuOlsonProblem>::gasEnergy_index) = Egas;
	consVar(i, j, k, RadSystem<SuOlsonProblem>::gasDensity_index) = rho0;
	consVar(i, j, k, RadSystem<SuOlsonProblem>::gasInternalEnergy_index) = Egas;
	consVar(i, j, k, RadSystem<SuOlsonProblem>::x1GasMomentum_index) = 0.;
	consVar(i, j, k, RadSystem<SuOlsonProblem>::x2GasMomentum_index) = 0.;
	consVar(i, j, k, RadSystem<SuOlsonProblem>::x3GasMomentum_index) = 0.;
}

template <> void QuokkaSimulation<SuOlsonProblem>::setInitialConditionsOnGrid(quokka::grid const &grid_elem)
{
	const amrex::Array4<double> &state_cc = grid_elem.array_;
	amrex::ParallelFor(grid_elem.indexRange_, [=](int i, int j, int k) {
		const double Egas = quokka::EOS<SuOlsonProblem>::ComputeEintFromTgas(rho0, T_initial);
		const double Erad = a_rad * std::pow(T_initial, 4);
		state_cc(i, j, k, RadSystem<SuOlsonProblem>::radEnergy_index) = Erad;
		state_cc(i, j, k, RadSystem<SuOlsonProblem>::x1RadFlux_index) = 0;
		state_cc(i, j, k, RadSystem<SuOlsonProblem>::x2RadFlux_index) = 0;
		state_cc(i, j, k, RadSystem<SuOlsonProblem>::x3RadFlux_index) = 0;
		state_cc(i, j, k, RadSystem<SuOlsonProblem>::gasDensity_index) = rho0;
		state_cc(i, j, k, RadSystem<SuOlsonProblem>::gasEnergy_index) = Egas;
		state_cc(i, j, k, RadSystem<SuOlsonProblem>::gasInternalEnergy_index) = Egas;
		state_cc(i, j, k, RadSystem<SuOlsonProblem>::x1GasMomentum_index) = 0.;
		state_cc(i, j, k, RadSystem<SuOlsonProblem>::x2GasMomentum_index) = 0.;
		state_cc(i, j, k, RadSystem<SuOlsonProblem>::x3GasMomentum_index) = 0.;
	});
}

auto problem_main() -> int
{
	const int max_timesteps = 2e4;
	const double CFL_number = 0.4;
	const double initial_dtau = 1e-9; // dimensionless times
	const double max_dtau = 1e-3;
	const double max_tau = 10.0;

	const double chi = rho0 * kappa;
	const double max_time = max_tau / (eps_SuOlson * c * chi);
	const double max_dt = max_dtau / (eps_SuOlson * c * chi);
	const double initial_dt = initial_dtau / (eps_SuOlson * c * chi);

	constexpr int nvars = RadSystem<SuOlsonProblem>::nvar_;
	amrex::Vector<amrex::BCRec> BCs_cc(nvars);
	for (int n = 0; n < nvars; ++n) {
		BCs_cc[n].setLo(0, amrex::BCType::ext_dir);  // custom (Marshak) x1
		BCs_cc[n].setHi(0, amrex::BCType::foextrap); // extrapolate x1
	}

	QuokkaSimulation<SuOlsonProblem> sim(BCs_cc);
	sim.stopTime_ = max_time;
	sim.radiationCflNumber_ = CFL_number;
	sim.initDt_ = initial_dt;
	sim.maxDt_ = max_dt;
	sim.maxTimesteps_ = max_timesteps;
	sim.plotfileInterval_ = -1;

	sim.setInitialConditions();
	sim.evolve();

	// radiation temperature on the scaled coordinate sqrt(3) x
	auto const &mf = sim.state_new_cc_[0];
	int const nx = sim.geom[0].Domain().length(0);
	std::vector<double> xs(nx), Trad(nx);
	for (int b = 0; b < mf.size(); ++b) {
		auto h = mf.copyToHost(b);
		amrex::Array4<double> a(h.data(), mf.fabbox(b), mf.nComp());
		amrex::ParallelFor(mf.validbox(b), [&](int i, int j, int k) {
			xs.at(i) = std::sqrt(3.0) * (sim.geom[0].ProbLo(0) + (i + 0.5) * sim.geom[0].CellSize(0));
			Trad.at(i) = std::pow(a(i, j, k, RadSystem<SuOlsonProblem>::radEnergy_index) / a_rad, 1. / 4.);
		});
	}

	std::string filename = "../extern/SuOlson/100pt_tau10p0.dat";
	amrex::ParmParse pp("marshak");
	pp.query("solution_file", filename);
	std::ifstream fstream(filename, std::ios::in);
	AMREX_ALWAYS_ASSERT(fstream.is_open());
	std::vector<double> xs_exact, Trad_exact;
	std::string header;
	std::getline(fstream, header);
	for (std::string line; std::getline(fstream, line);) {
		std::istringstream iss(line);
		std::vector<double> row;
		for (double value = NAN; iss >> value;) {
			row.push_back(value);
		}
		if (row.size() < 6) {
			continue;
		}
		xs_exact.push_back(std::sqrt(3.0) * row.at(1));
		Trad_exact.push_back(row.at(4));
	}

	std::vector<double> Trad_exact_interp(xs.size());
	interpolate_arrays(xs.data(), Trad_exact_interp.data(), static_cast<int>(xs.size()), xs_exact.data(), Trad_exact.data(),
			   static_cast<int>(xs_exact.size()));

	double err_norm = 0., sol_norm = 0.;
	const double xmax = c * sim.tNew_[0];
	for (size_t i = 0; i < xs.size(); ++i) {
		if (xs[i] < xmax) {
			err_norm += std::abs(Trad[i] - Trad_exact_interp[i]);
			sol_norm += std::abs(Trad_exact_interp[i]);
		}
	}
	const double error_tol = 0.02; // 2 per cent
	const double rel_error = err_norm / sol_norm;
	sim.errorNorm_ = rel_error;
	amrex::Print() << "Relative L1 error norm = " << rel_error << std::endl;
	qkDumpState(sim);
	return ((rel_error > error_tol) || std::isnan(rel_error)) ? 1 : 0;
}
