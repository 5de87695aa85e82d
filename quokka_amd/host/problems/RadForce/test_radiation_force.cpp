// Radiation-driven isothermal wind in 1-D — problem generator written against the reference's surface (cf. reference
// src/problems/RadForce/test_radiation_force.cpp; deck tests/RadForce.in).  Isothermal gas (gamma = 1), optically thin flux with a
// flux-mean opacity only, beta_order = 1, c_hat = 10 Mach1 a0.  Exit status = the reference's pass criterion: relative L1 error of the
// Mach number against the steady wind solution (extern/pressure_tube/optically_thin_wind.txt, path from the deck:
// `radforce.solution_file`) below 0.002.
#include <cmath>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "AMReX_BC_TYPES.H"
#include "AMReX_ParmParse.H"
#include "AMReX_Print.H"

#include "QuokkaSimulation.hpp"
#include "math/interpolate.hpp"
#include "radiation/radiation_system.hpp"

struct TubeProblem {
};

constexpr double kappa0 = 5.0;	     // cm^2 g^-1
constexpr double mu = 2.33 * C::m_u; // g
constexpr double gamma_gas = 1.0;    // isothermal gas EOS
constexpr double a0 = 0.2e5;	     // cm s^-1
constexpr double tau = 1.0e-6;	     // optical depth

constexpr double rho0 = 1.0e5 * mu; // g cm^-3
constexpr double Mach0 = 1.1;	    // Mach number at the wind base
constexpr double Mach1 = 2.128410288469465339;

constexpr double Frad0 = rho0 * a0 * c_light_cgs_ / tau; // erg cm^-2 s^-1
constexpr double g0 = kappa0 * Frad0 / c_light_cgs_;	 // cm s^-2
constexpr double Lx = (a0 * a0) / g0;			 // cm

template <> struct quokka::EOS_Traits<TubeProblem> {
	static constexpr double mean_molecular_weight = mu;
	static constexpr double boltzmann_constant = C::k_B;
	static constexpr double gamma = gamma_gas;
	static constexpr double cs_isothermal = a0; // only used when gamma = 1
};

template <> struct Physics_Traits<TubeProblem> {
	static constexpr bool is_hydro_enabled = true;
	static constexpr int numMassScalars = 0;
	static constexpr int numPassiveScalars = numMassScalars + 0;
	static constexpr bool is_radiation_enabled = true;
	static constexpr bool is_mhd_enabled = false;
	static constexpr int nGroups = 1;
};

template <> struct RadSystem_Traits<TubeProblem> {
	static constexpr double c_light = c_light_cgs_;
	static constexpr double c_hat = 10. * (Mach1 * a0);
	static constexpr double radiation_constant = radiation_constant_cgs_;
	static constexpr double Erad_floor = 0.;
	static constexpr int beta_order = 1;
};

template <> auto RadSystem<TubeProblem>::ComputePlanckOpacity(const double /*rho*/, const double /*Tgas*/) -> amrex::Real { return 0.; }
template <> auto RadSystem<TubeProblem>::ComputeFluxMeanOpacity(const double /*rho*/, const double /*Tgas*/) -> amrex::Real { return kappa0; }

template <> void QuokkaSimulation<TubeProblem>::setInitialConditionsOnGrid(quokka::grid const &grid_elem)
{
	const amrex::Array4<double> &state_cc = grid_elem.array_;
	amrex::ParallelFor(grid_elem.indexRange_, [=](int i, int j, int k) {
		state_cc(i, j, k, RadSystem<TubeProblem>::radEnergy_index) = Frad0 * 1.0 / c_light_cgs_;
		state_cc(i, j, k, RadSystem<TubeProblem>::x1RadFlux_index) = Frad0 * 1.0;
		state_cc(i, j, k, RadSystem<TubeProblem>::x2RadFlux_index) = 0;
		state_cc(i, j, k, RadSystem<TubeProblem>::x3RadFlux_index) = 0;
		state_cc(i, j, k, RadSystem<TubeProblem>::gasDensity_index) = rho0;
		state_cc(i, j, k, RadSystem<TubeProblem>::x1GasMomentum_index) = 0;
		state_cc(i, j, k, RadSystem<TubeProblem>::x2GasMomentum_index) = 0;
		state_cc(i, j, k, RadSystem<TubeProblem>::x3GasMomentum_index) = 0;
		state_cc(i, j, k, RadSystem<TubeProblem>::gasEnergy_index) = 0;
		state_cc(i, j, k, RadSystem<TubeProblem>::gasInternalEnergy_index) = 0.;
	});
}

template <>
void AMRSimulation<TubeProblem>::setCustomBoundaryConditions(const amrex::IntVect &iv, amrex::Array4<amrex::Real> const &consVar, int /*dcomp*/,
							     int /*numcomp*/, amrex::GeometryData const &geom, const amrex::Real /*time*/,
							     const amrex::BCRec * /*bcr*/, int /*bcomp*/, int /*orig_comp*/)
{
	auto const i = iv.toArray()[0];
	int const j = 0, k = 0;
	amrex::Box const &box = geom.Domain();
	if (i < box.loVect3d()[0]) { // wind base: inflow at Mach0 carrying the incident flux
		amrex::Real const rho = rho0;
		amrex::Real const vel = Mach0 * a0;
		consVar(i, j, k, RadSystem<TubeProblem>::radEnergy_index) = Frad0 / c_light_cgs_;
		consVar(i, j, k, RadSystem<TubeProblem>::x1RadFlux_index) = Frad0;
		consVar(i, j, k, RadSystem<TubeProblem>::x2RadFlux_index) = 0.;
		consVar(i, j, k, RadSystem<TubeProblem>::x3RadFlux_index) = 0.;
		consVar(i, j, k, RadSystem<TubeProblem>::gasDensity_index) = rho;
		consVar(i, j, k, RadSystem<TubeProblem>::gasEnergy_index) = 0.;
		consVar(i, j, k, RadSystem<TubeProblem>::gasInternalEnergy_index) = 0.;
		consVar(i, j, k, RadSystem<TubeProblem>::x1GasMomentum_index) = rho * vel;
		consVar(i, j, k, RadSystem<TubeProblem>::x2GasMomentum_index) = 0.;
		consVar(i, j, k, RadSystem<TubeProblem>::x3GasMomentum_index) = 0.;
	}
}

auto problem_main() -> int
{
	constexpr double CFL_number = 0.4;
	double max_dt = 1.0e10;
	constexpr double tmax = 10.0 * (Lx / a0);
	constexpr int max_timesteps = 1e6;

	constexpr int nvars = RadSystem<TubeProblem>::nvar_;
	amrex::Vector<amrex::BCRec> BCs_cc(nvars);
	for (int n = 0; n < nvars; ++n) {
		BCs_cc[n].setLo(0, amrex::BCType::ext_dir);
		BCs_cc[n].setHi(0, amrex::BCType::foextrap);
	}

	amrex::ParmParse const pp;
	pp.query("max_dt", max_dt);

	QuokkaSimulation<TubeProblem> sim(BCs_cc);
	sim.radiationReconstructionOrder_ = 3; // PPM
	sim.reconstructionOrder_ = 3;	       // PPM
	sim.stopTime_ = tmax;
	sim.cflNumber_ = CFL_number;
	sim.radiationCflNumber_ = CFL_number;
	sim.maxTimesteps_ = max_timesteps;
	sim.plotfileInterval_ = -1;
	sim.maxDt_ = max_dt;

	sim.setInitialConditions();
	sim.evolve();

	auto const &mf = sim.state_new_cc_[0];
	int const nx = sim.geom[0].Domain().length(0);
	std::vector<double> xs_norm(nx), Mach_arr(nx);
	for (int b = 0; b < mf.size(); ++b) {
		auto h = mf.copyToHost(b);
		amrex::Array4<double> a(h.data(), mf.fabbox(b), mf.nComp());
		amrex::ParallelFor(mf.validbox(b), [&](int i, int j, int k) {
			double const x = sim.geom[0].ProbLo(0) + (i + 0.5) * sim.geom[0].CellSize(0);
			double const rho = a(i, j, k, RadSystem<TubeProblem>::gasDensity_index);
			double const vx = a(i, j, k, RadSystem<TubeProblem>::x1GasMomentum_index) / rho;
			xs_norm.at(i) = x / Lx;
			Mach_arr.at(i) = vx / a0;
		});
	}

	std::string filename = "../extern/pressure_tube/optically_thin_wind.txt";
	amrex::ParmParse ppr("radforce");
	ppr.query("solution_file", filename);
	std::ifstream fstream(filename, std::ios::in);
	AMREX_ALWAYS_ASSERT(fstream.is_open());
	std::string header;
	std::getline(fstream, header);
	std::vector<double> x_exact, Mach_exact;
	for (std::string line; std::getline(fstream, line);) {
		std::istringstream iss(line);
		std::vector<double> row;
		for (double value = NAN; iss >> value;) {
			row.push_back(value);
		}
		if (row.size() < 3) {
			continue;
		}
		x_exact.push_back(row.at(0));
		Mach_exact.push_back(row.at(2));
	}

	std::vector<double> Mach_interp(nx);
	interpolate_arrays(xs_norm.data(), Mach_interp.data(), nx, x_exact.data(), Mach_exact.data(), static_cast<int>(x_exact.size()));

	double err_norm = 0., sol_norm = 0.;
	for (int i = 0; i < nx; ++i) {
		err_norm += std::abs(Mach_arr[i] - Mach_interp[i]);
		sol_norm += std::abs(Mach_interp[i]);
	}
	const double rel_err_norm = err_norm / sol_norm;
	const double rel_err_tol = 0.002;
	sim.errorNorm_ = rel_err_norm;
	amrex::Print() << "Relative L1 norm = " << rel_err_norm << std::endl;
	qkDumpState(sim);
	return (rel_err_norm < rel_err_tol) ? 0 : 1;
}
