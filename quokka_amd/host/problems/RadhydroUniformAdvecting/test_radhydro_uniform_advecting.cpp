// Gas and radiation in thermal equilibrium advecting at 0.01 c through a periodic 1-D box — problem generator written against the
// reference's surface (cf. reference src/problems/RadhydroUniformAdvecting/test_radhydro_uniform_advecting.cpp; deck
// tests/RadhydroUniformAdvecting.in).  beta_order = 2, constant opacity, radiation CFL 8.  Exit status = the reference's pass
// criterion: relative L1 deviation of T_gas / T0 from 1 below 1e-10.
#include <cmath>
#include <vector>

#include "AMReX_BC_TYPES.H"
#include "AMReX_Print.H"

#include "QuokkaSimulation.hpp"
#include "hydro/EOS.hpp"
#include "radiation/radiation_system.hpp"

struct PulseProblem {
};

constexpr int beta_order_ = 2;
constexpr double c = 1.0e8;
constexpr double chat = c;
constexpr double v0 = 1e-2 * c;
constexpr double kappa0 = 1.0e5;

constexpr double T0 = 1.0;
constexpr double rho0 = 1.0;
constexpr double a_rad = 1.0;
constexpr double mu = 1.0;
constexpr double k_B = 1.0;

constexpr double max_time = 10.0 / v0;
constexpr double Erad0 = a_rad * T0 * T0 * T0 * T0;
constexpr double Erad_beta2 = (1. + 4. / 3. * (v0 * v0) / (c * c)) * Erad0;

template <> struct quokka::EOS_Traits<PulseProblem> {
	static constexpr double mean_molecular_weight = mu;
	static constexpr double boltzmann_constant = k_B;
	static constexpr double gamma = 5. / 3.;
	static constexpr double cs_isothermal = std::numeric_limits<double>::quiet_NaN();
};

template <> struct Physics_Traits<PulseProblem> {
	static constexpr bool is_hydro_enabled = true;
	static constexpr int numMassScalars = 0;
	static constexpr int numPassiveScalars = numMassScalars + 0;
	static constexpr bool is_radiation_enabled = true;
	static constexpr bool is_mhd_enabled = false;
	static constexpr int nGroups = 1;
};

template <> struct RadSystem_Traits<PulseProblem> {
	static constexpr double c_light = c;
	static constexpr double c_hat = chat;
	static constexpr double radiation_constant = a_rad;
	static constexpr double Erad_floor = 0.0;
	static constexpr int beta_order = beta_order_;
};

template <> auto RadSystem<PulseProblem>::ComputePlanckOpacity(const double /*rho*/, const double /*Tgas*/) -> amrex::Real { return kappa0; }
template <> auto RadSystem<PulseProblem>::ComputeFluxMeanOpacity(const double /*rho*/, const double /*Tgas*/) -> amrex::Real { return kappa0; }

template <> void QuokkaSimulation<PulseProblem>::setInitialConditionsOnGrid(quokka::grid const &grid_elem)
{
	const amrex::Array4<double> &state_cc = grid_elem.array_;
	const auto Egas = quokka::EOS<PulseProblem>::ComputeEintFromTgas(rho0, T0);

	// lab-frame moments of an isotropic comoving field, to the order in v/c the solver keeps
	double erad = NAN;
	double frad = NAN;
	if constexpr (beta_order_ == 0) {
		erad = Erad0;
		frad = 0.0;
	} else if constexpr (beta_order_ == 1) {
		erad = Erad0;
		frad = 4. / 3. * v0 * Erad0;
	} else if constexpr (beta_order_ == 2) {
		erad = Erad_beta2;
		frad = 4. / 3. * v0 * Erad0;
	} else {
		erad = Erad_beta2;
		frad = 4. / 3. * v0 * Erad0 * (1. + (v0 * v0) / (c * c));
	}

	amrex::ParallelFor(grid_elem.indexRange_, [=](int i, int j, int k) {
		state_cc(i, j, k, RadSystem<PulseProblem>::radEnergy_index) = erad;
		state_cc(i, j, k, RadSystem<PulseProblem>::x1RadFlux_index) = frad;
		state_cc(i, j, k, RadSystem<PulseProblem>::x2RadFlux_index) = 0;
		state_cc(i, j, k, RadSystem<PulseProblem>::x3RadFlux_index) = 0;
		state_cc(i, j, k, RadSystem<PulseProblem>::gasEnergy_index) = Egas + 0.5 * rho0 * v0 * v0;
		state_cc(i, j, k, RadSystem<PulseProblem>::gasDensity_index) = rho0;
		state_cc(i, j, k, RadSystem<PulseProblem>::gasInternalEnergy_index) = Egas;
		state_cc(i, j, k, RadSystem<PulseProblem>::x1GasMomentum_index) = v0 * rho0;
		state_cc(i, j, k, RadSystem<PulseProblem>::x2GasMomentum_index) = 0.;
		state_cc(i, j, k, RadSystem<PulseProblem>::x3GasMomentum_index) = 0.;
	});
}

auto problem_main() -> int
{
	const int max_timesteps = 1e6;
	const double CFL_number_gas = 0.8;
	const double CFL_number_rad = 8.0;
	const double max_dt = 1.0;

	constexpr int nvars = RadSystem<PulseProblem>::nvar_;
	amrex::Vector<amrex::BCRec> BCs_cc(nvars);
	for (int n = 0; n < nvars; ++n) {
		for (int i = 0; i < AMREX_SPACEDIM; ++i) {
			BCs_cc[n].setLo(i, amrex::BCType::int_dir); // periodic
			BCs_cc[n].setHi(i, amrex::BCType::int_dir);
		}
	}

	QuokkaSimulation<PulseProblem> sim(BCs_cc);
	sim.radiationReconstructionOrder_ = 3; // PPM
	sim.stopTime_ = max_time;
	sim.radiationCflNumber_ = CFL_number_rad;
	sim.cflNumber_ = CFL_number_gas;
	sim.maxDt_ = max_dt;
	sim.maxTimesteps_ = max_timesteps;
	sim.plotfileInterval_ = -1;

	sim.setInitialConditions();
	sim.evolve();

	auto const &mf = sim.state_new_cc_[0];
	double err_norm = 0., sol_norm = 0., verr = 0.;
	for (int b = 0; b < mf.size(); ++b) {
		auto h = mf.copyToHost(b);
		amrex::Array4<double> a(h.data(), mf.fabbox(b), mf.nComp());
		amrex::ParallelFor(mf.validbox(b), [&](int i, int j, int k) {
			double const rho_t = a(i, j, k, RadSystem<PulseProblem>::gasDensity_index);
			double const Eint = a(i, j, k, RadSystem<PulseProblem>::gasInternalEnergy_index);
			double const Tgas = quokka::EOS<PulseProblem>::ComputeTgasFromEint(rho_t, Eint) / T0;
			err_norm += std::abs(Tgas - 1.0);
			sol_norm += 1.0;
			verr = std::max(verr, std::abs(a(i, j, k, RadSystem<PulseProblem>::x1GasMomentum_index) / rho_t / v0 - 1.0));
		});
	}
	const double error_tol = 1.0e-10; // "to machine accuracy"
	const double rel_error = err_norm / sol_norm;
	sim.errorNorm_ = rel_error;
	amrex::Print() << "Relative L1 error norm = " << rel_error << " (max |v / v0 - 1| = " << verr << ")" << std::endl;
	qkDumpState(sim);
	return (rel_error < error_tol && !std::isnan(rel_error)) ? 0 : 1;
}
