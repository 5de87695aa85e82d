// Steady subcritical radiative shock in cgs units (Skinner et al. 2019, Sec. 9.5; Lowrie & Edwards 2008) — problem generator written
// against the reference's surface (cf. reference src/problems/RadhydroShockCGS/test_radhydro_shock_cgs.cpp; deck tests/radshock.in).
// Compiled against quokka_amd/host (AMReX is absent), 1-D build.  The device hooks (opacity kappa = k0 / rho, the Eddington
// approximation, the constant states beyond both faces) are sampled on the host into the C-ABI's closed sets, see quokka_host.hpp.
// Exit status = the reference's pass criterion: relative L1 error of T_rad vs extern/LowrieEdwards/shock.txt <= 0.005.
#include <cmath>
#include <fstream>
#include <sstream>
#include <vector>

#include "AMReX_BC_TYPES.H"
#include "AMReX_ParmParse.H"
#include "AMReX_Print.H"

#include "QuokkaSimulation.hpp"
#include "math/interpolate.hpp"
#include "radiation/radiation_system.hpp"

struct ShockProblem {
};

constexpr double a_rad = 7.5646e-15; // erg cm^-3 K^-4
constexpr double c = 2.99792458e10;  // cm s^-1
constexpr double k_B = C::k_B;	     // erg K^-1
constexpr double c_s0 = 1.73e7;	     // adiabatic sound speed [cm s^-1]
constexpr double kappa = 577.0;	     // absorption coefficient rho * kappa [cm^-1]
constexpr double gamma_gas = (5. / 3.);
constexpr double c_v = k_B / ((C::m_p + C::m_e) * (gamma_gas - 1.0)); // specific heat [erg g^-1 K^-1]

// upstream (0) and downstream (1) states
constexpr double T0 = 2.18e6, rho0 = 5.69, v0 = 5.19e7;
constexpr double T1 = 7.98e6, rho1 = 17.1, v1 = 1.73e7;
constexpr double chat = 10.0 * (v0 + c_s0); // reduced speed of light
constexpr double Erad0 = a_rad * (T0 * T0 * T0 * T0), Egas0 = rho0 * c_v * T0;
constexpr double Erad1 = a_rad * (T1 * T1 * T1 * T1), Egas1 = rho1 * c_v * T1;
constexpr double shock_position = 0.01305; // cm (the shock drifts slightly to the right during the run)
constexpr double Lx = 0.01575;		   // cm

template <> struct RadSystem_Traits<ShockProblem> {
	static constexpr double c_light = c;
	static constexpr double c_hat = chat;
	static constexpr double radiation_constant = a_rad;
	static constexpr double Erad_floor = 0.;
	static constexpr int beta_order = 1;
};

template <> struct quokka::EOS_Traits<ShockProblem> {
	static constexpr double mean_molecular_weight = C::m_p + C::m_e;
	static constexpr double boltzmann_constant = k_B;
	static constexpr double gamma = gamma_gas;
	static constexpr double cs_isothermal = std::numeric_limits<double>::quiet_NaN();
};

template <> struct Physics_Traits<ShockProblem> {
	static constexpr bool is_hydro_enabled = true;
	static constexpr int numMassScalars = 0;
	static constexpr int numPassiveScalars = numMassScalars + 0;
	static constexpr bool is_radiation_enabled = true;
	static constexpr bool is_mhd_enabled = false;
	static constexpr int nGroups = 1;
};

template <> auto RadSystem<ShockProblem>::ComputePlanckOpacity(const double rho, const double /*Tgas*/) -> amrex::Real { return kappa / rho; }
template <> auto RadSystem<ShockProblem>::ComputeFluxMeanOpacity(const double rho, const double /*Tgas*/) -> amrex::Real { return ComputePlanckOpacity(rho, 0.0); }
template <> auto RadSystem<ShockProblem>::ComputeEddingtonFactor(double /*f*/) -> double { return (1. / 3.); } // Eddington approximation

namespace
{
// conserved state of a uniform flow (rho, v, E_gas, E_rad) with no radiation flux
void uniformState(amrex::Array4<amrex::Real> const &U, int i, int j, int k, double rho, double v, double Egas, double Erad)
{
	const double px = rho * v;
	U(i, j, k, RadSystem<ShockProblem>::gasDensity_index) = rho;
	U(i, j, k, RadSystem<ShockProblem>::x1GasMomentum_index) = px;
	U(i, j, k, RadSystem<ShockProblem>::x2GasMomentum_index) = 0.;
	U(i, j, k, RadSystem<ShockProblem>::x3GasMomentum_index) = 0.;
	U(i, j, k, RadSystem<ShockProblem>::gasEnergy_index) = Egas + (px * px) / (2 * rho);
	U(i, j, k, RadSystem<ShockProblem>::gasInternalEnergy_index) = Egas;
	U(i, j, k, RadSystem<ShockProblem>::radEnergy_index) = Erad;
	U(i, j, k, RadSystem<ShockProblem>::x1RadFlux_index) = 0;
	U(i, j, k, RadSystem<ShockProblem>::x2RadFlux_index) = 0;
	U(i, j, k, RadSystem<ShockProblem>::x3RadFlux_index) = 0;
}
} // namespace

template <>
void AMRSimulation<ShockProblem>::setCustomBoundaryConditions(const amrex::IntVect &iv, amrex::Array4<amrex::Real> const &consVar, int /*dcomp*/, int /*numcomp*/,
							      amrex::GeometryData const &geom, const amrex::Real /*time*/, const amrex::BCRec *bcr, int /*bcomp*/,
							      int /*orig_comp*/)
{
	if (!((bcr->lo(0) == amrex::BCType::ext_dir) || (bcr->hi(0) == amrex::BCType::ext_dir))) {
		return;
	}
	auto const i = iv.toArray()[0];
	int const j = 0, k = 0;
	amrex::Box const &box = geom.Domain();
	if (i < box.loVect3d()[0]) { // inflow: the upstream state
		uniformState(consVar, i, j, k, rho0, v0, Egas0, Erad0);
	} else if (i >= box.hiVect3d()[0]) { // outflow: the downstream state
		uniformState(consVar, i, j, k, rho1, v1, Egas1, Erad1);
	}
}

template <> void QuokkaSimulation<ShockProblem>::setInitialConditionsOnGrid(quokka::grid const &grid_elem)
{
	amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const dx = grid_elem.dx_;
	amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const prob_lo = grid_elem.prob_lo_;
	const amrex::Array4<double> &state_cc = grid_elem.array_;
	amrex::ParallelFor(grid_elem.indexRange_, [=](int i, int j, int k) {
		amrex::Real const x = prob_lo[0] + (i + amrex::Real(0.5)) * dx[0];
		bool const upstream = x < shock_position;
		const double density = upstream ? rho0 : rho1;
		const double x1Momentum = upstream ? rho0 * v0 : rho1 * v1;
		const double energy = upstream ? Egas0 + 0.5 * rho0 * (v0 * v0) : Egas1 + 0.5 * rho1 * (v1 * v1);
		uniformState(state_cc, i, j, k, density, 0.0, 0.0, upstream ? Erad0 : Erad1);
		state_cc(i, j, k, RadSystem<ShockProblem>::x1GasMomentum_index) = x1Momentum;
		state_cc(i, j, k, RadSystem<ShockProblem>::gasEnergy_index) = energy;
		state_cc(i, j, k, RadSystem<ShockProblem>::gasInternalEnergy_index) = energy - (x1Momentum * x1Momentum) / (2 * density);
	});
}

auto problem_main() -> int
{
	const int max_timesteps = 2e4;
	const double CFL_number = 0.4;
	const double max_time = 1.0e-9; // s

	constexpr int nvars = RadSystem<ShockProblem>::nvar_;
	amrex::Vector<amrex::BCRec> BCs_cc(nvars);
	for (int n = 0; n < nvars; ++n) {
		BCs_cc[n].setLo(0, amrex::BCType::ext_dir); // custom x1
		BCs_cc[n].setHi(0, amrex::BCType::ext_dir);
	}

	QuokkaSimulation<ShockProblem> sim(BCs_cc);
	sim.cflNumber_ = CFL_number;
	sim.radiationCflNumber_ = CFL_number;
	sim.maxTimesteps_ = max_timesteps;
	sim.stopTime_ = max_time;
	sim.plotfileInterval_ = -1;
	amrex::ParmParse pp; // max_timesteps stays a deck / CLI knob
	pp.query("max_timesteps", sim.maxTimesteps_);

	sim.setInitialConditions();
	sim.evolve();
	amrex::Print() << "radiation: " << sim.radiationCellUpdates_ << " cell updates, " << sim.radSolves_ << " solves, " << sim.radNewtonIterations_
		       << " Newton iterations (max " << sim.radMaxNewtonIterations_ << " per solve)\n";

	// radiation temperature along x (one box per row of cells in this 1-D build: concatenate the valid cells)
	std::vector<double> xs, Trad;
	auto const &mf = sim.state_new_cc_[0];
	int const nx = sim.geom[0].Domain().length(0);
	xs.resize(nx);
	Trad.resize(nx);
	for (int b = 0; b < mf.size(); ++b) {
		auto h = mf.copyToHost(b);
		amrex::Array4<double> a(h.data(), mf.fabbox(b), mf.nComp());
		amrex::ParallelFor(mf.validbox(b), [&](int i, int j, int k) {
			xs[i] = Lx * ((i + 0.5) / static_cast<double>(nx));
			Trad[i] = std::pow(a(i, j, k, RadSystem<ShockProblem>::radEnergy_index) / a_rad, 1. / 4.) / T0;
		});
	}

	// exact solution (x [cm], rho, vel, Tmat, Trad, Frad/c; temperatures in units of T0)
	std::string filename = "../extern/LowrieEdwards/shock.txt";
	amrex::ParmParse("qk").query("shock_exact", filename);
	std::ifstream fstream(filename, std::ios::in);
	const double error_tol = 0.005;
	double rel_error = NAN;
	if (fstream.is_open()) {
		std::vector<double> xs_exact, Trad_exact;
		std::string line;
		std::getline(fstream, line); // header
		while (std::getline(fstream, line)) {
			std::istringstream iss(line);
			std::vector<double> row;
			for (double value = NAN; iss >> value;) {
				row.push_back(value);
			}
			if (row.size() >= 5 && row[0] > 0.0 && row[0] < Lx) {
				xs_exact.push_back(row[0]);
				Trad_exact.push_back(row[4]);
			}
		}
		double err_norm = 0., sol_norm = 0.;
		for (size_t n = 0; n < xs_exact.size(); ++n) {
			double const Trad_interp = interpolate_value(xs_exact[n], xs.data(), Trad.data(), nx);
			err_norm += std::abs(Trad_interp - Trad_exact[n]);
			sol_norm += std::abs(Trad_exact[n]);
		}
		rel_error = err_norm / sol_norm;
		sim.errorNorm_ = rel_error;
		amrex::Print() << "Relative L1 error norm = " << rel_error << std::endl;
	} else {
		amrex::Print() << "cannot open " << filename << "\n";
	}
	qkDumpState(sim);
	return ((rel_error > error_tol) || std::isnan(rel_error)) ? 1 : 0;
}
