// Free-streaming radiation front in 1-D — problem generator written against the reference's surface (cf. reference
// src/problems/RadStreaming/test_radiation_streaming.cpp; deck tests/RadStreaming.in).  Radiation only (hydro disabled), Levermore
// closure at reduced flux 1, beta_order = 0.  Exit status = the reference's pass criterion: relative L1 error of E_rad against the
// step function E_rad = 1 for x <= c_hat t below 0.01.
#include <cmath>
#include <vector>

#include "AMReX_BC_TYPES.H"
#include "AMReX_Print.H"

#include "QuokkaSimulation.hpp"
#include "radiation/radiation_system.hpp"

struct StreamingProblem {
};

constexpr double initial_Erad = 1.0e-5;
constexpr double initial_Egas = 1.0e-5;
constexpr double c = 1.0;	   // speed of light
constexpr double chat = 0.2;	   // reduced speed of light
constexpr double kappa0 = 1.0e-10; // opacity
constexpr double rho = 1.0;

template <> struct quokka::EOS_Traits<StreamingProblem> {
	static constexpr double mean_molecular_weight = 1.0;
	static constexpr double boltzmann_constant = 1.0;
	static constexpr double gamma = 5. / 3.;
	static constexpr double cs_isothermal = std::numeric_limits<double>::quiet_NaN();
};

template <> struct Physics_Traits<StreamingProblem> {
	static constexpr bool is_hydro_enabled = false;
	static constexpr int numMassScalars = 0;
	static constexpr int numPassiveScalars = numMassScalars + 0;
	static constexpr bool is_radiation_enabled = true;
	static constexpr bool is_mhd_enabled = false;
	static constexpr int nGroups = 1;
};

template <> struct RadSystem_Traits<StreamingProblem> {
	static constexpr double c_light = c;
	static constexpr double c_hat = chat;
	static constexpr double radiation_constant = 1.0;
	static constexpr double Erad_floor = initial_Erad;
	static constexpr int beta_order = 0;
};

template <> auto RadSystem<StreamingProblem>::ComputePlanckOpacity(const double /*rho*/, const double /*Tgas*/) -> amrex::Real { return kappa0; }
template <> auto RadSystem<StreamingProblem>::ComputeFluxMeanOpacity(const double /*rho*/, const double /*Tgas*/) -> amrex::Real { return kappa0; }

namespace
{
void gasAtRest(amrex::Array4<amrex::Real> const &U, int i, int j, int k)
{
	U(i, j, k, RadSystem<StreamingProblem>::gasEnergy_index) = initial_Egas;
	U(i, j, k, RadSystem<StreamingProblem>::gasDensity_index) = rho;
	U(i, j, k, RadSystem<StreamingProblem>::gasInternalEnergy_index) = initial_Egas;
	U(i, j, k, RadSystem<StreamingProblem>::x1GasMomentum_index) = 0.;
	U(i, j, k, RadSystem<StreamingProblem>::x2GasMomentum_index) = 0.;
	U(i, j, k, RadSystem<StreamingProblem>::x3GasMomentum_index) = 0.;
}
void radiation(amrex::Array4<amrex::Real> const &U, int i, int j, int k, double Erad, double Frad)
{
	U(i, j, k, RadSystem<StreamingProblem>::radEnergy_index) = Erad;
	U(i, j, k, RadSystem<StreamingProblem>::x1RadFlux_index) = Frad;
	U(i, j, k, RadSystem<StreamingProblem>::x2RadFlux_index) = 0;
	U(i, j, k, RadSystem<StreamingProblem>::x3RadFlux_index) = 0;
}
} // namespace

template <> void QuokkaSimulation<StreamingProblem>::setInitialConditionsOnGrid(quokka::grid const &grid_elem)
{
	const amrex::Array4<double> &state_cc = grid_elem.array_;
	amrex::ParallelFor(grid_elem.indexRange_, [=](int i, int j, int k) {
		radiation(state_cc, i, j, k, initial_Erad, 0);
		gasAtRest(state_cc, i, j, k);
	});
}

// (does not consult the BCRec: the extrapolated cells beyond the upper face are overwritten as well, as in the reference)
template <>
void AMRSimulation<StreamingProblem>::setCustomBoundaryConditions(const amrex::IntVect &iv, amrex::Array4<amrex::Real> const &consVar, int /*dcomp*/,
								  int /*numcomp*/, amrex::GeometryData const &geom, const amrex::Real /*time*/,
								  const amrex::BCRec * /*bcr*/, int /*bcomp*/, int /*orig_comp*/)
{
	auto const i = iv.toArray()[0];
	int const j = 0, k = 0;
	amrex::Box const &box = geom.Domain();
	if (i < box.loVect3d()[0]) { // streaming incident flux: F = c E
		const double Erad = 1.0;
		radiation(consVar, i, j, k, Erad, c * Erad);
	} else if (i >= box.hiVect3d()[0]) {
		radiation(consVar, i, j, k, initial_Erad, 0);
	}
	gasAtRest(consVar, i, j, k);
}

auto problem_main() -> int
{
	const double CFL_number = 0.8;
	const double dt_max = 1e-2;
	const double tmax = 1.0;
	const int max_timesteps = 5000;

	constexpr int nvars = RadSystem<StreamingProblem>::nvar_;
	amrex::Vector<amrex::BCRec> BCs_cc(nvars);
	for (int n = 0; n < nvars; ++n) {
		BCs_cc[n].setLo(0, amrex::BCType::ext_dir);  // Dirichlet x1
		BCs_cc[n].setHi(0, amrex::BCType::foextrap); // extrapolate x1
	}

	QuokkaSimulation<StreamingProblem> sim(BCs_cc);
	sim.radiationReconstructionOrder_ = 3; // PPM
	sim.stopTime_ = tmax;
	sim.radiationCflNumber_ = CFL_number;
	sim.maxDt_ = dt_max;
	sim.maxTimesteps_ = max_timesteps;
	sim.plotfileInterval_ = -1;

	sim.setInitialConditions();
	sim.evolve();

	auto const &mf = sim.state_new_cc_[0];
	int const nx = sim.geom[0].Domain().length(0);
	double err_norm = 0., sol_norm = 0.;
	for (int b = 0; b < mf.size(); ++b) {
		auto h = mf.copyToHost(b);
		amrex::Array4<double> a(h.data(), mf.fabbox(b), mf.nComp());
		amrex::ParallelFor(mf.validbox(b), [&](int i, int j, int k) {
			double const x = sim.geom[0].ProbLo(0) + (i + 0.5) * sim.geom[0].CellSize(0);
			double const erad_exact = (x <= chat * tmax) ? 1.0 : 0.0;
			err_norm += std::abs(a(i, j, k, RadSystem<StreamingProblem>::radEnergy_index) - erad_exact);
			sol_norm += std::abs(erad_exact);
		});
	}
	const double rel_err_norm = err_norm / sol_norm;
	const double rel_err_tol = 0.01;
	sim.errorNorm_ = rel_err_norm;
	amrex::Print() << "Relative L1 norm = " << rel_err_norm << " (" << nx << " cells)" << std::endl;
	qkDumpState(sim);
	return (rel_err_norm < rel_err_tol) ? 0 : 1;
}
