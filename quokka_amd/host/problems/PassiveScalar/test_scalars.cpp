// Advected contact discontinuity carrying a passive scalar, periodic, optionally with one refined level — problem generator written
// against the reference's surface (cf. reference src/problems/PassiveScalar/test_scalars.cpp; deck tests/PassiveScalar.in).
// Compiled against quokka_amd/host (AMReX is absent), 1-D build.  Exit status = the reference's two criteria: the scalar is conserved
// to 1e-14 (relative) and the state after two time units (four box crossings) is within 0.008 of the initial one (relative rms L1).
#include <cmath>

#include "AMReX_BC_TYPES.H"
#include "AMReX_MultiFab.H"
#include "AMReX_ParmParse.H"
#include "AMReX_Print.H"

#include "QuokkaSimulation.hpp"
#include "hydro/hydro_system.hpp"

struct ScalarProblem {
};

template <> struct quokka::EOS_Traits<ScalarProblem> {
	static constexpr double gamma = 1.4;
	static constexpr double mean_molecular_weight = C::m_u;
	static constexpr double boltzmann_constant = C::k_B;
	static constexpr double cs_isothermal = std::numeric_limits<double>::quiet_NaN();
};

template <> struct Physics_Traits<ScalarProblem> {
	static constexpr bool is_hydro_enabled = true;
	static constexpr int numMassScalars = 0;
	static constexpr int numPassiveScalars = numMassScalars + 1;
	static constexpr bool is_radiation_enabled = false;
	static constexpr bool is_mhd_enabled = false;
	static constexpr int nGroups = 1;
};

constexpr double v_contact = 2.0; // contact wave velocity
static bool scalar_is_conserved = false;

namespace
{
// the state at t = 0, which is also the exact solution after a whole number of box crossings
void contactState(amrex::Array4<amrex::Real> const &U, int i, int j, int k, double x)
{
	bool const left = x < 0.5;
	double const rho = left ? 1.4 : 1.0, vx = v_contact, P = 1.0;
	for (int n = 0; n < U.nComp(); ++n) {
		U(i, j, k, n) = 0.;
	}
	auto const Eint = quokka::EOS<ScalarProblem>::ComputeEintFromPres(rho, P);
	U(i, j, k, HydroSystem<ScalarProblem>::density_index) = rho;
	U(i, j, k, HydroSystem<ScalarProblem>::x1Momentum_index) = rho * vx;
	U(i, j, k, HydroSystem<ScalarProblem>::energy_index) = Eint + 0.5 * rho * (vx * vx);
	U(i, j, k, HydroSystem<ScalarProblem>::internalEnergy_index) = Eint;
	U(i, j, k, HydroSystem<ScalarProblem>::scalar0_index) = left ? 1.0 : 0.0;
}
} // namespace

template <> void QuokkaSimulation<ScalarProblem>::setInitialConditionsOnGrid(quokka::grid const &grid_elem)
{
	auto const dx = grid_elem.dx_;
	auto const prob_lo = grid_elem.prob_lo_;
	const amrex::Array4<double> &state_cc = grid_elem.array_;
	amrex::ParallelFor(grid_elem.indexRange_, [=](int i, int j, int k) { contactState(state_cc, i, j, k, prob_lo[0] + (i + 0.5) * dx[0]); });
}

template <>
void QuokkaSimulation<ScalarProblem>::computeReferenceSolution(amrex::MultiFab &ref, amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const &dx,
							       amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const &prob_lo)
{
	for (int b = 0; b < ref.size(); ++b) {
		std::vector<double> h(static_cast<size_t>(ref.fabbox(b).numPts()) * ref.nComp(), 0.0);
		amrex::Array4<double> stateExact(h.data(), ref.fabbox(b), ref.nComp());
		amrex::ParallelFor(ref.validbox(b), [&](int i, int j, int k) { contactState(stateExact, i, j, k, prob_lo[0] + (i + 0.5) * dx[0]); });
		ref.copyFromHost(b, h);
	}
	const amrex::Real scalar_ref = ref.sum(HydroSystem<ScalarProblem>::scalar0_index);
	const amrex::Real scalar_sol = state_new_cc_[0].sum(HydroSystem<ScalarProblem>::scalar0_index);
	const amrex::Real reldiff = std::abs((scalar_sol - scalar_ref) / scalar_ref);
	const amrex::Real reltol = 1.0e-14;
	scalar_is_conserved = reldiff < reltol;
	if (scalar_is_conserved) {
		amrex::Print() << "Passive scalar is conserved to a relative difference of " << reldiff << "\n";
	} else {
		amrex::Print() << "Passive scalar differs by a factor of " << reldiff << "\n";
	}
}

template <> void QuokkaSimulation<ScalarProblem>::ErrorEst(int /*lev*/, amrex::TagBoxArray &tags, amrex::Real /*time*/, int /*ngrow*/)
{
	// tag cells for refinement: |(rho(i+1) - rho(i-1)) / 2| / rho > eta (the centred-difference form of the library with dx = 1)
	const amrex::Real eta_threshold = 0.05;
	tagCenteredGradient(tags, HydroSystem<ScalarProblem>::density_index, /*dir=*/0, eta_threshold, /*q_min=*/-1.0e300, /*min_inclusive=*/true, /*dx=*/1.0);
}

auto problem_main() -> int
{
	const int ncomp_cc = Physics_Indices<ScalarProblem>::nvarTotal_cc;
	amrex::Vector<amrex::BCRec> BCs_cc(ncomp_cc); // int_dir everywhere: periodic

	QuokkaSimulation<ScalarProblem> sim(BCs_cc);
	sim.computeReferenceSolution_ = true;
	sim.setInitialConditions();
	sim.evolve();
	qkDumpState(sim);

	const double error_tol = 0.008;
	int status = 0;
	if (!(sim.errorNorm_ <= error_tol) || !scalar_is_conserved) {
		status = 1;
	}
	amrex::Print() << "Finished." << std::endl;
	return status;
}
