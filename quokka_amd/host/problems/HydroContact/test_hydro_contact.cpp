// Stationary isolated contact discontinuity — problem generator written against the reference's surface (cf. reference
// src/problems/HydroContact/test_hydro_contact.cpp; deck tests/contact_wave.in).  With the HLLC solver the error must be zero in
// EVERY digit (Toro 1998, Sec. 10.7): the exit status is `errorNorm_ > 0.0`.  The sharpest discriminator the reference has for the
// operation order of the (un-vendored) gamma-law EOS and for anything that perturbs a flux; two (empty) passive scalars ride along.
#include "AMReX_BC_TYPES.H"
#include "AMReX_MultiFab.H"
#include "AMReX_Print.H"

#include "QuokkaSimulation.hpp"
#include "hydro/hydro_system.hpp"

struct ContactProblem {
};

template <> struct quokka::EOS_Traits<ContactProblem> {
	static constexpr double gamma = 1.4;
	static constexpr double mean_molecular_weight = C::m_u;
	static constexpr double boltzmann_constant = C::k_B;
	static constexpr double cs_isothermal = std::numeric_limits<double>::quiet_NaN();
};

template <> struct Physics_Traits<ContactProblem> {
	static constexpr bool is_hydro_enabled = true;
	static constexpr int numMassScalars = 0;
	static constexpr int numPassiveScalars = numMassScalars + 2;
	static constexpr bool is_radiation_enabled = false;
	static constexpr bool is_mhd_enabled = false;
	static constexpr int nGroups = 1;
};

constexpr double v_contact = 0.0; // contact wave velocity

namespace
{
void contact(amrex::Array4<amrex::Real> const &U, int i, int j, int k, double x)
{
	double const rho = (x < 0.5) ? 1.4 : 1.0, vx = v_contact, P = 1.0;
	for (int n = 0; n < U.nComp(); ++n) {
		U(i, j, k, n) = 0.;
	}
	auto const Eint = quokka::EOS<ContactProblem>::ComputeEintFromPres(rho, P);
	U(i, j, k, HydroSystem<ContactProblem>::density_index) = rho;
	U(i, j, k, HydroSystem<ContactProblem>::x1Momentum_index) = rho * vx;
	U(i, j, k, HydroSystem<ContactProblem>::energy_index) = Eint + 0.5 * rho * (vx * vx);
	U(i, j, k, HydroSystem<ContactProblem>::internalEnergy_index) = Eint;
}
} // namespace

template <> void QuokkaSimulation<ContactProblem>::setInitialConditionsOnGrid(quokka::grid const &grid_elem)
{
	auto const dx = grid_elem.dx_;
	auto const prob_lo = grid_elem.prob_lo_;
	const amrex::Array4<double> &state_cc = grid_elem.array_;
	amrex::ParallelFor(grid_elem.indexRange_, [=](int i, int j, int k) { contact(state_cc, i, j, k, prob_lo[0] + (i + 0.5) * dx[0]); });
}

template <>
void QuokkaSimulation<ContactProblem>::computeReferenceSolution(amrex::MultiFab &ref, amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const &dx,
								amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const &prob_lo)
{
	for (int b = 0; b < ref.size(); ++b) { // the contact does not move
		std::vector<double> h(static_cast<size_t>(ref.fabbox(b).numPts()) * ref.nComp(), 0.0);
		amrex::Array4<double> stateExact(h.data(), ref.fabbox(b), ref.nComp());
		amrex::ParallelFor(ref.validbox(b), [&](int i, int j, int k) { contact(stateExact, i, j, k, prob_lo[0] + (i + 0.5) * dx[0]); });
		ref.copyFromHost(b, h);
	}
}

auto problem_main() -> int
{
	const int ncomp_cc = Physics_Indices<ContactProblem>::nvarTotal_cc;
	amrex::Vector<amrex::BCRec> BCs_cc(ncomp_cc); // int_dir: periodic

	QuokkaSimulation<ContactProblem> sim(BCs_cc);
	sim.stopTime_ = 2.0;
	sim.cflNumber_ = 0.8;
	sim.maxTimesteps_ = 2000;
	sim.computeReferenceSolution_ = true;
	sim.plotfileInterval_ = -1;

	sim.setInitialConditions();
	sim.evolve();
	qkDumpState(sim);

	// the error should be *exactly* (i.e., to *every* digit) zero
	const double error_tol = 0.0; // this is not a typo
	int status = 0;
	if (!(sim.errorNorm_ <= error_tol)) {
		status = 1;
	}
	amrex::Print() << "Finished." << std::endl;
	return status;
}
