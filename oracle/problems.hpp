// ORACLE — TEST INFRASTRUCTURE ONLY (see grid.hpp header).
//
// problems.hpp: restatement of the problem generators on the hot path's configs
//   Sod shocktube   reference src/problems/HydroShocktube/test_hydro_shocktube.cpp:27-144,340-383, tests/shocktube.in
//   contact wave    reference src/problems/HydroContact/test_hydro_contact.cpp:21-82,188-222,  tests/contact_wave.in
//   Sedov blast     reference src/problems/HydroBlast3D/test_hydro3d_blast.cpp:22-115,222-262, tests/blast_unigrid_*.in
#ifndef ORACLE_PROBLEMS_HPP_
#define ORACLE_PROBLEMS_HPP_

#include <cmath>

#include "hydro_sim.hpp"

namespace oracle
{

inline void setupGeometry(HydroSim &sim, int ndim, int const n_cell[3], double const prob_lo[3], double const prob_hi[3], int const periodic[3],
			  int const max_grid_size[3])
{
	sim.geom.ndim = ndim;
	sim.hydro.tr.ndim = ndim;
	for (int d = 0; d < 3; ++d) {
		bool const active = d < ndim;
		sim.geom.domain.lo[d] = 0;
		sim.geom.domain.hi[d] = active ? n_cell[d] - 1 : 0;
		sim.geom.prob_lo[d] = prob_lo[d];
		sim.geom.prob_hi[d] = prob_hi[d];
		sim.geom.dx[d] = active ? (prob_hi[d] - prob_lo[d]) / n_cell[d] : 1.0;
		sim.geom.periodic[d] = active ? periodic[d] : 0;
	}
	int mgs[3];
	for (int d = 0; d < 3; ++d) {
		mgs[d] = (d < ndim) ? max_grid_size[d] : 1;
	}
	sim.grids = chopDomain(sim.geom.domain, mgs);
}

template <typename F> inline void forEachValidCell(HydroSim &sim, F &&f)
{
	for (int b = 0; b < sim.state_new_cc_.size(); ++b) {
		auto arr = sim.state_new_cc_.array(b);
		Box const &r = sim.grids[b];
		for (int k = r.lo[2]; k <= r.hi[2]; ++k) {
			for (int j = r.lo[1]; j <= r.hi[1]; ++j) {
				for (int i = r.lo[0]; i <= r.hi[0]; ++i) {
					f(arr, i, j, k);
				}
			}
		}
	}
}

// ---------------------------------------------------------------- Sod shocktube
// test_hydro_shocktube.cpp:47-50
constexpr double sod_rho_L = 10.0;
constexpr double sod_P_L = 100.0;
constexpr double sod_rho_R = 1.0;
constexpr double sod_P_R = 1.0;

inline void setupSod(HydroSim &sim)
{
	// EOS_Traits :29-33 ; HydroSystem_Traits default (reconstruct_eint = true)
	sim.hydro.tr.eos.tr.gamma = 1.4;
	sim.hydro.tr.eos.tr.mean_molecular_weight = C::m_u;
	sim.hydro.tr.eos.tr.boltzmann_constant = C::k_B;
	sim.hydro.tr.reconstruct_eint = true;
	sim.hydro.tr.nscalars = 0;
	sim.ncomp_cc = kNumHydroVars;
	// problem_main :354-370 and deck tests/shocktube.in (cfl = 0.6, reconstruction_order = 3)
	sim.BCs_cc.assign(sim.ncomp_cc, BCRec{});
	sim.BCs_cc[0].lo[0] = ext_dir; // only BCs_cc[0] is set in the reference loop (sic)
	sim.BCs_cc[0].hi[0] = ext_dir;
	sim.cflNumber_ = 0.6;
	sim.reconstructionOrder_ = 3;
	sim.stopTime_ = 0.4;
	sim.maxTimesteps_ = 8000;

	const double gamma = 1.4;
	// setCustomBoundaryConditions :94-144
	sim.customBC = [gamma](int i, int j, int k, Array4<double> const &consVar, Box const &dom, double /*time*/) {
		int const numcomp = consVar.ncomp;
		if (i < dom.lo[0]) {
			for (int n = 0; n < numcomp; ++n) {
				consVar(i, j, k, n) = 0;
			}
			consVar(i, j, k, energy_index) = sod_P_L / (gamma - 1.);
			consVar(i, j, k, internalEnergy_index) = sod_P_L / (gamma - 1.);
			consVar(i, j, k, density_index) = sod_rho_L;
			consVar(i, j, k, x1Momentum_index) = 0.;
			consVar(i, j, k, x2Momentum_index) = 0.;
			consVar(i, j, k, x3Momentum_index) = 0.;
		} else if (i >= dom.hi[0]) { // sic: `>=` in the reference (:130); only ghost cells are ever passed
			for (int n = 0; n < numcomp; ++n) {
				consVar(i, j, k, n) = 0;
			}
			consVar(i, j, k, energy_index) = sod_P_R / (gamma - 1.);
			consVar(i, j, k, internalEnergy_index) = sod_P_R / (gamma - 1.);
			consVar(i, j, k, density_index) = sod_rho_R;
			consVar(i, j, k, x1Momentum_index) = 0.;
			consVar(i, j, k, x2Momentum_index) = 0.;
			consVar(i, j, k, x3Momentum_index) = 0.;
		}
	};

	sim.define();
	// setInitialConditionsOnGrid :52-92
	double const dx0 = sim.geom.dx[0];
	double const lo0 = sim.geom.prob_lo[0];
	int const ncomp = sim.ncomp_cc;
	forEachValidCell(sim, [=](Array4<double> const &state_cc, int i, int j, int k) {
		double const x = lo0 + (i + 0.5) * dx0;
		const double vx = 0.0;
		double rho = NAN;
		double P = NAN;
		if (x < 2.0) {
			rho = sod_rho_L;
			P = sod_P_L;
		} else {
			rho = sod_rho_R;
			P = sod_P_R;
		}
		for (int n = 0; n < ncomp; ++n) {
			state_cc(i, j, k, n) = 0.;
		}
		state_cc(i, j, k, density_index) = rho;
		state_cc(i, j, k, x1Momentum_index) = rho * vx;
		state_cc(i, j, k, x2Momentum_index) = 0.;
		state_cc(i, j, k, x3Momentum_index) = 0.;
		state_cc(i, j, k, energy_index) = P / (gamma - 1.) + 0.5 * rho * (vx * vx);
		state_cc(i, j, k, internalEnergy_index) = P / (gamma - 1.);
	});
	sim.finishInitialConditions();
}

// ---------------------------------------------------------------- stationary contact wave
inline void setupContact(HydroSim &sim, int nscalars = 0)
{
	sim.hydro.tr.eos.tr.gamma = 1.4; // :24-28
	sim.hydro.tr.eos.tr.mean_molecular_weight = C::m_u;
	sim.hydro.tr.eos.tr.boltzmann_constant = C::k_B;
	sim.hydro.tr.reconstruct_eint = true;
	sim.hydro.tr.nscalars = nscalars; // reference uses 2 passive scalars (all zero), :34
	sim.ncomp_cc = kNumHydroVars + nscalars;
	sim.BCs_cc.assign(sim.ncomp_cc, BCRec{}); // periodic (:192-200)
	sim.stopTime_ = 2.0;			  // :205-207
	sim.cflNumber_ = 0.8;
	sim.maxTimesteps_ = 2000;

	sim.define();
	double const dx0 = sim.geom.dx[0];
	double const lo0 = sim.geom.prob_lo[0];
	int const ncomp = sim.ncomp_cc;
	EOS const eos = sim.hydro.tr.eos;
	const double v_contact = 0.0; // :42
	// :44-82
	forEachValidCell(sim, [=](Array4<double> const &state_cc, int i, int j, int k) {
		double const x = lo0 + (i + 0.5) * dx0;
		double vx = NAN, rho = NAN, P = NAN;
		if (x < 0.5) {
			rho = 1.4;
			vx = v_contact;
			P = 1.0;
		} else {
			rho = 1.0;
			vx = v_contact;
			P = 1.0;
		}
		for (int n = 0; n < ncomp; ++n) {
			state_cc(i, j, k, n) = 0.;
		}
		state_cc(i, j, k, density_index) = rho;
		state_cc(i, j, k, x1Momentum_index) = rho * vx;
		state_cc(i, j, k, x2Momentum_index) = 0.;
		state_cc(i, j, k, x3Momentum_index) = 0.;
		state_cc(i, j, k, energy_index) = eos.ComputeEintFromPres(rho, P) + 0.5 * rho * (vx * vx);
		state_cc(i, j, k, internalEnergy_index) = eos.ComputeEintFromPres(rho, P);
	});
	sim.finishInitialConditions();
}

// ---------------------------------------------------------------- Sedov blast (octant)
inline void setupSedov(HydroSim &sim)
{
	sim.hydro.tr.eos.tr.gamma = 1.4; // :30-34
	sim.hydro.tr.eos.tr.mean_molecular_weight = C::m_u;
	sim.hydro.tr.eos.tr.boltzmann_constant = C::k_B;
	sim.hydro.tr.reconstruct_eint = false; // :36-38
	sim.hydro.tr.nscalars = 0;
	sim.ncomp_cc = kNumHydroVars;

	// :224-251 octant symmetry
	sim.BCs_cc.assign(sim.ncomp_cc, BCRec{});
	for (int n = 0; n < sim.ncomp_cc; ++n) {
		for (int i = 0; i < sim.geom.ndim; ++i) {
			bool const isNormalComp = (n == x1Momentum_index + i);
			sim.BCs_cc[n].lo[i] = isNormalComp ? reflect_odd : reflect_even;
			sim.BCs_cc[n].hi[i] = isNormalComp ? reflect_odd : reflect_even;
		}
	}
	sim.reconstructionOrder_ = 3; // :259-261
	sim.stopTime_ = 1.0;
	sim.cflNumber_ = 0.3;

	sim.define();
	// :52-58, :60-115
	const double rho = 1.0;
	double E_blast = 0.851072;
	E_blast /= 8.0;
	double const cell_vol = sim.geom.dx[0] * sim.geom.dx[1] * sim.geom.dx[2];
	int const ncomp = sim.ncomp_cc;
	forEachValidCell(sim, [=](Array4<double> const &state_cc, int i, int j, int k) {
		double rho_e = NAN;
		if ((i == 0) && (j == 0) && (k == 0)) {
			rho_e = E_blast / cell_vol;
		} else {
			rho_e = 1.0e-10 * (E_blast / cell_vol);
		}
		for (int n = 0; n < ncomp; ++n) {
			state_cc(i, j, k, n) = 0.;
		}
		state_cc(i, j, k, density_index) = rho;
		state_cc(i, j, k, x1Momentum_index) = 0;
		state_cc(i, j, k, x2Momentum_index) = 0;
		state_cc(i, j, k, x3Momentum_index) = 0;
		state_cc(i, j, k, energy_index) = rho_e;
	});
	sim.finishInitialConditions();
}

} // namespace oracle

#endif // ORACLE_PROBLEMS_HPP_
