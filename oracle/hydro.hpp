// ORACLE — TEST INFRASTRUCTURE ONLY (see grid.hpp header).
//
// hydro.hpp: restatement of
//   reference src/hydro/HydroState.hpp:10-23
//   reference src/hydro/HLLC.hpp:22-153       (quokka::Riemann::HLLC)
//   reference src/hydro/LLF.hpp:16-43         (quokka::Riemann::LLF)
//   reference src/hydro/hydro_system.hpp      (HydroSystem<problem_t>, line refs per function)
// nmscalars (mass scalars) = 0 throughout: none of the configs uses them.
// Passive scalars (nscalars) are carried where the reference loops over them.
#ifndef ORACLE_HYDRO_HPP_
#define ORACLE_HYDRO_HPP_

#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>

#include "eos.hpp"
#include "grid.hpp"
#include "hyperbolic.hpp"

namespace oracle
{

constexpr int kMaxScalars = 4;
constexpr int kNumHydroVars = 6;			// physics_numVars.hpp:8
constexpr int kMaxVars = kNumHydroVars + kMaxScalars;

// HydroSystem::consVarIndex (hydro_system.hpp:54-62), hydroFirstIndex = 0 (physics_info.hpp:37)
enum consVarIndex { density_index = 0, x1Momentum_index, x2Momentum_index, x3Momentum_index, energy_index, internalEnergy_index, scalar0_index };
// HydroSystem::primVarIndex (hydro_system.hpp:64-72)
enum primVarIndex { primDensity_index = 0, x1Velocity_index, x2Velocity_index, x3Velocity_index, pressure_index, primEint_index, primScalar0_index };

enum RiemannSolver : int { riemann_HLLC = 0, riemann_LLF = 1, riemann_HLLD = 2 }; // hydro_system.hpp:43
enum redoFlagVal : int { redo_none = 0, redo_redo = 1 }; // hyperbolic_system.hpp:34

// runtime stand-in for the compile-time traits a problem supplies
struct HydroTraits {
	EOS eos;			 // quokka::EOS_Traits<problem_t>
	bool reconstruct_eint = true;	 // HydroSystem_Traits<problem_t> (hydro_system.hpp:38-41)
	int nscalars = 0;		 // Physics_Traits::numPassiveScalars
	int nmscalars = 0;		 // Physics_Traits::numMassScalars (the first nmscalars passive scalars are partial densities)
	int ndim = 3;			 // AMREX_SPACEDIM of the build
	[[nodiscard]] auto nvar() const -> int { return kNumHydroVars + nscalars; }
	[[nodiscard]] auto is_eos_isothermal() const -> bool { return eos.tr.gamma == 1.0; } // hydro_system.hpp:133
	[[nodiscard]] auto gamma() const -> double { return eos.tr.gamma; }
	[[nodiscard]] auto cs_iso() const -> double { return eos.tr.cs_isothermal; }
};

using valarray = std::array<double, kMaxVars>; // quokka::valarray<double, nvar_>; tail is zero-filled

// HydroState.hpp:10-23
struct HydroState {
	double rho, u, v, w, P, cs, E, Eint, by, bz;
	double scalar[kMaxScalars];
};

// HLLC.hpp:22-153
inline auto HLLC(HydroTraits const &tr, HydroState const &sL, HydroState const &sR, const double gamma, const double du, const double dw) -> valarray
{
	const int fluxdim = tr.nvar();
	const int N_scalars = tr.nscalars;

	// :27-36 Roe averages
	const double wl = std::sqrt(sL.rho);
	const double wr = std::sqrt(sR.rho);
	const double norm = 1. / (wl + wr);
	const double u_tilde = (wl * sL.u + wr * sR.u) * norm;
	const double v_tilde = (wl * sL.v + wr * sR.v) * norm;
	const double w_tilde = (wl * sL.w + wr * sR.w) * norm;
	const double vsq_tilde = u_tilde * u_tilde + v_tilde * v_tilde + w_tilde * w_tilde;
	const double H_L = (sL.E + sL.P) / sL.rho;
	const double H_R = (sR.E + sR.P) / sR.rho;
	const double H_tilde = (wl * H_L + wr * H_R) * norm;
	double cs_tilde = NAN;

	const double dU = sL.u - sR.u;
	double S_L = NAN;
	double S_R = NAN;
	if (gamma != 1.0) {
		// :47-48
		auto dL = tr.eos.ComputeOtherDerivatives(sL.rho, sL.P);
		auto dR = tr.eos.ComputeOtherDerivatives(sR.rho, sR.P);
		const double dedr_L = dL.deint_dRho, dedp_L = dL.deint_dP, drdp_L = dL.dRho_dP, G_L = dL.G;
		const double dedr_R = dR.deint_dRho, dedp_R = dR.deint_dP, drdp_R = dR.dRho_dP, G_R = dR.G;

		// :52 eq. A.5a of Kershaw+1998
		const double C_tilde_rho = 0.5 * ((sL.Eint / sL.rho) + (sR.Eint / sR.rho) + sL.rho * dedr_L + sR.rho * dedr_R);
		// :55 eq. A.5b
		const double C_tilde_P = 0.5 * ((sL.Eint / sL.rho) * drdp_L + (sR.Eint / sR.rho) * drdp_R + sL.rho * dedp_L + sR.rho * dedp_R);

		// :58-65
		const double cs_exp = H_tilde - 0.5 * vsq_tilde - C_tilde_rho;
		if (cs_exp <= 0) {
			cs_tilde = 0.5 * (sL.cs + sR.cs);
		} else {
			cs_tilde = std::sqrt(cs_exp / C_tilde_P);
		}

		// :67-72
		const double s_NL = 0.5 * G_L * std::max(dU, 0.);
		const double s_NR = 0.5 * G_R * std::max(dU, 0.);
		S_L = std::min(sL.u - (sL.cs + s_NL), u_tilde - (cs_tilde + s_NL));
		S_R = std::max(sR.u + (sR.cs + s_NR), u_tilde + (cs_tilde + s_NR));
	} else {
		// :74-87
		cs_tilde = 0.5 * (sL.cs + sR.cs);
		const double G_gamma_L = 1.0;
		const double G_gamma_R = 1.0;
		const double G_L = 0.5 * (G_gamma_L + 1.);
		const double G_R = 0.5 * (G_gamma_R + 1.);
		const double s_NL = 0.5 * G_L * std::max(dU, 0.);
		const double s_NR = 0.5 * G_R * std::max(dU, 0.);
		S_L = std::min(sL.u - (sL.cs + s_NL), u_tilde - (cs_tilde + s_NL));
		S_R = std::max(sR.u + (sR.cs + s_NR), u_tilde + (cs_tilde + s_NR));
	}

	// :91-93 carbuncle correction
	const double cs_max = std::max(sL.cs, sR.cs);
	const double tp = std::min(1., (cs_max - std::min(du, 0.)) / (cs_max - std::min(dw, 0.)));
	const double theta = tp * tp * tp * tp;

	// :97-98
	const double S_star =
	    (theta * (sR.P - sL.P) + (sL.rho * sL.u * (S_L - sL.u) - sR.rho * sR.u * (S_R - sR.u))) / (sL.rho * (S_L - sL.u) - sR.rho * (S_R - sR.u));

	// :102-105
	const double vmag_L = std::sqrt(sL.u * sL.u + sL.v * sL.v + sL.w * sL.w);
	const double vmag_R = std::sqrt(sR.u * sR.u + sR.v * sR.v + sR.w * sR.w);
	const double chi = std::min(1., std::max(vmag_L, vmag_R) / cs_max);
	const double phi = chi * (2. - chi);

	// :107
	const double P_LR = 0.5 * (sL.P + sR.P) + 0.5 * phi * (sL.rho * (S_L - sL.u) * (S_star - sL.u) + sR.rho * (S_R - sR.u) * (S_star - sR.u));

	// :116-121 (initializer lists shorter than fluxdim zero-fill the tail, valarray.hpp:41-46)
	valarray D_L{}, D_R{}, D_star{}, U_L{}, U_R{};
	D_L[1] = 1.;
	D_L[4] = sL.u;
	D_R[1] = 1.;
	D_R[4] = sR.u;
	D_star[1] = 1.;
	D_star[4] = S_star;
	U_L[0] = sL.rho;
	U_L[1] = sL.rho * sL.u;
	U_L[2] = sL.rho * sL.v;
	U_L[3] = sL.rho * sL.w;
	U_L[4] = sL.E;
	U_L[5] = sL.Eint;
	U_R[0] = sR.rho;
	U_R[1] = sR.rho * sR.u;
	U_R[2] = sR.rho * sR.v;
	U_R[3] = sR.rho * sR.w;
	U_R[4] = sR.E;
	U_R[5] = sR.Eint;

	// :126-130
	for (int n = 0; n < N_scalars; ++n) {
		const int nstart = fluxdim - N_scalars;
		U_L[nstart + n] = sL.scalar[n];
		U_R[nstart + n] = sR.scalar[n];
	}

	valarray F{};
	const double SLP = S_L * P_LR; // `S_L * P_LR * D_star` evaluates (S_L * P_LR) first
	const double SRP = S_R * P_LR;
	const double dSL = S_L - S_star;
	const double dSR = S_R - S_star;
	for (int n = 0; n < fluxdim; ++n) {
		// :132-133
		const double F_L = sL.u * U_L[n] + sL.P * D_L[n];
		const double F_R = sR.u * U_R[n] + sR.P * D_R[n];
		// :135-136
		const double F_starL = (S_star * (S_L * U_L[n] - F_L) + SLP * D_star[n]) / dSL;
		const double F_starR = (S_star * (S_R * U_R[n] - F_R) + SRP * D_star[n]) / dSR;
		// :142-150
		if (S_L > 0.0) {
			F[n] = F_L;
		} else if ((S_star > 0.0) && (S_L <= 0.0)) {
			F[n] = F_starL;
		} else if ((S_star <= 0.0) && (S_R >= 0.0)) {
			F[n] = F_starR;
		} else {
			F[n] = F_R;
		}
	}
	return F;
}

// HLLD.hpp:26-334 (Miyoshi & Kusano 2005 as Athena++ codes it).  The reference calls it with bx = 0 and sL.by = sL.bz = sR.by = sR.bz = 0
// (hydro_system.hpp:987-1003, :1044-1048: "set to zero to test that the HLLD solver works with hydro only"); restated for general fields.
// Returns {rho, mx, my, mz, E, 0.0}: no flux of the auxiliary internal energy, none of the passive scalars (:331-332).
struct ConsHydro1D { // HLLD.hpp:21-29
	double rho, mx, my, mz, E, by, bz;
};
inline auto FastMagnetoSonicSpeed(double gamma, HydroState const &state, const double bx) -> double // HLLD.hpp:31-42
{
	double gp = gamma * state.P;
	double bx_sq = bx * bx;
	double byz_sq = state.by * state.by + state.bz * state.bz;
	double b_sq = bx_sq + byz_sq;
	double bgp_p = b_sq + gp;
	double bgp_m = b_sq - gp;
	return std::sqrt(0.5 * (bgp_p + std::sqrt(bgp_m * bgp_m + 4.0 * gp * byz_sq)) / state.rho);
}
inline auto HLLD(HydroState const &sL, HydroState const &sR, const double gamma, const double bx) -> valarray
{
	constexpr double DELTA = 1.0e-4; // :18
	auto SQUARE = [](double x) { return x * x; };
	ConsHydro1D u_L{}, u_R{}, f_x{}, f_L{}, f_R{}, u_star_L{}, u_dstar_L{}, u_dstar_R{}, u_star_R{};
	std::array<double, 5> spds{};
	double const bx_sq = SQUARE(bx);
	// :75-96 left and right conserved states
	double const pb_L = 0.5 * (bx_sq + (SQUARE(sL.by) + SQUARE(sL.bz)));
	double const pb_R = 0.5 * (bx_sq + (SQUARE(sR.by) + SQUARE(sR.bz)));
	double const ke_L = 0.5 * sL.rho * (SQUARE(sL.u) + (SQUARE(sL.v) + SQUARE(sL.w)));
	double const ke_R = 0.5 * sR.rho * (SQUARE(sR.u) + (SQUARE(sR.v) + SQUARE(sR.w)));
	u_L.rho = sL.rho;
	u_L.mx = sL.u * u_L.rho;
	u_L.my = sL.v * u_L.rho;
	u_L.mz = sL.w * u_L.rho;
	u_L.E = ke_L + pb_L + sL.P / (gamma - 1.0);
	u_L.by = sL.by;
	u_L.bz = sL.bz;
	u_R.rho = sR.rho;
	u_R.mx = sR.u * u_R.rho;
	u_R.my = sR.v * u_R.rho;
	u_R.mz = sR.w * u_R.rho;
	u_R.E = ke_R + pb_R + sR.P / (gamma - 1.0);
	u_R.by = sR.by;
	u_R.bz = sR.bz;
	// :100-104 outer wave speeds
	const double cfs_L = FastMagnetoSonicSpeed(gamma, sL, bx);
	const double cfs_R = FastMagnetoSonicSpeed(gamma, sR, bx);
	spds[0] = std::min(sL.u - cfs_L, sR.u - cfs_R);
	spds[4] = std::max(sL.u + cfs_L, sR.u + cfs_R);
	// :108-125 left and right fluxes
	double ptot_L = sL.P + pb_L;
	double ptot_R = sR.P + pb_R;
	f_L.rho = u_L.mx;
	f_L.mx = u_L.mx * sL.u + ptot_L - bx_sq;
	f_L.my = u_L.my * sL.u + bx * u_L.by;
	f_L.mz = u_L.mz * sL.u + bx * u_L.bz;
	f_L.E = sL.u * (u_L.E + ptot_L - bx_sq) - bx * (sL.v * u_L.by + sL.w * u_L.bz);
	f_L.by = u_L.by * sL.u - bx * sL.v;
	f_L.bz = u_L.bz * sL.u - bx * sL.w;
	f_R.rho = u_R.mx;
	f_R.mx = u_R.mx * sR.u + ptot_R - bx_sq;
	f_R.my = u_R.my * sR.u + bx * u_R.by;
	f_R.mz = u_R.mz * sR.u + bx * u_R.bz;
	f_R.E = sR.u * (u_R.E + ptot_R - bx_sq) - bx * (sR.v * u_R.by + sR.w * u_R.bz);
	f_R.by = u_R.by * sR.u - bx * sR.v;
	f_R.bz = u_R.bz * sR.u - bx * sR.w;
	// :129-145 middle and Alfven wave speeds
	double siui_L = spds[0] - sL.u;
	double siui_R = spds[4] - sR.u;
	spds[2] = (siui_R * u_R.mx - siui_L * u_L.mx + (ptot_L - ptot_R)) / (siui_R * u_R.rho - siui_L * u_L.rho);
	double sism_L = spds[0] - spds[2];
	double sism_R = spds[4] - spds[2];
	double sism_inv_L = 1.0 / sism_L;
	double sism_inv_R = 1.0 / sism_R;
	u_star_L.rho = u_L.rho * siui_L * sism_inv_L;
	u_star_R.rho = u_R.rho * siui_R * sism_inv_R;
	double u_star_rho_inv_L = 1.0 / u_star_L.rho;
	double u_star_rho_inv_R = 1.0 / u_star_R.rho;
	double rho_sqrt_L = std::sqrt(u_star_L.rho);
	double rho_sqrt_R = std::sqrt(u_star_R.rho);
	spds[1] = spds[2] - std::abs(bx) / rho_sqrt_L;
	spds[3] = spds[2] + std::abs(bx) / rho_sqrt_R;
	// :149-152 star-region total pressure
	double ptot_star_L = ptot_L - u_L.rho * siui_L * (spds[2] - sL.u);
	double ptot_star_R = ptot_R - u_R.rho * siui_R * (spds[2] - sR.u);
	double ptot_star = 0.5 * (ptot_star_L + ptot_star_R);
	// :154-174 left star state
	u_star_L.mx = u_star_L.rho * spds[2];
	if (std::abs(u_L.rho * siui_L * sism_L - bx_sq) < (DELTA)*ptot_star) {
		u_star_L.my = u_star_L.rho * sL.v;
		u_star_L.mz = u_star_L.rho * sL.w;
		u_star_L.by = u_L.by;
		u_star_L.bz = u_L.bz;
	} else {
		double tmp = bx * (siui_L - sism_L) / (u_L.rho * siui_L * sism_L - bx_sq);
		u_star_L.my = u_star_L.rho * (sL.v - u_L.by * tmp);
		u_star_L.mz = u_star_L.rho * (sL.w - u_L.bz * tmp);
		tmp = (u_L.rho * SQUARE(siui_L) - bx_sq) / (u_L.rho * siui_L * sism_L - bx_sq);
		u_star_L.by = u_L.by * tmp;
		u_star_L.bz = u_L.bz * tmp;
	}
	double vb_star_L = (u_star_L.mx * bx + (u_star_L.my * u_star_L.by + u_star_L.mz * u_star_L.bz)) * u_star_rho_inv_L;
	u_star_L.E = (siui_L * u_L.E - ptot_L * sL.u + ptot_star * spds[2] + bx * (sL.u * bx + (sL.v * u_L.by + sL.w * u_L.bz) - vb_star_L)) * sism_inv_L;
	// :176-196 right star state
	u_star_R.mx = u_star_R.rho * spds[2];
	if (std::abs(u_R.rho * siui_R * sism_R - bx_sq) < (DELTA)*ptot_star) {
		u_star_R.my = u_star_R.rho * sR.v;
		u_star_R.mz = u_star_R.rho * sR.w;
		u_star_R.by = u_R.by;
		u_star_R.bz = u_R.bz;
	} else {
		double tmp = bx * (siui_R - sism_R) / (u_R.rho * siui_R * sism_R - bx_sq);
		u_star_R.my = u_star_R.rho * (sR.v - u_R.by * tmp);
		u_star_R.mz = u_star_R.rho * (sR.w - u_R.bz * tmp);
		tmp = (u_R.rho * SQUARE(siui_R) - bx_sq) / (u_R.rho * siui_R * sism_R - bx_sq);
		u_star_R.by = u_R.by * tmp;
		u_star_R.bz = u_R.bz * tmp;
	}
	double vb_star_R = (u_star_R.mx * bx + (u_star_R.my * u_star_R.by + u_star_R.mz * u_star_R.bz)) * u_star_rho_inv_R;
	u_star_R.E = (siui_R * u_R.E - ptot_R * sR.u + ptot_star * spds[2] + bx * (sR.u * bx + (sR.v * u_R.by + sR.w * u_R.bz) - vb_star_R)) * sism_inv_R;
	// :198-237 double-star states
	if (0.5 * bx_sq < (DELTA)*ptot_star) {
		u_dstar_L = u_star_L;
		u_dstar_R = u_star_R;
	} else {
		double rho_sum_inv = 1.0 / (rho_sqrt_L + rho_sqrt_R);
		double bx_sign = (bx > 0.0 ? 1.0 : -1.0);
		u_dstar_L.rho = u_star_L.rho;
		u_dstar_R.rho = u_star_R.rho;
		u_dstar_L.mx = u_star_L.mx;
		u_dstar_R.mx = u_star_R.mx;
		double tmp = rho_sum_inv * (rho_sqrt_L * (u_star_L.my * u_star_rho_inv_L) + rho_sqrt_R * (u_star_R.my * u_star_rho_inv_R) +
					    bx_sign * (u_star_R.by - u_star_L.by));
		u_dstar_L.my = u_dstar_L.rho * tmp;
		u_dstar_R.my = u_dstar_R.rho * tmp;
		tmp = rho_sum_inv *
		      (rho_sqrt_L * (u_star_L.mz * u_star_rho_inv_L) + rho_sqrt_R * (u_star_R.mz * u_star_rho_inv_R) + bx_sign * (u_star_R.bz - u_star_L.bz));
		u_dstar_L.mz = u_dstar_L.rho * tmp;
		u_dstar_R.mz = u_dstar_R.rho * tmp;
		tmp = rho_sum_inv * (rho_sqrt_L * u_star_R.by + rho_sqrt_R * u_star_L.by +
				     bx_sign * rho_sqrt_L * rho_sqrt_R * ((u_star_R.my * u_star_rho_inv_R) - (u_star_L.my * u_star_rho_inv_L)));
		u_dstar_L.by = tmp;
		u_dstar_R.by = tmp;
		tmp = rho_sum_inv * (rho_sqrt_L * u_star_R.bz + rho_sqrt_R * u_star_L.bz +
				     bx_sign * rho_sqrt_L * rho_sqrt_R * ((u_star_R.mz * u_star_rho_inv_R) - (u_star_L.mz * u_star_rho_inv_L)));
		u_dstar_L.bz = tmp;
		u_dstar_R.bz = tmp;
		tmp = spds[2] * bx + (u_dstar_L.my * u_dstar_L.by + u_dstar_L.mz * u_dstar_L.bz) / u_dstar_L.rho;
		u_dstar_L.E = u_star_L.E - rho_sqrt_L * bx_sign * (vb_star_L - tmp);
		u_dstar_R.E = u_star_R.E + rho_sqrt_R * bx_sign * (vb_star_R - tmp);
	}
	// :241-271 flux increments across the waves (the states are overwritten by them)
	auto jump = [](double s, ConsHydro1D const &a, ConsHydro1D const &b) {
		return ConsHydro1D{s * (a.rho - b.rho), s * (a.mx - b.mx), s * (a.my - b.my), s * (a.mz - b.mz), s * (a.E - b.E), s * (a.by - b.by), s * (a.bz - b.bz)};
	};
	u_dstar_L = jump(spds[1], u_dstar_L, u_star_L);
	u_star_L = jump(spds[0], u_star_L, u_L);
	u_dstar_R = jump(spds[3], u_dstar_R, u_star_R);
	u_star_R = jump(spds[4], u_star_R, u_R);
	// :273-329 the flux at the interface
	auto add2 = [](ConsHydro1D const &a, ConsHydro1D const &b) {
		return ConsHydro1D{a.rho + b.rho, a.mx + b.mx, a.my + b.my, a.mz + b.mz, a.E + b.E, a.by + b.by, a.bz + b.bz};
	};
	auto add3 = [](ConsHydro1D const &a, ConsHydro1D const &b, ConsHydro1D const &c) {
		return ConsHydro1D{a.rho + b.rho + c.rho, a.mx + b.mx + c.mx, a.my + b.my + c.my, a.mz + b.mz + c.mz, a.E + b.E + c.E, a.by + b.by + c.by, a.bz + b.bz + c.bz};
	};
	if (spds[0] >= 0.0) {
		f_x = f_L;
	} else if (spds[4] <= 0.0) {
		f_x = f_R;
	} else if (spds[1] >= 0.0) {
		f_x = add2(f_L, u_star_L);
	} else if (spds[2] >= 0.0) {
		f_x = add3(f_L, u_star_L, u_dstar_L);
	} else if (spds[3] > 0.0) {
		f_x = add3(f_R, u_star_R, u_dstar_R);
	} else {
		f_x = add2(f_R, u_star_R);
	}
	valarray F_hydro{};
	F_hydro[0] = f_x.rho;
	F_hydro[1] = f_x.mx;
	F_hydro[2] = f_x.my;
	F_hydro[3] = f_x.mz;
	F_hydro[4] = f_x.E;
	F_hydro[5] = 0.0;
	return F_hydro;
}

// LLF.hpp:16-43
inline auto LLF(HydroTraits const &tr, HydroState const &sL, HydroState const &sR) -> valarray
{
	const int fluxdim = tr.nvar();
	const int N_scalars = tr.nscalars;
	// :21
	const double Sp = std::max(std::abs(sL.u) + sL.cs, std::abs(sR.u) + sR.cs);

	valarray D_L{}, D_R{}, U_L{}, U_R{};
	U_L[0] = sL.rho;
	U_L[1] = sL.rho * sL.u;
	U_L[2] = sL.rho * sL.v;
	U_L[3] = sL.rho * sL.w;
	U_L[4] = sL.E;
	U_L[5] = sL.Eint;
	U_R[0] = sR.rho;
	U_R[1] = sR.rho * sR.u;
	U_R[2] = sR.rho * sR.v;
	U_R[3] = sR.rho * sR.w;
	U_R[4] = sR.E;
	U_R[5] = sR.Eint;
	for (int n = 0; n < N_scalars; ++n) {
		const int nstart = fluxdim - N_scalars;
		U_L[nstart + n] = sL.scalar[n];
		U_R[nstart + n] = sR.scalar[n];
	}
	D_L[1] = 1.;
	D_L[4] = sL.u;
	D_R[1] = 1.;
	D_R[4] = sR.u;

	valarray F{};
	for (int n = 0; n < fluxdim; ++n) {
		const double F_L = sL.u * U_L[n] + sL.P * D_L[n];
		const double F_R = sR.u * U_R[n] + sR.P * D_R[n];
		// :41  `0.5 * Sp * (U_R - U_L)` evaluates (0.5 * Sp) first
		F[n] = 0.5 * (F_L + F_R) - (0.5 * Sp) * (U_R[n] - U_L[n]);
	}
	return F;
}

struct HydroSystem {
	HydroTraits tr;

	// hydro_system.hpp:349-372
	[[nodiscard]] auto ComputePressure(Array4<const double> const &cons, int i, int j, int k) const -> double
	{
		const auto rho = cons(i, j, k, density_index);
		const auto px = cons(i, j, k, x1Momentum_index);
		const auto py = cons(i, j, k, x2Momentum_index);
		const auto pz = cons(i, j, k, x3Momentum_index);
		const auto E = cons(i, j, k, energy_index);
		const auto vx = px / rho;
		const auto vy = py / rho;
		const auto vz = pz / rho;
		const auto kinetic_energy = 0.5 * rho * (vx * vx + vy * vy + vz * vz);
		const auto thermal_energy = E - kinetic_energy;
		double P = NAN;
		if (tr.is_eos_isothermal()) {
			P = rho * tr.cs_iso() * tr.cs_iso();
		} else {
			P = tr.eos.ComputePressure(rho, thermal_energy);
		}
		return P;
	}

	// hydro_system.hpp:374-394
	[[nodiscard]] auto ComputeSoundSpeed(Array4<const double> const &cons, int i, int j, int k) const -> double
	{
		const auto rho = cons(i, j, k, density_index);
		const auto px = cons(i, j, k, x1Momentum_index);
		const auto py = cons(i, j, k, x2Momentum_index);
		const auto pz = cons(i, j, k, x3Momentum_index);
		const auto E = cons(i, j, k, energy_index);
		const auto vx = px / rho;
		const auto vy = py / rho;
		const auto vz = pz / rho;
		const auto kinetic_energy = 0.5 * rho * (vx * vx + vy * vy + vz * vz);
		const auto thermal_energy = E - kinetic_energy;
		const double P = tr.eos.ComputePressure(rho, thermal_energy);
		const double cs = tr.eos.ComputeSoundSpeed(rho, P);
		return cs;
	}

	// hydro_system.hpp:396-421
	[[nodiscard]] static auto ComputeVelocity(Array4<const double> const &cons, int i, int j, int k, int dir) -> double
	{
		double const rho = cons(i, j, k, density_index);
		return cons(i, j, k, x1Momentum_index + dir) / rho;
	}

	// hydro_system.hpp:423-446 (nmscalars = 0)
	[[nodiscard]] auto isStateValid(Array4<const double> const &cons, int i, int j, int k) const -> bool
	{
		const double rho = cons(i, j, k, density_index);
		bool const isDensityPositive = (rho > 0.);
		bool isMassScalarPositive = true; // :430-441
		for (int idx = 0; idx < tr.nmscalars; ++idx) {
			if (cons(i, j, k, scalar0_index + idx) < 0.0) {
				isMassScalarPositive = false;
				break;
			}
		}
		return isDensityPositive && isMassScalarPositive;
	}

	// hydro_system.hpp:138-196; launched over valid + nghost
	void ConservedToPrimitive(Array4<const double> const &cons, Array4<double> const &primVar, Box const &range) const
	{
		for (int k = range.lo[2]; k <= range.hi[2]; ++k) {
			for (int j = range.lo[1]; j <= range.hi[1]; ++j) {
				for (int i = range.lo[0]; i <= range.hi[0]; ++i) {
					const auto rho = cons(i, j, k, density_index);
					const auto px = cons(i, j, k, x1Momentum_index);
					const auto py = cons(i, j, k, x2Momentum_index);
					const auto pz = cons(i, j, k, x3Momentum_index);
					const auto E = cons(i, j, k, energy_index);
					const auto Eint_aux = cons(i, j, k, internalEnergy_index);

					const auto vx = px / rho;
					const auto vy = py / rho;
					const auto vz = pz / rho;
					const auto kinetic_energy = 0.5 * rho * (vx * vx + vy * vy + vz * vz);
					const auto Eint_cons = E - kinetic_energy;

					const double Pgas = ComputePressure(cons, i, j, k);
					const double eint_cons = Eint_cons / rho;
					const double eint_aux = Eint_aux / rho;

					primVar(i, j, k, primDensity_index) = rho;
					primVar(i, j, k, x1Velocity_index) = vx;
					primVar(i, j, k, x2Velocity_index) = vy;
					primVar(i, j, k, x3Velocity_index) = vz;

					if (tr.reconstruct_eint) {
						primVar(i, j, k, pressure_index) = eint_cons;
						primVar(i, j, k, primEint_index) = eint_aux;
					} else {
						primVar(i, j, k, pressure_index) = Pgas;
						primVar(i, j, k, primEint_index) = Eint_aux;
					}
					for (int nc = 0; nc < tr.nscalars; ++nc) {
						primVar(i, j, k, primScalar0_index + nc) = cons(i, j, k, scalar0_index + nc);
					}
				}
			}
		}
	}

	// hydro_system.hpp:223-252
	void ComputeMaxSignalSpeed(Array4<const double> const &cons, Array4<double> const &maxSignal, Box const &range) const
	{
		for (int k = range.lo[2]; k <= range.hi[2]; ++k) {
			for (int j = range.lo[1]; j <= range.hi[1]; ++j) {
				for (int i = range.lo[0]; i <= range.hi[0]; ++i) {
					const auto rho = cons(i, j, k, density_index);
					const auto px = cons(i, j, k, x1Momentum_index);
					const auto py = cons(i, j, k, x2Momentum_index);
					const auto pz = cons(i, j, k, x3Momentum_index);
					const auto vx = px / rho;
					const auto vy = py / rho;
					const auto vz = pz / rho;
					const double vel_mag = std::sqrt(vx * vx + vy * vy + vz * vz);
					double cs = NAN;
					if (tr.is_eos_isothermal()) {
						cs = tr.cs_iso();
					} else {
						cs = ComputeSoundSpeed(cons, i, j, k);
					}
					const double signal_max = cs + vel_mag;
					maxSignal(i, j, k) = signal_max;
				}
			}
		}
	}

	// hydro_system.hpp:198-221 (ParReduce max over valid cells of one box)
	[[nodiscard]] auto maxSignalSpeedLocal(Array4<const double> const &cons, Box const &range) const -> double
	{
		double result = -std::numeric_limits<double>::infinity();
		for (int k = range.lo[2]; k <= range.hi[2]; ++k) {
			for (int j = range.lo[1]; j <= range.hi[1]; ++j) {
				for (int i = range.lo[0]; i <= range.hi[0]; ++i) {
					const auto rho = cons(i, j, k, density_index);
					const auto px = cons(i, j, k, x1Momentum_index);
					const auto py = cons(i, j, k, x2Momentum_index);
					const auto pz = cons(i, j, k, x3Momentum_index);
					const auto kinetic_energy = (px * px + py * py + pz * pz) / (2.0 * rho);
					const double abs_vel = std::sqrt(2.0 * kinetic_energy / rho);
					double cs = NAN;
					if (tr.is_eos_isothermal()) {
						cs = tr.cs_iso();
					} else {
						cs = ComputeSoundSpeed(cons, i, j, k);
					}
					result = std::max(result, cs + abs_vel);
				}
			}
		}
		return result;
	}

	// hydro_system.hpp:531-626; launched over valid + nghost(=2)
	void ComputeFlatteningCoefficients(int dir, Array4<const double> const &primVar_in, Array4<double> const &x1Chi_in, Box const &range) const
	{
		constexpr double beta_max = 0.85;
		constexpr double beta_min = 0.75;
		constexpr double Zmax = 0.75;
		constexpr double Zmin = 0.25;

		View<const double> primVar(primVar_in, dir);
		View<double> x1Chi(x1Chi_in, dir);
		for (int k_in = range.lo[2]; k_in <= range.hi[2]; ++k_in) {
			for (int j_in = range.lo[1]; j_in <= range.hi[1]; ++j_in) {
				for (int i_in = range.lo[0]; i_in <= range.hi[0]; ++i_in) {
					auto [i, j, k] = reorderMultiIndex(dir, i_in, j_in, k_in);

					double Pplus2 = primVar(i + 2, j, k, pressure_index);
					double Pplus1 = primVar(i + 1, j, k, pressure_index);
					double P = primVar(i, j, k, pressure_index);
					double Pminus1 = primVar(i - 1, j, k, pressure_index);
					double Pminus2 = primVar(i - 2, j, k, pressure_index);

					if (tr.reconstruct_eint) { // :561-577
						Pplus2 = tr.eos.ComputePressure(primVar(i + 2, j, k, primDensity_index), primVar(i + 2, j, k, primDensity_index) * Pplus2);
						Pplus1 = tr.eos.ComputePressure(primVar(i + 1, j, k, primDensity_index), primVar(i + 1, j, k, primDensity_index) * Pplus1);
						P = tr.eos.ComputePressure(primVar(i, j, k, primDensity_index), primVar(i, j, k, primDensity_index) * P);
						Pminus1 =
						    tr.eos.ComputePressure(primVar(i - 1, j, k, primDensity_index), primVar(i - 1, j, k, primDensity_index) * Pminus1);
						Pminus2 =
						    tr.eos.ComputePressure(primVar(i - 2, j, k, primDensity_index), primVar(i - 2, j, k, primDensity_index) * Pminus2);
					}

					if (tr.is_eos_isothermal()) { // :579-586
						const double cs_sq = tr.cs_iso() * tr.cs_iso();
						Pplus2 = primVar(i + 2, j, k, primDensity_index) * cs_sq;
						Pplus1 = primVar(i + 1, j, k, primDensity_index) * cs_sq;
						P = primVar(i, j, k, primDensity_index) * cs_sq;
						Pminus1 = primVar(i - 1, j, k, primDensity_index) * cs_sq;
						Pminus2 = primVar(i - 2, j, k, primDensity_index) * cs_sq;
					}

					// :593-598
					const double beta_denom = std::abs(Pplus2 - Pminus2);
					const double beta = (beta_denom != 0) ? (std::abs(Pplus1 - Pminus1) / beta_denom) : 0;
					const double chi_min = std::max(0., std::min(1., (beta_max - beta) / (beta_max - beta_min)));

					// :601-606 (std::pow(cs, 2) == cs*cs)
					const double cs = tr.eos.ComputeSoundSpeed(primVar(i, j, k, primDensity_index), P);
					double K_S = (cs * cs) * primVar(i, j, k, primDensity_index);
					if (tr.is_eos_isothermal()) {
						K_S = primVar(i, j, k, primDensity_index) * tr.cs_iso() * tr.cs_iso();
					}

					const double Z = std::abs(Pplus1 - Pminus1) / K_S;

					// :611-622
					const int velocity_index = x1Velocity_index + dir;
					double chi = 1.0;
					if (primVar(i + 1, j, k, velocity_index) < primVar(i - 1, j, k, velocity_index)) {
						chi = std::max(chi_min, std::min(1., (Zmax - Z) / (Zmax - Zmin)));
					}
					x1Chi(i, j, k) = chi;
				}
			}
		}
	}

	// hydro_system.hpp:628-694; launched over valid + nghost(=1) x nvars
	void FlattenShocks(int dir, Array4<const double> const &q_in, Array4<const double> const &x1Chi_in, Array4<const double> const &x2Chi_in,
			   Array4<const double> const &x3Chi_in, Array4<double> const &x1LeftState_in, Array4<double> const &x1RightState_in,
			   Box const &range, int nvars) const
	{
		View<const double> q(q_in, dir);
		View<double> x1LeftState(x1LeftState_in, dir);
		View<double> x1RightState(x1RightState_in, dir);
		for (int n = 0; n < nvars; ++n) {
			for (int k_in = range.lo[2]; k_in <= range.hi[2]; ++k_in) {
				for (int j_in = range.lo[1]; j_in <= range.hi[1]; ++j_in) {
					for (int i_in = range.lo[0]; i_in <= range.hi[0]; ++i_in) {
						// :655-669
						double chi_ijk = std::min({x1Chi_in(i_in - 1, j_in, k_in), x1Chi_in(i_in, j_in, k_in), x1Chi_in(i_in + 1, j_in, k_in)});
						if (tr.ndim >= 2) {
							chi_ijk =
							    std::min({chi_ijk, x2Chi_in(i_in, j_in - 1, k_in), x2Chi_in(i_in, j_in, k_in), x2Chi_in(i_in, j_in + 1, k_in)});
						}
						if (tr.ndim == 3) {
							chi_ijk =
							    std::min({chi_ijk, x3Chi_in(i_in, j_in, k_in - 1), x3Chi_in(i_in, j_in, k_in), x3Chi_in(i_in, j_in, k_in + 1)});
						}

						auto [i, j, k] = reorderMultiIndex(dir, i_in, j_in, k_in);

						// :674-685
						const double a_minus = x1RightState(i, j, k, n);
						const double a_plus = x1LeftState(i + 1, j, k, n);
						const double a_mean = q(i, j, k, n);
						const double new_a_minus = chi_ijk * a_minus + (1. - chi_ijk) * a_mean;
						const double new_a_plus = chi_ijk * a_plus + (1. - chi_ijk) * a_mean;
						x1RightState(i, j, k, n) = new_a_minus;
						x1LeftState(i + 1, j, k, n) = new_a_plus;
					}
				}
			}
		}
	}

	// hydro_system.hpp:852-1112; launched over the face box (no ghosts)
	void ComputeFluxes(int riemann, int dir, Array4<double> const &x1Flux_in, Array4<double> const &x1FaceVel_in,
			   Array4<const double> const &x1LeftState_in, Array4<const double> const &x1RightState_in,
			   Array4<const double> const &primVar_in, const double K_visc, Box const &faceRange) const
	{
		View<const double> x1LeftState(x1LeftState_in, dir);
		View<const double> x1RightState(x1RightState_in, dir);
		View<double> x1Flux(x1Flux_in, dir);
		View<double> x1FaceVel(x1FaceVel_in, dir);
		View<const double> q(primVar_in, dir);
		const int nvar_ = tr.nvar();
		const int nscalars_ = tr.nscalars;
		const double gamma_ = tr.gamma();
		const double cs_iso_ = tr.cs_iso();

		for (int k_in = faceRange.lo[2]; k_in <= faceRange.hi[2]; ++k_in) {
			for (int j_in = faceRange.lo[1]; j_in <= faceRange.hi[1]; ++j_in) {
				for (int i_in = faceRange.lo[0]; i_in <= faceRange.hi[0]; ++i_in) {
					auto [i, j, k] = reorderMultiIndex(dir, i_in, j_in, k_in);

					// :881-894
					const double rho_L = x1LeftState(i, j, k, primDensity_index);
					const double rho_R = x1RightState(i, j, k, primDensity_index);
					const double vx_L = x1LeftState(i, j, k, x1Velocity_index);
					const double vx_R = x1RightState(i, j, k, x1Velocity_index);
					const double vy_L = x1LeftState(i, j, k, x2Velocity_index);
					const double vy_R = x1RightState(i, j, k, x2Velocity_index);
					const double vz_L = x1LeftState(i, j, k, x3Velocity_index);
					const double vz_R = x1RightState(i, j, k, x3Velocity_index);
					const double ke_L = 0.5 * rho_L * (vx_L * vx_L + vy_L * vy_L + vz_L * vz_L);
					const double ke_R = 0.5 * rho_R * (vx_R * vx_R + vy_R * vy_R + vz_R * vz_R);

					double Eint_L = NAN, Eint_R = NAN, P_L = NAN, P_R = NAN, E_L = NAN, E_R = NAN, cs_L = NAN, cs_R = NAN;

					if (tr.is_eos_isothermal()) { // :910-915
						P_L = rho_L * (cs_iso_ * cs_iso_);
						P_R = rho_R * (cs_iso_ * cs_iso_);
						cs_L = cs_iso_;
						cs_R = cs_iso_;
					} else {
						if (tr.reconstruct_eint) { // :917-929
							const double eint_L = x1LeftState(i, j, k, pressure_index);
							const double eint_R = x1RightState(i, j, k, pressure_index);
							P_L = tr.eos.ComputePressure(rho_L, eint_L * rho_L);
							P_R = tr.eos.ComputePressure(rho_R, eint_R * rho_R);
							Eint_L = rho_L * x1LeftState(i, j, k, primEint_index);
							Eint_R = rho_R * x1RightState(i, j, k, primEint_index);
						} else { // :930-938
							P_L = x1LeftState(i, j, k, pressure_index);
							P_R = x1RightState(i, j, k, pressure_index);
							Eint_L = x1LeftState(i, j, k, primEint_index);
							Eint_R = x1RightState(i, j, k, primEint_index);
						}
						// :940-946
						cs_L = tr.eos.ComputeSoundSpeed(rho_L, P_L);
						E_L = tr.eos.ComputeEintFromPres(rho_L, P_L) + ke_L;
						cs_R = tr.eos.ComputeSoundSpeed(rho_R, P_R);
						E_R = tr.eos.ComputeEintFromPres(rho_R, P_R) + ke_R;
					}

					// :954-976 (in 1-D only X1 occurs; X2: the swap of a 2-D build :963-966, the cyclic permutation of a 3-D build :967-970)
					int velN_index = x1Velocity_index;
					int velV_index = x2Velocity_index;
					int velW_index = x3Velocity_index;
					if (dir == X2) {
						if (g_spacedim == 2) {
							velN_index = x2Velocity_index;
							velV_index = x1Velocity_index;
							velW_index = x3Velocity_index; // unchanged in 2D
						} else {
							velN_index = x2Velocity_index;
							velV_index = x3Velocity_index;
							velW_index = x1Velocity_index;
						}
					} else if (dir == X3) {
						velN_index = x3Velocity_index;
						velV_index = x1Velocity_index;
						velW_index = x2Velocity_index;
					}

					// :978-1003
					HydroState sL{};
					sL.rho = rho_L;
					sL.u = x1LeftState(i, j, k, velN_index);
					sL.v = x1LeftState(i, j, k, velV_index);
					sL.w = x1LeftState(i, j, k, velW_index);
					sL.P = P_L;
					sL.cs = cs_L;
					sL.E = E_L;
					sL.Eint = Eint_L;
					sL.by = 0.0;
					sL.bz = 0.0;
					HydroState sR{};
					sR.rho = rho_R;
					sR.u = x1RightState(i, j, k, velN_index);
					sR.v = x1RightState(i, j, k, velV_index);
					sR.w = x1RightState(i, j, k, velW_index);
					sR.P = P_R;
					sR.cs = cs_R;
					sR.E = E_R;
					sR.Eint = Eint_R;
					sR.by = 0.0;
					sR.bz = 0.0;

					// :1008-1016
					for (int n = 0; n < nscalars_; ++n) {
						sL.scalar[n] = x1LeftState(i, j, k, scalar0_index + n);
						sR.scalar[n] = x1RightState(i, j, k, scalar0_index + n);
					}

					// :1019
					const double du = q(i, j, k, velN_index) - q(i - 1, j, k, velN_index);

					// :1022-1034
					double dw = 0.;
					double dvl = 0., dvr = 0., dwl = 0., dwr = 0.;
					if (tr.ndim >= 2) {
						dvl = std::min(q(i - 1, j + 1, k, velV_index) - q(i - 1, j, k, velV_index),
							       q(i - 1, j, k, velV_index) - q(i - 1, j - 1, k, velV_index));
						dvr = std::min(q(i, j + 1, k, velV_index) - q(i, j, k, velV_index), q(i, j, k, velV_index) - q(i, j - 1, k, velV_index));
						dw = std::min(dvl, dvr);
					}
					if (tr.ndim == 3) {
						dwl = std::min(q(i - 1, j, k + 1, velW_index) - q(i - 1, j, k, velW_index),
							       q(i - 1, j, k, velW_index) - q(i - 1, j, k - 1, velW_index));
						dwr = std::min(q(i, j, k + 1, velW_index) - q(i, j, k, velW_index), q(i, j, k, velW_index) - q(i, j, k - 1, velW_index));
						dw = std::min(std::min(dwl, dwr), dw);
					}

					// :1037-1048
					valarray F_canonical{};
					if (riemann == riemann_HLLC) {
						F_canonical = HLLC(tr, sL, sR, gamma_, du, dw);
					} else if (riemann == riemann_LLF) {
						F_canonical = LLF(tr, sL, sR);
					} else { // :1044-1048: bx = 0 "for testing purposes" (the reference's MHD stub; sL / sR carry by = bz = 0)
						F_canonical = HLLD(sL, sR, gamma_, 0.0);
					}
					valarray F = F_canonical;

					// :1054-1055  AMREX_D_TERM(du, +0.5*(dvl+dvr), +0.5*(dwl+dwr))
					double div_v = du;
					if (tr.ndim >= 2) {
						div_v = div_v + 0.5 * (dvl + dvr);
					}
					if (tr.ndim == 3) {
						div_v = div_v + 0.5 * (dwl + dwr);
					}
					const double viscosity = K_visc * std::max(-div_v, 0.);

					// :1057-1074
					valarray U_L{}, U_R{};
					U_L[0] = sL.rho;
					U_L[1] = sL.rho * sL.u;
					U_L[2] = sL.rho * sL.v;
					U_L[3] = sL.rho * sL.w;
					U_L[4] = sL.E;
					U_L[5] = sL.Eint;
					U_R[0] = sR.rho;
					U_R[1] = sR.rho * sR.u;
					U_R[2] = sR.rho * sR.v;
					U_R[3] = sR.rho * sR.w;
					U_R[4] = sR.E;
					U_R[5] = sR.Eint;
					for (int n = 0; n < nscalars_; ++n) {
						const int nstart = nvar_ - nscalars_;
						U_L[nstart + n] = sL.scalar[n];
						U_R[nstart + n] = sR.scalar[n];
					}

					// :1076
					for (int n = 0; n < nvar_; ++n) {
						F[n] = F[n] + viscosity * (U_L[n] - U_R[n]);
					}

					// :1079-1081 (momentum components take the canonical flux WITHOUT viscosity)
					F[velN_index] = F_canonical[x1Momentum_index];
					F[velV_index] = F_canonical[x2Momentum_index];
					F[velW_index] = F_canonical[x3Momentum_index];

					// :1084-1087
					if (tr.is_eos_isothermal()) {
						F[energy_index] = 0;
						F[internalEnergy_index] = 0;
					}

					// :1090-1091
					const double v_norm = (F[density_index] >= 0.) ? (F[density_index] / rho_R) : (F[density_index] / rho_L);
					x1FaceVel(i, j, k) = v_norm;

					// :1094-1104 consistent multi-fluid advection (Plewa & Mueller 1999): the partial-density fluxes are the mass
					// flux split in the proportions of the upwind state, so that they sum to it
					if (tr.nmscalars > 0) {
						double fluxSum_U_L = 0, fluxSum_U_R = 0; // :1062-1073
						const int nstart = nvar_ - nscalars_;
						for (int n = 0; n < tr.nmscalars; ++n) {
							fluxSum_U_L += U_L[nstart + n];
							fluxSum_U_R += U_R[nstart + n];
						}
						if (F[density_index] >= 0.) {
							for (int n = 0; n < tr.nmscalars; ++n) {
								F[nstart + n] = F[density_index] * U_L[nstart + n] / fluxSum_U_L;
							}
						} else {
							for (int n = 0; n < tr.nmscalars; ++n) {
								F[nstart + n] = F[density_index] * U_R[nstart + n] / fluxSum_U_R;
							}
						}
					}

					// :1107-1110
					for (int nc = 0; nc < nvar_; ++nc) {
						x1Flux(i, j, k, nc) = F[nc];
					}
				}
			}
		}
	}

	// hydro_system.hpp:448-473
	void ComputeRhsFromFluxes(Array4<double> const &rhs, std::array<Array4<const double>, 3> const &flux, double const dx[3], int nvars,
				  Box const &range) const
	{
		for (int n = 0; n < nvars; ++n) {
			for (int k = range.lo[2]; k <= range.hi[2]; ++k) {
				for (int j = range.lo[1]; j <= range.hi[1]; ++j) {
					for (int i = range.lo[0]; i <= range.hi[0]; ++i) {
						double r = (1.0 / dx[0]) * (flux[0](i, j, k, n) - flux[0](i + 1, j, k, n));
						if (tr.ndim >= 2) {
							r = r + (1.0 / dx[1]) * (flux[1](i, j, k, n) - flux[1](i, j + 1, k, n));
						}
						if (tr.ndim == 3) {
							r = r + (1.0 / dx[2]) * (flux[2](i, j, k, n) - flux[2](i, j, k + 1, n));
						}
						rhs(i, j, k, n) = r;
					}
				}
			}
		}
	}

	// hydro_system.hpp:775-814
	void AddInternalEnergyPdV(Array4<double> const &rhs, Array4<const double> const &consVar, double const dx[3],
				  std::array<Array4<const double>, 3> const &vel, Array4<const int> const &redoFlag, Box const &range) const
	{
		for (int k = range.lo[2]; k <= range.hi[2]; ++k) {
			for (int j = range.lo[1]; j <= range.hi[1]; ++j) {
				for (int i = range.lo[0]; i <= range.hi[0]; ++i) {
					const double Pgas = ComputePressure(consVar, i, j, k);
					double div_v = NAN;
					if (redoFlag(i, j, k) == redo_none) {
						div_v = (vel[0](i + 1, j, k) - vel[0](i, j, k)) / dx[0];
						if (tr.ndim >= 2) {
							div_v = div_v + (vel[1](i, j + 1, k) - vel[1](i, j, k)) / dx[1];
						}
						if (tr.ndim == 3) {
							div_v = div_v + (vel[2](i, j, k + 1) - vel[2](i, j, k)) / dx[2];
						}
					} else {
						double s = (ComputeVelocity(consVar, i + 1, j, k, 0) - ComputeVelocity(consVar, i - 1, j, k, 0)) / dx[0];
						if (tr.ndim >= 2) {
							s = s + (ComputeVelocity(consVar, i, j + 1, k, 1) - ComputeVelocity(consVar, i, j - 1, k, 1)) / dx[1];
						}
						if (tr.ndim == 3) {
							s = s + (ComputeVelocity(consVar, i, j, k + 1, 2) - ComputeVelocity(consVar, i, j, k - 1, 2)) / dx[2];
						}
						div_v = 0.5 * s;
					}
					rhs(i, j, k, internalEnergy_index) += -Pgas * div_v;
				}
			}
		}
	}

	// hydro_system.hpp:475-497
	void PredictStep(Array4<const double> const &consVarOld, Array4<double> const &consVarNew, Array4<const double> const &rhs, const double dt,
			 const int nvars, Array4<int> const &redoFlag, Box const &range) const
	{
		for (int k = range.lo[2]; k <= range.hi[2]; ++k) {
			for (int j = range.lo[1]; j <= range.hi[1]; ++j) {
				for (int i = range.lo[0]; i <= range.hi[0]; ++i) {
					for (int n = 0; n < nvars; ++n) {
						consVarNew(i, j, k, n) = consVarOld(i, j, k, n) + dt * rhs(i, j, k, n);
					}
					Array4<const double> cn(consVarNew.p, consVarNew.box(), consVarNew.ncomp);
					if (!isStateValid(cn, i, j, k)) {
						redoFlag(i, j, k) = redo_redo;
					} else {
						redoFlag(i, j, k) = redo_none;
					}
				}
			}
		}
	}

	// hydro_system.hpp:696-773
	void EnforceLimits(double const densityFloor, double const tempFloor, Array4<double> const &state, Box const &range) const
	{
		for (int k = range.lo[2]; k <= range.hi[2]; ++k) {
			for (int j = range.lo[1]; j <= range.hi[1]; ++j) {
				for (int i = range.lo[0]; i <= range.hi[0]; ++i) {
					double rho_new = NAN;
					{
						double const rho = state(i, j, k, density_index);
						rho_new = rho;
						if (rho < densityFloor) {
							rho_new = densityFloor;
							state(i, j, k, density_index) = rho_new;
							for (int n = 0; n < tr.nscalars; ++n) {
								if (rho_new == 0.0) {
									state(i, j, k, scalar0_index + n) = 0.0;
								} else {
									state(i, j, k, scalar0_index + n) *= rho / rho_new;
								}
							}
						}
					}
					// :725-744 mass-scalar floor and renormalisation (network_rp::small_x of the un-vendored Microphysics: its default 1e-30)
					if (tr.nmscalars > 0) {
						double sp_sum = 0.0;
						for (int idx = 0; idx < tr.nmscalars; ++idx) {
							if (state(i, j, k, scalar0_index + idx) < 0.0) {
								state(i, j, k, scalar0_index + idx) = 1.0e-30 * rho_new;
							}
							sp_sum += state(i, j, k, scalar0_index + idx);
						}
						if ((sp_sum > std::numeric_limits<double>::min()) && (rho_new > std::numeric_limits<double>::min())) {
							sp_sum /= rho_new;
							for (int idx = 0; idx < tr.nmscalars; ++idx) {
								state(i, j, k, scalar0_index + idx) /= sp_sum;
							}
						}
					}
					if ((rho_new > std::numeric_limits<double>::min()) && !tr.is_eos_isothermal()) {
						double const vx1 = state(i, j, k, x1Momentum_index) / rho_new;
						double const vx2 = state(i, j, k, x2Momentum_index) / rho_new;
						double const vx3 = state(i, j, k, x3Momentum_index) / rho_new;
						double const Ekin = 0.5 * rho_new * (vx1 * vx1 + vx2 * vx2 + vx3 * vx3);

						double const Etot = state(i, j, k, energy_index);
						double const primTemp = tr.eos.ComputeTgasFromEint(rho_new, (Etot - Ekin));
						if (primTemp < tempFloor) {
							double const prim_eint = tr.eos.ComputeEintFromTgas(rho_new, tempFloor);
							state(i, j, k, energy_index) = Ekin + prim_eint;
						}
						double const auxEint = state(i, j, k, internalEnergy_index);
						double const auxTemp = tr.eos.ComputeTgasFromEint(rho_new, auxEint);
						if (auxTemp < tempFloor) {
							double const new_Eint = tr.eos.ComputeEintFromTgas(rho_new, tempFloor);
							state(i, j, k, internalEnergy_index) = new_Eint;
						}
					}
				}
			}
		}
	}

	// hydro_system.hpp:816-850
	void SyncDualEnergy(Array4<double> const &consVar, Box const &range) const
	{
		const double eta = 1.0e-3;
		for (int k = range.lo[2]; k <= range.hi[2]; ++k) {
			for (int j = range.lo[1]; j <= range.hi[1]; ++j) {
				for (int i = range.lo[0]; i <= range.hi[0]; ++i) {
					double const rho = consVar(i, j, k, density_index);
					double const px = consVar(i, j, k, x1Momentum_index);
					double const py = consVar(i, j, k, x2Momentum_index);
					double const pz = consVar(i, j, k, x3Momentum_index);
					double const Etot = consVar(i, j, k, energy_index);
					double const Eint_aux = consVar(i, j, k, internalEnergy_index);
					if (rho <= 0.) {
						std::fprintf(stderr, "density is negative in SyncDualEnergy! abort!!\n");
						std::abort();
					}
					double const Ekin = (px * px + py * py + pz * pz) / (2.0 * rho);
					double const Eint_cons = Etot - Ekin;
					if (Eint_cons > eta * Etot) {
						consVar(i, j, k, internalEnergy_index) = Eint_cons;
					} else {
						consVar(i, j, k, internalEnergy_index) = Eint_aux;
						consVar(i, j, k, energy_index) = Eint_aux + Ekin;
					}
				}
			}
		}
	}
};

} // namespace oracle

#endif // ORACLE_HYDRO_HPP_
