// ORACLE — TEST INFRASTRUCTURE ONLY (see grid.hpp header).
//
// hydro_fused.hpp: the hydro flux evaluation of one RK stage (HydroSimulation::computeHydroFluxes — ConservedToPrimitive,
// ComputeFlatteningCoefficients x3, ReconstructStatesPPM + FlattenShocks + ComputeFluxes<HLLC> x3; reference src/QuokkaSimulation.hpp:1403-1517,
// src/hydro/hydro_system.hpp:138-196, :531-694, :852-1112, src/hydro/HLLC.hpp:22-153, src/hyperbolic_system.hpp:337-433) restated ONCE MORE as a
// fused, cache-blocked, x-vectorised loop nest: what the reference's "MPI + vectorised CPU path" does with AMReX's tiling and the compiler's
// vectoriser (scripts/cpu-sedov-1node.pbs:1-20), and the CPU leg bench.py times beside the GPU (`cpu_baseline`).
//
// Same arithmetic, operation by operation, as the operator-at-a-time functions of hydro.hpp / hyperbolic.hpp / eos.hpp (built with
// -ffp-contract=off, no reassociation): tests/test_oracle_fused_cpu.py holds the two forms equal in EVERY BIT on developed and random states and
// over whole Sedov runs.  What differs is the traversal: per box the primitives and the three flattening coefficients are formed once into
// box-sized scratch (a 32^3 box with its ghost cells: 64000 doubles per array — L2-resident), then every direction is swept row by row with x
// innermost: the two edge states of a row of cells live in row buffers (never as face MultiFabs), the previous row's right edges are kept for the
// faces between two rows, and the Riemann solve runs over a row of faces with selects instead of branches (every branch of HLLC.hpp is evaluated
// and the result chosen: the chosen value has the bits the branch would have produced).  EOS calls are the closed forms of eos.hpp's direct
// gamma-law variant (ORACLE_EOS_VARIANT == 0) — p = (gamma - 1) rho e etc. — instead of the generic eos() that fills every field of eos_state.
//
// Covered (what BASELINE configs 2 / 3 / 5 run): 3-D, gamma law (gamma != 1), no passive scalars, PPM, HLLC, reconstruct_eint = false.
// Anything else: fusedHydroFluxesApplicable() is false and the caller takes the operator path.
#ifndef ORACLE_HYDRO_FUSED_HPP_
#define ORACLE_HYDRO_FUSED_HPP_

#include <array>
#include <cmath>
#include <vector>

#include "hydro.hpp"
#include "hyperbolic.hpp"

// The loops below are compiled in their own translation unit (hydro_fused.cpp, ORACLE_FUSED_IMPL) with -fno-math-errno -fno-trapping-math on top of
// the oracle's flags: sqrt becomes the vector instruction (not a libm call that may set errno) and a conditional select may evaluate both arms —
// what the vectoriser needs.  Neither changes a result: no reassociation, no contraction, the same correctly rounded operations per lane.
namespace oracle
{
namespace fused
{

// std::min / std::max as the standard defines them for two arguments (first argument on ties and for unordered operands)
inline auto mn(double a, double b) -> double { return (b < a) ? b : a; }
inline auto mx(double a, double b) -> double { return (a < b) ? b : a; }
inline auto sgnd(double v) -> double { return static_cast<double>(static_cast<int>(0.0 < v) - static_cast<int>(v < 0.0)); }

struct Work {
	std::vector<double> prim;     // 6 x ghost-4 box
	std::vector<double> chi[3];   // per direction, ghost-4 geometry (filled on valid + 2)
	std::vector<double> chim;     // combined coefficient of FlattenShocks (filled on valid + 1)
	std::vector<double> am, ap, apPrev; // edge states of one row of cells: 6 x row
	void resize(size_t ncell, size_t row)
	{
		prim.resize(6 * ncell);
		for (auto &c : chi) {
			c.resize(ncell);
		}
		chim.resize(ncell);
		am.resize(6 * row);
		ap.resize(6 * row);
		apPrev.resize(6 * row);
	}
};

#ifdef ORACLE_FUSED_IMPL
// PPM edge states of a row of `n` cells for variable arrays q (element e of the row is q[e * 1], its neighbours along the sweep axis are sd apart),
// flattened with the row's combined coefficient: hyperbolic_system.hpp:337-433 + hydro_system.hpp:674-685
__attribute__((always_inline)) inline void edgesRow(const double *__restrict q, int64_t sd, const double *__restrict chi, int n, double *__restrict am_out, double *__restrict ap_out)
{
	constexpr double coef_1 = (7. / 12.);
	constexpr double coef_2 = (-1. / 12.);
#pragma omp simd
	for (int e = 0; e < n; ++e) {
		const double q0 = q[e], qm1 = q[e - sd], qm2 = q[e - 2 * sd], qp1 = q[e + sd], qp2 = q[e + 2 * sd];
		// std::minmax({q0, qm1, qp1})
		const double lo = mn(mn(q0, qm1), qp1);
		const double hi = mx(mx(q0, qm1), qp1);
		const double a_minus = (coef_1 * q0 + coef_2 * qp1) + (coef_1 * qm1 + coef_2 * qm2);
		const double a_plus = (coef_1 * qp1 + coef_2 * qp2) + (coef_1 * q0 + coef_2 * qm1);
		double new_a_minus = (a_minus < lo) ? lo : (hi < a_minus) ? hi : a_minus;
		double new_a_plus = (a_plus < lo) ? lo : (hi < a_plus) ? hi : a_plus;
		const double a = q0;
		const double dq_minus = (a - new_a_minus);
		const double dq_plus = (new_a_plus - a);
		const double qa = dq_plus * dq_minus;
		// MC(qp1 - q0, q0 - qm1)
		const double sa = qp1 - q0, sb = q0 - qm1;
		const double dq0 = 0.5 * (sgnd(sa) + sgnd(sb)) * mn(0.5 * std::abs(sa + sb), mn(2.0 * std::abs(sa), 2.0 * std::abs(sb)));
		const double ext_minus = a - 0.5 * dq0, ext_plus = a + 0.5 * dq0;
		const double ovr_minus = (std::abs(dq_minus) >= 2.0 * std::abs(dq_plus)) ? (a - 2.0 * dq_plus) : new_a_minus;
		const double ovr_plus = (std::abs(dq_plus) >= 2.0 * std::abs(dq_minus)) ? (a + 2.0 * dq_minus) : new_a_plus;
		new_a_minus = (qa <= 0.0) ? ext_minus : ovr_minus;
		new_a_plus = (qa <= 0.0) ? ext_plus : ovr_plus;
		// FlattenShocks
		const double c = chi[e];
		am_out[e] = c * new_a_minus + (1. - c) * q0;
		ap_out[e] = c * new_a_plus + (1. - c) * q0;
	}
}

// One row of faces: left states L[6][*] (right edges of the cells before the faces), right states R[6][*] (left edges of the cells after them);
// pR[c] = the primitive component c at the cell AFTER the first face, its neighbour before the face is sd earlier, the transverse neighbours of the
// carbuncle differences sV / sW away.  hydro_system.hpp:881-1110 + HLLC.hpp:22-153, DIR selects the velocity permutation (:954-976).
//
// MODE: what becomes of the row's fluxes F and face velocities v (QuokkaSimulation.hpp:1117-1127, :1219-1229: flux_rk2 += 0.5 F after every stage's
// flux evaluation, here done while the row is in registers):
//   0  F, v stored
//   1  F, v stored and the accumulators SET to 0.0 + 0.5 F (what Saxpy onto the cleared flux_rk2 computes, sign of zero included)
//   2  the accumulators += 0.5 F; F and v themselves are not kept (stage 2 needs only the sum)
template <int DIR, int MODE>
__attribute__((always_inline)) inline void fluxRow(double gamma, double kB_user, double K_visc, int n, const double *const *L, const double *const *R, const double *const *pR, int64_t sd, int64_t sV, int64_t sW,
		    double *const *Fout, double *__restrict Vout, double *const *Acc, double *__restrict Vacc)
{
	constexpr int velN = x1Velocity_index + DIR, velV = x1Velocity_index + (DIR + 1) % 3, velW = x1Velocity_index + (DIR + 2) % 3;
	const double gm1 = gamma - 1.0;
	const double G = 0.5 * (1.0 + gamma);
	const double *__restrict qN = pR[velN];
	const double *__restrict qV = pR[velV];
	const double *__restrict qW = pR[velW];
#pragma omp simd
	for (int e = 0; e < n; ++e) {
		const double rho_L = L[0][e], rho_R = R[0][e];
		const double vx_L = L[1][e], vx_R = R[1][e], vy_L = L[2][e], vy_R = R[2][e], vz_L = L[3][e], vz_R = R[3][e];
		const double ke_L = 0.5 * rho_L * (vx_L * vx_L + vy_L * vy_L + vz_L * vz_L);
		const double ke_R = 0.5 * rho_R * (vx_R * vx_R + vy_R * vy_R + vz_R * vz_R);
		const double P_L = L[4][e], P_R = R[4][e];
		const double Eint_L = L[5][e], Eint_R = R[5][e];
		// eos(rp): cs = sqrt(gamma p / rho), e = p / ((gamma - 1) rho); ComputeEintFromPres = e * rho
		const double cs_L = std::sqrt(gamma * P_L / rho_L);
		const double E_L = (P_L / (gm1 * rho_L)) * rho_L + ke_L;
		const double cs_R = std::sqrt(gamma * P_R / rho_R);
		const double E_R = (P_R / (gm1 * rho_R)) * rho_R + ke_R;
		const double uL = L[velN][e], vL = L[velV][e], wL = L[velW][e];
		const double uR = R[velN][e], vR = R[velV][e], wR = R[velW][e];

		// :1019-1034
		const double du = qN[e] - qN[e - sd];
		const double dvl = mn(qV[e - sd + sV] - qV[e - sd], qV[e - sd] - qV[e - sd - sV]);
		const double dvr = mn(qV[e + sV] - qV[e], qV[e] - qV[e - sV]);
		double dw = mn(dvl, dvr);
		const double dwl = mn(qW[e - sd + sW] - qW[e - sd], qW[e - sd] - qW[e - sd - sW]);
		const double dwr = mn(qW[e + sW] - qW[e], qW[e] - qW[e - sW]);
		dw = mn(mn(dwl, dwr), dw);

		// ---- HLLC.hpp:27-36
		const double wl = std::sqrt(rho_L);
		const double wr = std::sqrt(rho_R);
		const double norm = 1. / (wl + wr);
		const double u_tilde = (wl * uL + wr * uR) * norm;
		const double v_tilde = (wl * vL + wr * vR) * norm;
		const double w_tilde = (wl * wL + wr * wR) * norm;
		const double vsq_tilde = u_tilde * u_tilde + v_tilde * v_tilde + w_tilde * w_tilde;
		const double H_L = (E_L + P_L) / rho_L;
		const double H_R = (E_R + P_R) / rho_R;
		const double H_tilde = (wl * H_L + wr * H_R) * norm;
		const double dU = uL - uR;
		// ComputeOtherDerivatives (eos rp): dedr = 0, deint_dP = 1 / ((gamma - 1) rho), dRho_dP = 1 / ((p / rho) * k_B / k_B_user), G = (1 + gamma) / 2
		const double dedp_L = 1.0 / (gm1 * rho_L), dedp_R = 1.0 / (gm1 * rho_R);
		const double drdp_L = 1.0 / ((P_L / rho_L) * C::k_B / kB_user), drdp_R = 1.0 / ((P_R / rho_R) * C::k_B / kB_user);
		const double C_tilde_rho = 0.5 * ((Eint_L / rho_L) + (Eint_R / rho_R) + rho_L * 0.0 + rho_R * 0.0);
		const double C_tilde_P = 0.5 * ((Eint_L / rho_L) * drdp_L + (Eint_R / rho_R) * drdp_R + rho_L * dedp_L + rho_R * dedp_R);
		const double cs_exp = H_tilde - 0.5 * vsq_tilde - C_tilde_rho;
		const double cs_tilde = (cs_exp <= 0) ? (0.5 * (cs_L + cs_R)) : std::sqrt(cs_exp / C_tilde_P);
		const double s_NL = 0.5 * G * mx(dU, 0.);
		const double s_NR = 0.5 * G * mx(dU, 0.);
		const double S_L = mn(uL - (cs_L + s_NL), u_tilde - (cs_tilde + s_NL));
		const double S_R = mx(uR + (cs_R + s_NR), u_tilde + (cs_tilde + s_NR));
		// :91-93
		const double cs_max = mx(cs_L, cs_R);
		const double tp = mn(1., (cs_max - mn(du, 0.)) / (cs_max - mn(dw, 0.)));
		const double theta = tp * tp * tp * tp;
		// :97-98
		const double S_star = (theta * (P_R - P_L) + (rho_L * uL * (S_L - uL) - rho_R * uR * (S_R - uR))) / (rho_L * (S_L - uL) - rho_R * (S_R - uR));
		// :102-107
		const double vmag_L = std::sqrt(uL * uL + vL * vL + wL * wL);
		const double vmag_R = std::sqrt(uR * uR + vR * vR + wR * wR);
		const double chi = mn(1., mx(vmag_L, vmag_R) / cs_max);
		const double phi = chi * (2. - chi);
		const double P_LR = 0.5 * (P_L + P_R) + 0.5 * phi * (rho_L * (S_L - uL) * (S_star - uL) + rho_R * (S_R - uR) * (S_star - uR));
		const double UL[6] = {rho_L, rho_L * uL, rho_L * vL, rho_L * wL, E_L, Eint_L};
		const double UR[6] = {rho_R, rho_R * uR, rho_R * vR, rho_R * wR, E_R, Eint_R};
		const double DL[6] = {0., 1., 0., 0., uL, 0.};
		const double DR[6] = {0., 1., 0., 0., uR, 0.};
		const double DS[6] = {0., 1., 0., 0., S_star, 0.};
		const double SLP = S_L * P_LR;
		const double SRP = S_R * P_LR;
		const double dSL = S_L - S_star;
		const double dSR = S_R - S_star;
		const bool takeL = S_L > 0.0;
		const bool takeSL = (S_star > 0.0) && (S_L <= 0.0);
		const bool takeSR = (S_star <= 0.0) && (S_R >= 0.0);
		double Fc[6];
#pragma GCC unroll 6
		for (int m = 0; m < 6; ++m) {
			const double F_L = uL * UL[m] + P_L * DL[m];
			const double F_R = uR * UR[m] + P_R * DR[m];
			const double F_starL = (S_star * (S_L * UL[m] - F_L) + SLP * DS[m]) / dSL;
			const double F_starR = (S_star * (S_R * UR[m] - F_R) + SRP * DS[m]) / dSR;
			Fc[m] = takeL ? F_L : (takeSL ? F_starL : (takeSR ? F_starR : F_R));
		}
		// ---- hydro_system.hpp:1054-1091
		double div_v = du;
		div_v = div_v + 0.5 * (dvl + dvr);
		div_v = div_v + 0.5 * (dwl + dwr);
		const double viscosity = K_visc * mx(-div_v, 0.);
		// the state vectors of the viscosity term are in (x1, x2, x3) component order with sL.u, sL.v, sL.w = the permuted velocities (:1057-1069)
		double F[6];
#pragma GCC unroll 6
		for (int m = 0; m < 6; ++m) {
			F[m] = Fc[m] + viscosity * (UL[m] - UR[m]);
		}
		F[velN] = Fc[x1Momentum_index];
		F[velV] = Fc[x2Momentum_index];
		F[velW] = Fc[x3Momentum_index];
		const double vface = (F[density_index] >= 0.) ? (F[density_index] / rho_R) : (F[density_index] / rho_L);
		if constexpr (MODE != 2) {
			Vout[e] = vface;
#pragma GCC unroll 6
			for (int m = 0; m < 6; ++m) {
				Fout[m][e] = F[m];
			}
		}
		if constexpr (MODE == 1) {
			Vacc[e] = 0.0 + 0.5 * vface;
#pragma GCC unroll 6
			for (int m = 0; m < 6; ++m) {
				Acc[m][e] = 0.0 + 0.5 * F[m];
			}
		}
		if constexpr (MODE == 2) {
			Vacc[e] = Vacc[e] + 0.5 * vface;
#pragma GCC unroll 6
			for (int m = 0; m < 6; ++m) {
				Acc[m][e] = Acc[m][e] + 0.5 * F[m];
			}
		}
	}
}

#endif // ORACLE_FUSED_IMPL

} // namespace fused

inline auto fusedHydroFluxesApplicable(HydroTraits const &tr, int reconstructionOrder, bool is_mhd) -> bool
{
	return kEosVariant == 0 && tr.ndim == 3 && g_spacedim == 3 && tr.gamma() != 1.0 && tr.nscalars == 0 && tr.nmscalars == 0 && !tr.reconstruct_eint &&
	       reconstructionOrder == 3 && !is_mhd;
}

// fluxes and face velocities of one box from its ghost-filled conserved state (4 ghost cells); F[d], V[d]: nodal in d, no ghost cells.
// mode 0: F, V written.  mode 1: F, V written and Facc, Vacc = 0.0 + 0.5 F (stage 1 of RK2: the Saxpy onto the cleared flux_rk2).
// mode 2: Facc, Vacc += 0.5 F and F, V are not touched (stage 2).
void fusedHydroFluxesBox(HydroTraits const &tr, Array4<const double> const &U, Box const &vb, std::array<Array4<double>, 3> const &F,
			 std::array<Array4<double>, 3> const &V, double K_visc, fused::Work &w, int mode = 0, std::array<Array4<double>, 3> const &Facc = {},
			 std::array<Array4<double>, 3> const &Vacc = {});

// What follows a stage's flux evaluation for the cells of one box (HydroSimulation::rhsPdvPredict + limitsAndSync; reference
// src/hydro/hydro_system.hpp:448-497 ComputeRhsFromFluxes + PredictStep, :775-814 AddInternalEnergyPdV, :696-773 EnforceLimits, :816-850
// SyncDualEnergy) in ONE pass over rows of x: Unew = limits(Uold + dt * rhs(F, V)), redo(i,j,k) = the flag PredictStep sets from the state BEFORE
// the limits.  Returns the number of flagged cells; when any box returns non-zero the caller discards Unew and takes the operator path
// (first-order flux correction), which rewrites every cell this function wrote.
auto fusedHydroUpdateBox(HydroTraits const &tr, Array4<const double> const &Uold, Array4<double> const &Unew, Box const &vb,
			 std::array<Array4<const double>, 3> const &F, std::array<Array4<const double>, 3> const &V, double const dx[3], double dt, double densityFloor,
			 double tempFloor, bool dualEnergy, Array4<int> const &redo) -> long;

#ifdef ORACLE_FUSED_IMPL
// fluxes and face velocities of one box from its ghost-filled conserved state (4 ghost cells); F[d], V[d]: nodal in d, no ghost cells.
// One clone per vector ISA, chosen when the library loads (the .so is built in one container and run on another machine's host cores); vector
// width changes no bit: every lane does the scalar arithmetic, and -ffp-contract=off keeps multiply-adds apart in every clone.
namespace fused
{
template <int MODE>
__attribute__((always_inline)) inline void fluxesBoxImpl(HydroTraits const &tr, Array4<const double> const &U, Box const &vb, std::array<Array4<double>, 3> const &F,
						       std::array<Array4<double>, 3> const &V, double K_visc, fused::Work &w, std::array<Array4<double>, 3> const &Facc,
						       std::array<Array4<double>, 3> const &Vacc)
{
	const double gamma = tr.gamma();
	const double gm1 = gamma - 1.0;
	const double kBu = tr.eos.tr.boltzmann_constant;
	constexpr int NGH = 4;
	const int nx = vb.length(0) + 2 * NGH, ny = vb.length(1) + 2 * NGH, nz = vb.length(2) + 2 * NGH;
	const int64_t sx = 1, sy = nx, sz = static_cast<int64_t>(nx) * ny;
	const int64_t ncell = sz * nz;
	w.resize(static_cast<size_t>(ncell), static_cast<size_t>(nx));
	auto at = [&](int i, int j, int k) -> int64_t { return (i - vb.lo[0] + NGH) + sy * (j - vb.lo[1] + NGH) + sz * (k - vb.lo[2] + NGH); };
	double *P[6];
	for (int n = 0; n < 6; ++n) {
		P[n] = w.prim.data() + n * ncell;
	}

	// ---- ConservedToPrimitive (hydro_system.hpp:138-196) on valid + 4
	for (int k = vb.lo[2] - NGH; k <= vb.hi[2] + NGH; ++k) {
		for (int j = vb.lo[1] - NGH; j <= vb.hi[1] + NGH; ++j) {
			const double *__restrict u0 = &U(vb.lo[0] - NGH, j, k, 0);
			const int64_t ns = U.nstride;
			const int64_t o = at(vb.lo[0] - NGH, j, k);
#pragma omp simd
			for (int e = 0; e < nx; ++e) {
				const double rho = u0[e], px = u0[e + ns], py = u0[e + 2 * ns], pz = u0[e + 3 * ns], E = u0[e + 4 * ns], Eaux = u0[e + 5 * ns];
				const double vx = px / rho, vy = py / rho, vz = pz / rho;
				const double kinetic_energy = 0.5 * rho * (vx * vx + vy * vy + vz * vz);
				const double thermal_energy = E - kinetic_energy;
				// ComputePressure: e = (rho == 0) ? 0 : Eint / rho; p = (gamma - 1) rho e
				const double es = (rho == 0.0) ? 0.0 : thermal_energy / rho;
				P[0][o + e] = rho;
				P[1][o + e] = vx;
				P[2][o + e] = vy;
				P[3][o + e] = vz;
				P[4][o + e] = gm1 * rho * es;
				P[5][o + e] = Eaux;
			}
		}
	}

	// ---- ComputeFlatteningCoefficients<DIR> (hydro_system.hpp:531-626) on valid + 2
	constexpr double beta_max = 0.85, beta_min = 0.75, Zmax = 0.75, Zmin = 0.25;
	const int64_t sdir[3] = {sx, sy, sz};
	for (int d = 0; d < 3; ++d) {
		const int64_t s = sdir[d];
		double *__restrict chi = w.chi[d].data();
		const double *__restrict Pr = P[4];
		const double *__restrict rho = P[0];
		const double *__restrict vel = P[1 + d];
		for (int k = vb.lo[2] - 2; k <= vb.hi[2] + 2; ++k) {
			for (int j = vb.lo[1] - 2; j <= vb.hi[1] + 2; ++j) {
				const int64_t o = at(vb.lo[0] - 2, j, k);
				const int n = vb.length(0) + 4;
#pragma omp simd
				for (int e = 0; e < n; ++e) {
					const int64_t c = o + e;
					const double Pp2 = Pr[c + 2 * s], Pp1 = Pr[c + s], Pc = Pr[c], Pm1 = Pr[c - s], Pm2 = Pr[c - 2 * s];
					const double beta_denom = std::abs(Pp2 - Pm2);
					const double beta = (beta_denom != 0) ? (std::abs(Pp1 - Pm1) / beta_denom) : 0;
					const double chi_min = mx(0., mn(1., (beta_max - beta) / (beta_max - beta_min)));
					const double cs = std::sqrt(gamma * Pc / rho[c]);
					const double K_S = (cs * cs) * rho[c];
					const double Z = std::abs(Pp1 - Pm1) / K_S;
					const double compressive = mx(chi_min, mn(1., (Zmax - Z) / (Zmax - Zmin)));
					chi[c] = (vel[c + s] < vel[c - s]) ? compressive : 1.0;
				}
			}
		}
	}
	// ---- the combined coefficient of FlattenShocks (hydro_system.hpp:655-669) on valid + 1
	{
		const double *__restrict cx = w.chi[0].data();
		const double *__restrict cy = w.chi[1].data();
		const double *__restrict cz = w.chi[2].data();
		double *__restrict cm = w.chim.data();
		for (int k = vb.lo[2] - 1; k <= vb.hi[2] + 1; ++k) {
			for (int j = vb.lo[1] - 1; j <= vb.hi[1] + 1; ++j) {
				const int64_t o = at(vb.lo[0] - 1, j, k);
				const int n = vb.length(0) + 2;
#pragma omp simd
				for (int e = 0; e < n; ++e) {
					const int64_t c = o + e;
					double m = mn(mn(cx[c - 1], cx[c]), cx[c + 1]);
					m = mn(mn(mn(m, cy[c - sy]), cy[c]), cy[c + sy]);
					m = mn(mn(mn(m, cz[c - sz]), cz[c]), cz[c + sz]);
					cm[c] = m;
				}
			}
		}
	}

	const int n0 = vb.length(0);
	double *am[6], *ap[6], *apPrev[6];
	for (int n = 0; n < 6; ++n) {
		am[n] = w.am.data() + n * nx;
		ap[n] = w.ap.data() + n * nx;
		apPrev[n] = w.apPrev.data() + n * nx;
	}
	// ---- X: per row the edge states of cells lo-1 .. hi+1, then the faces lo .. hi+1 (left state = right edge of the cell before)
	for (int k = vb.lo[2]; k <= vb.hi[2]; ++k) {
		for (int j = vb.lo[1]; j <= vb.hi[1]; ++j) {
			const int64_t o = at(vb.lo[0] - 1, j, k);
			for (int n = 0; n < 6; ++n) {
				edgesRow(P[n] + o, sx, w.chim.data() + o, n0 + 2, am[n], ap[n]);
			}
			const double *Lp[6], *Rp[6], *pR[6];
			double *Fo[6] = {}, *Ao[6] = {};
			for (int n = 0; n < 6; ++n) {
				Lp[n] = ap[n];	   // cell lo-1+e is the cell before face lo+e
				Rp[n] = am[n] + 1; // cell lo+e
				pR[n] = P[n] + o + 1;
				if constexpr (MODE != 2) {
					Fo[n] = &F[0](vb.lo[0], j, k, n);
				}
				if constexpr (MODE != 0) {
					Ao[n] = &Facc[0](vb.lo[0], j, k, n);
				}
			}
			fluxRow<0, MODE>(gamma, kBu, K_visc, n0 + 1, Lp, Rp, pR, sx, sy, sz, Fo, (MODE != 2) ? &V[0](vb.lo[0], j, k, 0) : nullptr, Ao,
					 (MODE != 0) ? &Vacc[0](vb.lo[0], j, k, 0) : nullptr);
		}
	}
	// ---- Y and Z: march along the direction, rows of x; the right edges of the previous row are the left states of the faces between the rows
	for (int d = 1; d <= 2; ++d) {
		const int64_t s = sdir[d];
		const int ot = 3 - d; // the other transverse direction
		for (int t = vb.lo[ot]; t <= vb.hi[ot]; ++t) {
			for (int m = vb.lo[d] - 1; m <= vb.hi[d] + 1; ++m) {
				const int j = (d == 1) ? m : t, k = (d == 1) ? t : m;
				const int64_t o = at(vb.lo[0], j, k);
				for (int n = 0; n < 6; ++n) {
					std::swap(ap[n], apPrev[n]);
					edgesRow(P[n] + o, s, w.chim.data() + o, n0, am[n], ap[n]);
				}
				if (m >= vb.lo[d]) { // the face between rows m-1 and m
					const double *Lp[6], *Rp[6], *pR[6];
					double *Fo[6] = {}, *Ao[6] = {};
					for (int n = 0; n < 6; ++n) {
						Lp[n] = apPrev[n];
						Rp[n] = am[n];
						pR[n] = P[n] + o;
						if constexpr (MODE != 2) {
							Fo[n] = &F[d](vb.lo[0], j, k, n);
						}
						if constexpr (MODE != 0) {
							Ao[n] = &Facc[d](vb.lo[0], j, k, n);
						}
					}
					double *const Vo = (MODE != 2) ? &V[d](vb.lo[0], j, k, 0) : nullptr;
					double *const Va = (MODE != 0) ? &Vacc[d](vb.lo[0], j, k, 0) : nullptr;
					if (d == 1) {
						fluxRow<1, MODE>(gamma, kBu, K_visc, n0, Lp, Rp, pR, sy, sz, sx, Fo, Vo, Ao, Va);
					} else {
						fluxRow<2, MODE>(gamma, kBu, K_visc, n0, Lp, Rp, pR, sz, sx, sy, Fo, Vo, Ao, Va);
					}
				}
			}
		}
	}
}
} // namespace fused

__attribute__((target_clones("avx512f", "avx2", "default"))) void fusedHydroFluxesBox(HydroTraits const &tr, Array4<const double> const &U, Box const &vb, std::array<Array4<double>, 3> const &F,
				std::array<Array4<double>, 3> const &V, double K_visc, fused::Work &w, int mode, std::array<Array4<double>, 3> const &Facc,
				std::array<Array4<double>, 3> const &Vacc)
{
	if (mode == 1) {
		fused::fluxesBoxImpl<1>(tr, U, vb, F, V, K_visc, w, Facc, Vacc);
	} else if (mode == 2) {
		fused::fluxesBoxImpl<2>(tr, U, vb, F, V, K_visc, w, Facc, Vacc);
	} else {
		fused::fluxesBoxImpl<0>(tr, U, vb, F, V, K_visc, w, Facc, Vacc);
	}
}

__attribute__((target_clones("avx512f", "avx2", "default"))) auto fusedHydroUpdateBox(HydroTraits const &tr, Array4<const double> const &Uold, Array4<double> const &Unew, Box const &vb,
				std::array<Array4<const double>, 3> const &F, std::array<Array4<const double>, 3> const &V, double const dx[3], double dt,
				double densityFloor, double tempFloor, bool dualEnergy, Array4<int> const &redo) -> long
{
	const double gamma = tr.gamma();
	const double gm1 = gamma - 1.0;
	const double kBu = tr.eos.tr.boltzmann_constant;
	const double mu = tr.eos.tr.mean_molecular_weight / C::m_u; // (estate.mu of EOS.hpp:74-159)
	const double m_nucleon = C::m_u;
	const double idx0 = 1.0 / dx[0], idx1 = 1.0 / dx[1], idx2 = 1.0 / dx[2];
	const double dx0 = dx[0], dx1 = dx[1], dx2 = dx[2];
	constexpr double tiny = std::numeric_limits<double>::min();
	constexpr double eta = 1.0e-3;
	const int n0 = vb.length(0);
	const int64_t nso = Uold.nstride, nsn = Unew.nstride;
	long nbad = 0;
	for (int k = vb.lo[2]; k <= vb.hi[2]; ++k) {
		for (int j = vb.lo[1]; j <= vb.hi[1]; ++j) {
			const double *__restrict uo = &Uold(vb.lo[0], j, k, 0);
			double *__restrict un = &Unew(vb.lo[0], j, k, 0);
			int *__restrict rf = &redo(vb.lo[0], j, k);
			const double *fx[6], *fy[6], *fyp[6], *fz[6], *fzp[6];
			for (int m = 0; m < 6; ++m) {
				fx[m] = &F[0](vb.lo[0], j, k, m);
				fy[m] = &F[1](vb.lo[0], j, k, m);
				fyp[m] = &F[1](vb.lo[0], j + 1, k, m);
				fz[m] = &F[2](vb.lo[0], j, k, m);
				fzp[m] = &F[2](vb.lo[0], j, k + 1, m);
			}
			const double *__restrict vx = &V[0](vb.lo[0], j, k, 0);
			const double *__restrict vy = &V[1](vb.lo[0], j, k, 0);
			const double *__restrict vyp = &V[1](vb.lo[0], j + 1, k, 0);
			const double *__restrict vz = &V[2](vb.lo[0], j, k, 0);
			const double *__restrict vzp = &V[2](vb.lo[0], j, k + 1, 0);
			long bad = 0;
#pragma omp simd reduction(+ : bad)
			for (int e = 0; e < n0; ++e) {
				const double rho = uo[e], px = uo[e + nso], py = uo[e + 2 * nso], pz = uo[e + 3 * nso], E = uo[e + 4 * nso], Eaux = uo[e + 5 * nso];
				const double old[6] = {rho, px, py, pz, E, Eaux};
				// ---- ComputeRhsFromFluxes (:448-473)
				double r[6];
#pragma GCC unroll 6
				for (int m = 0; m < 6; ++m) {
					double q = idx0 * (fx[m][e] - fx[m][e + 1]);
					q = q + idx1 * (fy[m][e] - fyp[m][e]);
					q = q + idx2 * (fz[m][e] - fzp[m][e]);
					r[m] = q;
				}
				// ---- AddInternalEnergyPdV (:775-814), the branch of an unflagged cell (every flag is redo_none when a stage starts)
				const double ux = px / rho, uy = py / rho, uz = pz / rho;
				const double kinetic_energy = 0.5 * rho * (ux * ux + uy * uy + uz * uz);
				const double thermal_energy = E - kinetic_energy;
				const double es = (rho == 0.0) ? 0.0 : thermal_energy / rho;
				const double Pgas = gm1 * rho * es;
				double div_v = (vx[e + 1] - vx[e]) / dx0;
				div_v = div_v + (vyp[e] - vy[e]) / dx1;
				div_v = div_v + (vzp[e] - vz[e]) / dx2;
				r[5] = r[5] + (-Pgas * div_v);
				// ---- PredictStep (:475-497)
				double s[6];
#pragma GCC unroll 6
				for (int m = 0; m < 6; ++m) {
					s[m] = old[m] + dt * r[m];
				}
				const bool valid = s[0] > 0.;
				rf[e] = valid ? redo_none : redo_redo;
				bad += valid ? 0 : 1;
				// ---- EnforceLimits (:696-773; no scalars)
				const double rho_new = (s[0] < densityFloor) ? densityFloor : s[0];
				{
					const double w1 = s[1] / rho_new, w2 = s[2] / rho_new, w3 = s[3] / rho_new;
					const double Ekin = 0.5 * rho_new * (w1 * w1 + w2 * w2 + w3 * w3);
					// ComputeTgasFromEint (EOS.hpp:74-114): T = e mu m_u (gamma - 1) / k_B with e = Eint / rho; Tgas = T k_B / k_B_user
					const double primTemp = ((((((s[4] - Ekin) / rho_new) * mu) * m_nucleon) * gm1) / C::k_B) * C::k_B / kBu;
					// ComputeEintFromTgas (EOS.hpp:116-159): p = rho T k_B / (mu m_u), e = p / ((gamma - 1) rho); Eint = e rho k_B_user / k_B
					const double floorEint = (((((rho_new * tempFloor) * C::k_B) / (mu * m_nucleon)) / (gm1 * rho_new)) * rho_new) * kBu / C::k_B;
					const double auxTemp = (((((s[5] / rho_new) * mu) * m_nucleon) * gm1) / C::k_B) * C::k_B / kBu;
					const bool lim = rho_new > tiny;
					s[4] = (lim && (primTemp < tempFloor)) ? (Ekin + floorEint) : s[4];
					s[5] = (lim && (auxTemp < tempFloor)) ? floorEint : s[5];
				}
				s[0] = rho_new;
				// ---- SyncDualEnergy (:816-850; its abort on a non-positive density is the operator path's to raise: such a cell is flagged above)
				if (dualEnergy) {
					const double Ekin = (s[1] * s[1] + s[2] * s[2] + s[3] * s[3]) / (2.0 * s[0]);
					const double Eint_cons = s[4] - Ekin;
					const bool keep = Eint_cons > eta * s[4];
					const double Eint_aux = s[5];
					s[5] = keep ? Eint_cons : Eint_aux;
					s[4] = keep ? s[4] : (Eint_aux + Ekin);
				}
#pragma GCC unroll 6
				for (int m = 0; m < 6; ++m) {
					un[e + m * nsn] = s[m];
				}
			}
			nbad += bad;
		}
	}
	return nbad;
}

#endif // ORACLE_FUSED_IMPL

} // namespace oracle

#endif // ORACLE_HYDRO_FUSED_HPP_
