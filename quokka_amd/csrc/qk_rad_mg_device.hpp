// qk_rad_mg_device.hpp — per-cell device arithmetic of the MULTIGROUP matter-radiation exchange, NG photon groups in registers.
// Counterparts (same association order, -ffp-contract=off):
//   reference src/radiation/source_terms_multi_group.hpp   ComputeModelDependentKappaEAndKappaP, ComputeModelDependentKappaFAndDeltaTerms,
//                                                           ComputeJacobianForGas, SolveGasRadiationEnergyExchange, UpdateFlux, AddSourceTermsMultiGroup
//   reference src/radiation/radiation_system.hpp           ComputePlanckEnergyFractions :430-461, ComputeThermalRadiationMultiGroup :483-497,
//                                                           ComputeThermalRadiationTempDerivativeMultiGroup :505-513, SolveLinearEqs :547-558,
//                                                           ComputeRadQuantityExponents :1169-1250, ComputeGroupMeanOpacity :1252-1287,
//                                                           PlanckFunction :1311-1326, ComputeDiffusionFluxMeanOpacity :1328-1352
//   reference src/radiation/planck_integral.hpp            interpolate_planck_integral, integrate_planck_from_0_to_x
// Gas + radiation only (ISM_Traits defaults: no dust / photoelectric / cooling / cosmic-ray terms).  Hyper-parameters as the reference
// compiles them (radiation_system.hpp:35-45): include_delta_B, use_diffuse_flux_mean_opacity, include_work_term_in_source, enable_dE_constrain on;
// special_edge_bin_slopes, force_rad_floor_in_iteration, use_D_as_base, PPL_free_slope_st_total off; max_iter_to_update_alpha_E = 5.
// The Planck-integral table is data/planck_integral_table.inc (computed from the definition by tools/make_planck_table.py).
#ifndef QK_RAD_MG_DEVICE_HPP_
#define QK_RAD_MG_DEVICE_HPP_

#include "qk_planck.hpp"
#include "qk_rad_device.hpp"

namespace qk
{

enum { MG_PIECEWISE_CONSTANT = 1, MG_PPL_FIXED_SLOPE = 2, MG_PPL_FULL_SPECTRUM = 3 }; // OpacityModel (radiation_system.hpp:64-71)

constexpr int PLANCK_INTERP_SIZE = 1000;
constexpr double PLANCK_LOG_X_MIN = -3., PLANCK_LOG_X_MAX = 2.;
__device__ const double d_planck_Y[PLANCK_INTERP_SIZE] = {
#include "../data/planck_integral_table.inc"
};

// planck_integral.hpp:22-262 (USE_SECOND_ORDER = false): the body shared with the host mirror (qk_planck.hpp) on the device-resident table
struct DevicePlanckTable {
	QK_DEV auto operator[](int j) const -> double { return d_planck_Y[j]; }
};
QK_DEV auto integratePlanckFrom0ToX(double x) -> double { return planck::fractionBelow(x, DevicePlanckTable{}); }

// the multigroup part of RadSystem_Traits + the DefineOpacityExponentsAndLowerValues closed set (include/quokka_amd.h)
template <int NG> struct RadMG {
	double bnd[NG + 1];
	double kexp[NG + 1], klow[NG + 1];
	double energy_unit, kB; // RadSystem_Traits::energy_unit, EOS_Traits::boltzmann_constant
	double k_rho_exp, k_T_ref, k_T_exp;
	double cool[NG]; // DefineNetCoolingRate(T, n)[g] = cool[g] * T (closed set; the dust instantiations)
	int model;
	__host__ RadMG(qk_rad_traits const &t, double kB_user) : energy_unit(t.energy_unit), kB(kB_user), k_rho_exp(t.mg_kappa_rho_exponent), k_T_ref(t.mg_kappa_T_ref),
								 k_T_exp(t.mg_kappa_T_exponent), model(t.mg_opacity_model)
	{
		for (int g = 0; g < NG + 1; ++g) {
			bnd[g] = t.rad_boundaries[g];
			kexp[g] = t.mg_kappa_exponent[g];
			klow[g] = t.mg_kappa_lower[g];
		}
		for (int g = 0; g < NG; ++g) {
			cool[g] = t.cooling_linear_coeff[g];
		}
	}
	// DefineOpacityExponentsAndLowerValues(rad_boundaries, rho, Tgas) is about to be read at (rho, Tgas): nothing to do for the closed set (the
	// exponents are constants, the lower values are scaled in lower()).  A type derived from this one in a problem's translation unit
	// (quokka_amd/host/qk_problem_kernels.hpp: ProblemRadMG) refreshes kexp / klow from the problem's compiled hook here.
	QK_DEV void at(double /*rho*/, double /*T*/) {}
	// DefineOpacityExponentsAndLowerValues(rad_boundaries, rho, Tgas): the factor every lower value carries
	QK_DEV auto lowerScale(double rho, double T) const -> double
	{
		double f = 1.0;
		if (k_rho_exp == -1.0) {
			f = 1.0 / rho;
		} else if (k_rho_exp != 0.0) {
			f = pow(rho, k_rho_exp);
		}
		if (k_T_exp != 0.0) {
			f = f * pow(T / k_T_ref, k_T_exp);
		}
		return f;
	}
	// lower value of group edge g (k_rho_exp == -1: kappa / rho, one division as the problem files write it; scale-free sets: the constant itself)
	QK_DEV auto lower(int g, double rho, double T) const -> double
	{
		if (k_rho_exp == -1.0 && k_T_exp == 0.0) {
			return klow[g] / rho;
		}
		if (k_rho_exp == 0.0 && k_T_exp == 0.0) {
			return klow[g];
		}
		return klow[g] * lowerScale(rho, T);
	}
};

// radiation_system.hpp:430-461
template <int NG> QK_DEV void planckEnergyFractions(RadMG<NG> const &m, double T, double f[NG])
{
	planck::groupFractions<NG>(m.bnd, m.energy_unit / (m.kB * T), DevicePlanckTable{}, f);
}

// :483-497 and :505-513 from one set of fractions (the reference evaluates the fractions twice at the same temperature)
// HOOKS (the dust instantiations): thermal_model 1 is RadDustMG's specialisation of the two functions (test_rad_dust_MG.cpp:83-104): a T, no floor
template <int NG, bool HOOKS = false> QK_DEV void thermalRadiationMG(Rad const &r, const double frac[NG], double T, double E[NG])
{
	if constexpr (HOOKS) {
		if (r.thermal_model == 1) {
			const double power = r.arad * T;
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				E[g] = power * frac[g];
			}
			return;
		}
	}
	const double power = r.arad * r.pow4(T);
#pragma unroll
	for (int g = 0; g < NG; ++g) {
		E[g] = power * frac[g];
		if (E[g] < r.Erad_floor) {
			E[g] = r.Erad_floor;
		}
	}
}
template <int NG, bool HOOKS = false> QK_DEV void thermalRadiationTempDerivativeMG(Rad const &r, const double frac[NG], double T, double dE[NG])
{
	double d_power_dt = 4. * r.arad * r.pow3(T);
	if constexpr (HOOKS) {
		if (r.thermal_model == 1) {
			d_power_dt = r.arad;
		}
	}
#pragma unroll
	for (int g = 0; g < NG; ++g) {
		dE[g] = d_power_dt * frac[g];
	}
}

QK_DEV auto sgnInt(double v) -> int { return static_cast<int>(0. < v) - static_cast<int>(v < 0.); }
QK_DEV auto minmodFunc(double a, double b) -> double { return 0.5 * (sgnInt(a) + sgnInt(b)) * fmin(fabs(a), fabs(b)); }

// :1169-1250 (special_edge_bin_slopes = false, PPL_free_slope_st_total = false)
template <int NG> QK_DEV void radQuantityExponents(RadMG<NG> const &m, const double quant[NG], double exponents[NG])
{
	constexpr double inf = 1.7976931348623157e308; // std::numeric_limits<double>::max() (radiation_system.hpp:61)
	double logslopes[NG];
	double center_prev = 0., mean_prev = 0.;
#pragma unroll
	for (int g = 0; g < NG; ++g) {
		const double bin_center = sqrt(m.bnd[g] * m.bnd[g + 1]);
		const double quant_mean = quant[g] / (m.bnd[g + 1] - m.bnd[g]);
		if (g > 0) {
			if (quant_mean == 0.0 && mean_prev == 0.0) {
				logslopes[g - 1] = 0.0;
			} else if (mean_prev * quant_mean <= 0.0) {
				logslopes[g - 1] = (quant_mean > mean_prev) ? inf : -inf;
			} else {
				logslopes[g - 1] = log(fabs(quant_mean / mean_prev)) / log(bin_center / center_prev);
			}
		}
		center_prev = bin_center;
		mean_prev = quant_mean;
	}
#pragma unroll
	for (int g = 0; g < NG; ++g) {
		if (g == 0 || g == NG - 1) {
			exponents[g] = -1.0;
		} else {
			exponents[g] = minmodFunc(logslopes[g - 1], logslopes[g]);
		}
	}
}

// :1252-1287
template <int NG> QK_DEV void groupMeanOpacity(RadMG<NG> const &m, const double kappa_lower[NG + 1], const double ratios[NG], const double alpha_quant[NG], double kappa[NG])
{
#pragma unroll
	for (int g = 0; g < NG; ++g) {
		double alpha = alpha_quant[g] + 1.0;
		if (alpha > 100.) {
			kappa[g] = kappa_lower[g] * pow(ratios[g], m.kexp[g]);
			continue;
		}
		if (alpha < -100.) {
			kappa[g] = kappa_lower[g];
			continue;
		}
		double part1;
		if (fabs(alpha) < 1e-8) {
			part1 = log(ratios[g]);
		} else {
			part1 = (pow(ratios[g], alpha) - 1.0) / alpha;
		}
		alpha += m.kexp[g];
		double part2;
		if (fabs(alpha) < 1e-8) {
			part2 = log(ratios[g]);
		} else {
			part2 = (pow(ratios[g], alpha) - 1.0) / alpha;
		}
		kappa[g] = kappa_lower[g] / part1 * part2;
	}
}

// :1311-1326 (4 pi B(nu) / c); x^3 and T^4 are std::pow(x, 3) / std::pow(T, 4), faithfully rounded (qk_rad_device.hpp)
template <int NG> QK_DEV auto planckFunction(Rad const &r, RadMG<NG> const &m, double nu, double T) -> double
{
	constexpr double PI4 = 97.40909103400242; // std::pow(M_PI, 4): the fourth power of the double nearest pi, correctly rounded
	return planck::spectralDensity(m.energy_unit / (m.kB * T), nu, PI4 / 15.0, r.arad * Rad::pow4Faithful(T), [](double x) { return Rad::pow3Faithful(x); });
}

template <int NG> struct OpacityTermsMG {
	double kappaE[NG], kappaP[NG], kappaF[NG], kappaPoverE[NG], delta_nu_kappa_B_at_edge[NG], alpha_P[NG], alpha_E[NG];
};

// source_terms_multi_group.hpp:7-60 (kappaF / delta terms of `ot` are left alone: they are recomputed before they are read)
template <int NG, class M>
QK_DEV void kappaEAndKappaP(M &m, double T, double rho, const double ratios[NG], const double fourPiBoverC[NG], const double Erad[NG], int n_iter,
			    OpacityTermsMG<NG> &ot)
{
	m.at(rho, T); // :16
	double lower[NG + 1];
#pragma unroll
	for (int g = 0; g < NG + 1; ++g) {
		lower[g] = m.lower(g, rho, T);
	}
	if (m.model == MG_PIECEWISE_CONSTANT) {
#pragma unroll
		for (int g = 0; g < NG; ++g) {
			ot.kappaP[g] = lower[g];
			ot.kappaE[g] = lower[g];
		}
	} else if (m.model == MG_PPL_FIXED_SLOPE) {
		double am1[NG];
#pragma unroll
		for (int g = 0; g < NG; ++g) {
			am1[g] = -1.0;
		}
		groupMeanOpacity<NG>(m, lower, ratios, am1, ot.kappaP);
#pragma unroll
		for (int g = 0; g < NG; ++g) {
			ot.kappaE[g] = ot.kappaP[g];
		}
	} else {
		if (n_iter < 5) { // max_iter_to_update_alpha_E
			radQuantityExponents<NG>(m, Erad, ot.alpha_E);
			radQuantityExponents<NG>(m, fourPiBoverC, ot.alpha_P);
		}
		groupMeanOpacity<NG>(m, lower, ratios, ot.alpha_E, ot.kappaE);
		groupMeanOpacity<NG>(m, lower, ratios, ot.alpha_P, ot.kappaP);
	}
#pragma unroll
	for (int g = 0; g < NG; ++g) {
		if (ot.kappaE[g] > 0.0) {
			ot.kappaPoverE[g] = ot.kappaP[g] / ot.kappaE[g];
		} else {
			ot.kappaPoverE[g] = 1.0;
		}
	}
}

// source_terms_multi_group.hpp:62-96 + ComputeDiffusionFluxMeanOpacity (radiation_system.hpp:1328-1352)
template <int NG, class M> QK_DEV void kappaFAndDeltaTerms(Rad const &r, M &m, double T, double rho, const double fourPiBoverC[NG], OpacityTermsMG<NG> &ot)
{
	m.at(rho, T); // :70 (the work term that follows the n == 0 call reads the exponents of the same point, :266)
	double delta_nu_B_at_edge[NG];
	// B at the group edges: each interior edge is the right edge of one group and the left edge of the next (same operands, evaluated once)
	double B_edge[NG + 1];
#pragma unroll
	for (int g = 0; g < NG + 1; ++g) {
		B_edge[g] = planckFunction<NG>(r, m, m.bnd[g], T);
	}
#pragma unroll
	for (int g = 0; g < NG; ++g) {
		const double nu_L = m.bnd[g];
		const double nu_R = m.bnd[g + 1];
		const double B_L = B_edge[g];
		const double B_R = B_edge[g + 1];
		const double kappa_L = m.lower(g, rho, T);
		const double kappa_R = kappa_L * pow(nu_R / nu_L, m.kexp[g]);
		ot.delta_nu_kappa_B_at_edge[g] = nu_R * kappa_R * B_R - nu_L * kappa_L * B_L;
		delta_nu_B_at_edge[g] = nu_R * B_R - nu_L * B_L;
	}
	if (m.model == MG_PIECEWISE_CONSTANT) {
#pragma unroll
		for (int g = 0; g < NG; ++g) {
			ot.kappaF[g] = ot.kappaP[g];
		}
	} else {
#pragma unroll
		for (int g = 0; g < NG; ++g) {
			double kF = (ot.kappaP[g] + 1. / 3. * ot.kappaE[g]) * fourPiBoverC[g] +
				    1. / 3. * (m.kexp[g] * ot.kappaE[g] * fourPiBoverC[g] - ot.delta_nu_kappa_B_at_edge[g]);
			const double denom = 4. / 3. * fourPiBoverC[g] - 1. / 3. * delta_nu_B_at_edge[g];
			if (denom <= 0.0) {
				kF = 0.0;
			} else {
				kF /= denom;
			}
			ot.kappaF[g] = kF;
		}
	}
}

template <int NG> QK_DEV auto sumOf(const double v[NG]) -> double
{
	double s = 0;
#pragma unroll
	for (int g = 0; g < NG; ++g) {
		s += v[g];
	}
	return s;
}

template <int NG> struct NewtonResultMG {
	double Egas, T_gas, T_d;
	double EradVec[NG], work[NG];
	OpacityTermsMG<NG> ot;
};

// source_terms_multi_group.hpp:149-358 (+ ComputeJacobianForGas :98-147 and SolveLinearEqs, radiation_system.hpp:547-558, in line)
template <int NG, class M>
QK_DEV void solveGasRadiationEnergyExchange(Rad const &r, M &m, Eos const &eos, double Egas0, const double Erad0Vec[NG], double rho, double dt,
					    int n_outer_iter, const double work[NG], const double vel_times_F[NG], const double Src[NG], NewtonResultMG<NG> &res,
					    int &n_newton_total, int &n_newton_max, int &n_solves, int &fail_newton)
{
	const double c = r.c;
	const double chat = r.chat;
	const double cscale = c / chat;
	const double Etot0 = Egas0 + cscale * (sumOf<NG>(Erad0Vec) + sumOf<NG>(Src));
	const EosCell ec(eos, rho); // (qk_rad_device.hpp: the EOS quotients with unchanging denominators through their reciprocals, same bits)

	double T_gas = __builtin_nan(""), T_d = __builtin_nan("");
	double Rvec[NG], tau[NG], work_local[NG], fourPiBoverC[NG], ratios[NG], frac[NG];
	OpacityTermsMG<NG> &ot = res.ot;
#pragma unroll
	for (int g = 0; g < NG; ++g) {
		ot.alpha_E[g] = 0.;
		ot.alpha_P[g] = 0.;
		ratios[g] = m.bnd[g + 1] / m.bnd[g];
		Rvec[g] = 0.;
		tau[g] = 0.;
		work_local[g] = 0.;
	}
	double Egas_guess = Egas0;
	double EradVec_guess[NG];
#pragma unroll
	for (int g = 0; g < NG; ++g) {
		EradVec_guess[g] = Erad0Vec[g];
	}

	const double resid_tol = 1.0e-11;
	const int maxIter = 100;
	int n = 0;
	for (; n < maxIter; ++n) {
		T_gas = ec.tgasFromEint(Egas_guess);
		T_d = T_gas;
		planckEnergyFractions<NG>(m, T_d, frac);
		thermalRadiationMG<NG>(r, frac, T_d, fourPiBoverC);
		kappaEAndKappaP<NG>(m, T_d, rho, ratios, fourPiBoverC, EradVec_guess, n, ot);
		if (n == 0) {
			kappaFAndDeltaTerms<NG>(r, m, T_d, rho, fourPiBoverC, ot);
			if (r.beta_order == 1) { // include_work_term_in_source
				if (n_outer_iter == 0) {
#pragma unroll
					for (int g = 0; g < NG; ++g) {
						if (m.model == MG_PIECEWISE_CONSTANT) {
							work_local[g] = vel_times_F[g] * ot.kappaF[g] * chat / (c * c) * dt;
						} else {
							work_local[g] = vel_times_F[g] * ot.kappaF[g] * chat / (c * c) * dt * (1.0 + m.kexp[g]);
						}
					}
				} else {
#pragma unroll
					for (int g = 0; g < NG; ++g) {
						work_local[g] = work[g];
					}
				}
			} else {
#pragma unroll
				for (int g = 0; g < NG; ++g) {
					work_local[g] = 0.0;
				}
			}
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				const double tau0 = dt * rho * ot.kappaP[g] * chat;
				tau[g] = tau0;
				Rvec[g] = (fourPiBoverC[g] - EradVec_guess[g] / ot.kappaPoverE[g]) * tau0 + work_local[g];
			}
		} else {
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				tau[g] = dt * rho * ot.kappaP[g] * chat;
				if (tau[g] > 0.0) {
					EradVec_guess[g] = ot.kappaPoverE[g] * (fourPiBoverC[g] - (Rvec[g] - work_local[g]) / tau[g]);
				}
			}
		}

		// ComputeJacobianForGas: the residuals first — the converged iteration (the last one of every solve) needs neither the emission
		// derivative nor the Jacobian entries, which the reference evaluates before it tests the residuals (:98-147, :300-303); same values
		const double Egas_diff = Egas_guess - Egas0;
		const double CR_heating = 0.0 * dt;
		const double F0 = Egas_diff + cscale * sumOf<NG>(Rvec) - CR_heating;
		double Fg[NG], Jg0[NG], Jgg[NG];
		double Fg_abs_sum = 0.0;
#pragma unroll
		for (int g = 0; g < NG; ++g) {
			const double Erad_diff = EradVec_guess[g] - Erad0Vec[g];
			Fg[g] = Erad_diff - (Rvec[g] + Src[g]);
			if (tau[g] > 0.0) {
				Fg_abs_sum += fabs(Fg[g]);
			}
		}
		if ((fabs(F0 / Etot0) < resid_tol) && (cscale * Fg_abs_sum / Etot0 < resid_tol)) {
			break;
		}

		double d_fourpiboverc_d_t[NG];
		thermalRadiationTempDerivativeMG<NG>(r, frac, T_d, d_fourpiboverc_d_t);
		const double c_v = ec.eintTempDerivative(T_gas);
#pragma unroll
		for (int g = 0; g < NG; ++g) {
			const double dEg_dT = ot.kappaPoverE[g] * d_fourpiboverc_d_t[g];
			Jg0[g] = 1.0 / c_v * dEg_dT;
			if (tau[g] <= 0.0) {
				Jgg[g] = -__builtin_inf();
			} else {
				Jgg[g] = -1.0 * ot.kappaPoverE[g] / tau[g] - 1.0;
			}
		}
		const double J00 = 1.0;

		// SolveLinearEqs (J0g = cscale for every group)
		double s1 = 0, s2 = 0;
		double ratio[NG];
#pragma unroll
		for (int g = 0; g < NG; ++g) {
			ratio[g] = cscale / Jgg[g];
		}
#pragma unroll
		for (int g = 0; g < NG; ++g) {
			s1 += ratio[g] * Fg[g];
		}
#pragma unroll
		for (int g = 0; g < NG; ++g) {
			s2 += ratio[g] * Jg0[g];
		}
		const double delta_x = (s1 - F0) / (-s2 + J00);

		const double T_rad = sqrt(sqrt(sumOf<NG>(EradVec_guess) / r.arad));
		if (delta_x / c_v > smax(T_gas, T_rad)) { // enable_dE_constrain
			Egas_guess = eos.eintFromTgas(rho, T_rad);
		} else {
			Egas_guess += delta_x;
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				const double delta_R = (-1.0 * Fg[g] - Jg0[g] * delta_x) / Jgg[g];
				Rvec[g] = Rvec[g] + delta_R;
			}
		}
	}
	if (n >= maxIter) {
		fail_newton += 1;
	}
	n_solves += 1;
	n_newton_total += n + 1;
	n_newton_max = max(n_newton_max, n + 1);

	if (n > 0) {
		kappaFAndDeltaTerms<NG>(r, m, T_d, rho, fourPiBoverC, ot);
	}
	res.Egas = Egas_guess;
	res.T_gas = T_gas;
	res.T_d = T_d;
#pragma unroll
	for (int g = 0; g < NG; ++g) {
		res.EradVec[g] = EradVec_guess[g];
		res.work[g] = work_local[g];
	}
}

// radiation_system.hpp:1420-1483 (nGroups_ > 1), n_step == 0: the dust temperature at which absorption, emission and the gas-dust exchange balance,
// by BackwardEulerOneVariable (:1387-1418)
template <int NG, class M>
QK_DEV auto dustTemperatureBateKetoMG(Rad const &r, M &m, double T_gas, double T_d_init, double rho, const double Erad[NG], double N_d, double dt,
				      const double ratios[NG]) -> double
{
	const double Lambda_compare = N_d * sqrt(T_gas) * T_gas;
	double x = T_d_init;
	const double rel_tol = 1.0e-8;
	const double rel_change_tol = 1.0e-6;
	const int max_iter_td = 100;
	int iter_Td = 0;
	OpacityTermsMG<NG> ot;
	double frac[NG], fourPiBoverC[NG], dB[NG];
	for (; iter_Td < max_iter_td; ++iter_Td) {
#pragma unroll
		for (int g = 0; g < NG; ++g) {
			ot.alpha_E[g] = 0.;
			ot.alpha_P[g] = 0.;
		}
		planckEnergyFractions<NG>(m, x, frac);
		thermalRadiationMG<NG, true>(r, frac, x, fourPiBoverC);
		kappaEAndKappaP<NG>(m, x, rho, ratios, fourPiBoverC, Erad, 0, ot);
		double s = 0;
#pragma unroll
		for (int g = 0; g < NG; ++g) {
			s += ot.kappaE[g] * Erad[g] - ot.kappaP[g] * fourPiBoverC[g];
		}
		const double the_rhs = r.chat * dt * rho * s + N_d * sqrt(T_gas) * (T_gas - x);
		if (fabs(the_rhs) < rel_tol * Lambda_compare) {
			break;
		}
		thermalRadiationTempDerivativeMG<NG, true>(r, frac, x, dB);
		double sj = 0;
#pragma unroll
		for (int g = 0; g < NG; ++g) {
			sj += ot.kappaP[g] * dB[g];
		}
		const double jac = -r.chat * dt * rho * sj - N_d * sqrt(T_gas);
		const double dT = -the_rhs / jac;
		x += dT;
		if (iter_Td > 0) {
			if (fabs(dT) < rel_change_tol * fabs(x)) {
				break;
			}
		}
	}
	if (iter_Td >= max_iter_td) {
		x = -1.0;
	}
	return x;
}

// radiation_dust_system.hpp:228-576 (+ ComputeJacobianForGasAndDust :22-83, ...Decoupled :85-128, SolveLinearEqs in line; net cooling and cosmic-ray
// heating are the defaults, zero).  dust_model 1: gas, dust and the groups in one system; dust_model 2 (weak gas-dust exchange): the dust temperature and
// the groups are iterated with the exchange rate frozen, the gas energy follows from it afterwards.
template <int NG, class M>
QK_DEV void solveGasDustRadiationEnergyExchange(Rad const &r, M &m, Eos const &eos, double Egas0, const double Erad0Vec[NG], double rho, double coeff_n,
						double dt, int n_outer_iter, const double work[NG], const double vel_times_F[NG], const double Src[NG],
						NewtonResultMG<NG> &res, int &n_newton_total, int &n_newton_max, int &n_solves, int &n_decoupled, int &fail_newton,
						int &fail_dust)
{
	const double c = r.c;
	const double chat = r.chat;
	const double cscale = c / chat;
	const EosCell ec(eos, rho);

	double ratios[NG], frac[NG];
#pragma unroll
	for (int g = 0; g < NG; ++g) {
		ratios[g] = m.bnd[g + 1] / m.bnd[g];
	}
	int dust_model = 1;
	double lambda_gd_times_dt = __builtin_nan("");
	const double T_gas0 = ec.tgasFromEint(Egas0);
	const double T_d0 = dustTemperatureBateKetoMG<NG>(r, m, T_gas0, T_gas0, rho, Erad0Vec, coeff_n, dt, ratios);
	if (T_d0 < 0.0) {
		fail_dust += 1;
	}
	const double max_Gamma_gd = coeff_n * fmax(sqrt(T_gas0) * T_gas0, sqrt(T_d0) * T_d0);
	if (cscale * max_Gamma_gd < r.dust_threshold * Egas0) {
		dust_model = 2;
		lambda_gd_times_dt = coeff_n * sqrt(T_gas0) * (T_gas0 - T_d0);
	}
	double Etot0;
	if (dust_model == 1) {
		Etot0 = Egas0 + cscale * (sumOf<NG>(Erad0Vec) + sumOf<NG>(Src));
	} else if (r.pe_on == 0) {
		double B0[NG];
		planckEnergyFractions<NG>(m, T_d0, frac);
		thermalRadiationMG<NG, true>(r, frac, T_d0, B0);
		Etot0 = fabs(lambda_gd_times_dt) + sumOf<NG>(B0) + (sumOf<NG>(Erad0Vec) + sumOf<NG>(Src));
	} else { // ...WithPE (:631)
		Etot0 = fabs(lambda_gd_times_dt) + (sumOf<NG>(Erad0Vec) + sumOf<NG>(Src));
	}

	double T_gas = __builtin_nan(""), T_d = __builtin_nan("");
	double Rvec[NG], tau[NG], work_local[NG], fourPiBoverC[NG];
	OpacityTermsMG<NG> &ot = res.ot;
#pragma unroll
	for (int g = 0; g < NG; ++g) {
		ot.alpha_E[g] = 0.;
		ot.alpha_P[g] = 0.;
		Rvec[g] = 0.;
		tau[g] = 0.;
		work_local[g] = 0.;
	}
	double Egas_guess = Egas0;
	double EradVec_guess[NG];
#pragma unroll
	for (int g = 0; g < NG; ++g) {
		EradVec_guess[g] = Erad0Vec[g];
	}
	T_gas = T_gas0;
	// ...WithPE (:683-685): the photoelectric heating rate per unit FUV energy density, evaluated once
	const bool with_PE = (r.pe_on != 0);
	const double PE_heating_energy_derivative = with_PE ? dt * r.pe_rate : 0.0;
	const double CR_heating = r.cr_heat * dt;

	const double resid_tol = 1.0e-11;
	const int maxIter = 100;
	int n = 0;
	for (; n < maxIter; ++n) {
		if (n > 0) {
			T_gas = ec.tgasFromEint(Egas_guess);
		}
		if (dust_model == 1) {
			if (n == 0) {
				T_d = T_d0;
			} else {
				T_d = T_gas - sumOf<NG>(Rvec) / (coeff_n * sqrt(T_gas));
			}
		} else {
			if (n == 0) {
				T_d = T_d0;
			}
		}
		if (T_d < 0.0) {
			fail_dust += 1;
		}
		planckEnergyFractions<NG>(m, T_d, frac);
		thermalRadiationMG<NG, true>(r, frac, T_d, fourPiBoverC);
		kappaEAndKappaP<NG>(m, T_d, rho, ratios, fourPiBoverC, EradVec_guess, n, ot);
		if (n == 0) {
			kappaFAndDeltaTerms<NG>(r, m, T_d, rho, fourPiBoverC, ot);
			if (r.beta_order == 1) { // include_work_term_in_source
				if (n_outer_iter == 0) {
#pragma unroll
					for (int g = 0; g < NG; ++g) {
						if (m.model == MG_PIECEWISE_CONSTANT) {
							work_local[g] = vel_times_F[g] * ot.kappaF[g] * chat / (c * c) * dt;
						} else {
							work_local[g] = vel_times_F[g] * ot.kappaF[g] * chat / (c * c) * dt * (1.0 + m.kexp[g]);
						}
					}
				} else {
#pragma unroll
					for (int g = 0; g < NG; ++g) {
						work_local[g] = work[g];
					}
				}
			} else {
#pragma unroll
				for (int g = 0; g < NG; ++g) {
					work_local[g] = 0.0;
				}
			}
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				const double tau0 = dt * rho * ot.kappaP[g] * chat;
				tau[g] = tau0;
				Rvec[g] = (fourPiBoverC[g] - EradVec_guess[g] / ot.kappaPoverE[g]) * tau0 + work_local[g];
			}
		} else {
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				tau[g] = dt * rho * ot.kappaP[g] * chat;
				if (tau[g] > 0.0) {
					EradVec_guess[g] = ot.kappaPoverE[g] * (fourPiBoverC[g] - (Rvec[g] - work_local[g]) / tau[g]);
				}
			}
		}

		double d_fourpiboverc_d_t[NG];
		thermalRadiationTempDerivativeMG<NG, true>(r, frac, T_d, d_fourpiboverc_d_t);
		const double c_v = ec.eintTempDerivative(T_gas);
		const double Egas_diff = Egas_guess - Egas0;

		double F0, J00;
		double J0g[NG], Jg1[NG];
		double Fg[NG], Jg0[NG], Jgg[NG];
		double Fg_abs_sum = 0.0;
		if (dust_model == 1) { // ComputeJacobianForGasAndDust (:22-83) / ...WithPE (:130-196)
			double cooling[NG], cooling_derivative[NG];
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				cooling[g] = (m.cool[g] * T_gas) * dt;
				cooling_derivative[g] = m.cool[g] * dt;
			}
			if (with_PE) {
				F0 = Egas_diff + cscale * sumOf<NG>(Rvec) + sumOf<NG>(cooling) - PE_heating_energy_derivative * EradVec_guess[NG - 1] - CR_heating;
			} else {
				F0 = Egas_diff + cscale * sumOf<NG>(Rvec) + sumOf<NG>(cooling) - CR_heating;
			}
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				const double Erad_diff = EradVec_guess[g] - Erad0Vec[g];
				Fg[g] = Erad_diff - (Rvec[g] + Src[g]);
				if (tau[g] > 0.0) {
					Fg_abs_sum += fabs(Fg[g]);
				} else {
					Fg_abs_sum += fabs(Fg[g] + Rvec[g]);
				}
			}
			J00 = 1.0 + sumOf<NG>(cooling_derivative) / c_v;
			const double d_Td_d_T = 3. / 2. - T_d / (2. * T_gas);
			const double dTd_dRg = -1.0 / (coeff_n * sqrt(T_gas));
			double rg[NG], d_Eg_d_Rg[NG];
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				J0g[g] = cscale;
				rg[g] = ot.kappaPoverE[g] * d_fourpiboverc_d_t[g] * dTd_dRg;
				if (with_PE) {
					d_Eg_d_Rg[g] = -1.0 * ot.kappaPoverE[g];
					if (tau[g] <= 0.0) {
						d_Eg_d_Rg[g] = -1.0e100; // LARGE
					} else {
						d_Eg_d_Rg[g] /= tau[g];
					}
				}
			}
			if (with_PE) {
				J0g[NG - 1] -= PE_heating_energy_derivative * d_Eg_d_Rg[NG - 1];
			}
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				const double dEg_dT = ot.kappaPoverE[g] * d_fourpiboverc_d_t[g] * d_Td_d_T;
				Jg0[g] = 1.0 / c_v * dEg_dT - (1 / cscale) * cooling_derivative[g] - 1.0 / cscale * rg[g] * J00;
				Fg[g] = Fg[g] - 1.0 / cscale * rg[g] * F0;
				if (with_PE) {
					Jgg[g] = d_Eg_d_Rg[g] + (-1.0);
					Jg1[g] = rg[g] - 1.0 / cscale * rg[g] * J0g[NG - 1];
				} else {
					Jg1[g] = 0.0;
					if (tau[g] <= 0.0) {
						Jgg[g] = -__builtin_inf();
					} else {
						Jgg[g] = -1.0 * ot.kappaPoverE[g] / tau[g] - 1.0;
					}
				}
			}
			if (with_PE) {
				Jgg[NG - 1] += rg[NG - 1] - (rg[NG - 1] / cscale) * PE_heating_energy_derivative * d_Eg_d_Rg[NG - 1];
			}
		} else { // ComputeJacobianForGasAndDustDecoupled (:85-128)
			F0 = -lambda_gd_times_dt + sumOf<NG>(Rvec);
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				const double Erad_diff = EradVec_guess[g] - Erad0Vec[g];
				Fg[g] = Erad_diff - (Rvec[g] + Src[g]);
				if (tau[g] > 0.0) {
					Fg_abs_sum += fabs(Fg[g]);
				}
				Jg0[g] = ot.kappaPoverE[g] * d_fourpiboverc_d_t[g];
				J0g[g] = 1.0;
				Jg1[g] = 0.0;
				if (tau[g] <= 0.0) {
					Jgg[g] = -__builtin_inf();
				} else {
					Jgg[g] = -1.0 * ot.kappaPoverE[g] / tau[g] - 1.0;
				}
			}
			J00 = 0.0;
		}

		if ((fabs(F0 / Etot0) < resid_tol) && (cscale * Fg_abs_sum / Etot0 < resid_tol)) {
			break;
		}

		double delta_x;
		double delta_R[NG];
		double ratio[NG];
#pragma unroll
		for (int g = 0; g < NG; ++g) {
			ratio[g] = J0g[g] / Jgg[g];
		}
		if (with_PE) { // SolveLinearEqsWithLastColumn (:198-226)
			constexpr int pe = NG - 1;
			double sa = 0, sy = 0, s1 = 0;
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				sa += ratio[g] * Jg0[g];
			}
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				sy += ratio[g] * Fg[g];
			}
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				s1 += ratio[g] * Jg1[g];
			}
			const double a00_new = J00 - sa;
			const double y0_new = F0 - sy;
			double a01_new = J0g[pe] - s1;
			a01_new = a01_new + ratio[pe] * Jg1[pe] - ratio[pe] * Jgg[pe];
			const double a10 = Jg0[pe];
			const double a11 = Jgg[pe];
			const double y1 = Fg[pe];
			double x0 = (y0_new - a01_new / a11 * y1) / (a00_new - a01_new / a11 * a10);
			const double x1 = (y1 - a10 * x0) / a11;
			delta_R[pe] = x1;
#pragma unroll
			for (int g = 0; g < pe; ++g) {
				delta_R[g] = (Fg[g] - Jg0[g] * x0 - Jg1[g] * x1) / Jgg[g];
			}
			x0 *= -1.0;
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				delta_R[g] = delta_R[g] * -1.0;
			}
			delta_x = x0;
		} else { // SolveLinearEqs (radiation_system.hpp:547-558)
			double s1 = 0, s2 = 0;
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				s1 += ratio[g] * Fg[g];
			}
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				s2 += ratio[g] * Jg0[g];
			}
			delta_x = (s1 - F0) / (-s2 + J00);
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				delta_R[g] = (-1.0 * Fg[g] - Jg0[g] * delta_x) / Jgg[g];
			}
		}

		if (dust_model == 2) {
			T_d += delta_x;
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				Rvec[g] = Rvec[g] + delta_R[g];
			}
		} else {
			const double T_rad = sqrt(sqrt(sumOf<NG>(EradVec_guess) / r.arad));
			if (delta_x / c_v > smax(T_gas, T_rad)) { // enable_dE_constrain
				Egas_guess = eos.eintFromTgas(rho, T_rad);
			} else {
				Egas_guess += delta_x;
#pragma unroll
				for (int g = 0; g < NG; ++g) {
					Rvec[g] = Rvec[g] + delta_R[g];
				}
			}
		}
	}

	// :515-543 (PE :867-897): the line-cooling tendency at the last gas temperature; decoupled: the gas energy from a scalar backward-Euler solve of
	// E - E0 + cscale lambda dt + sum(cooling(T(E))) dt [- PE E_FUV] - CR = 0
	double cooling_tend[NG];
#pragma unroll
	for (int g = 0; g < NG; ++g) {
		cooling_tend[g] = (m.cool[g] * T_gas) * dt;
	}
	if (dust_model == 2) {
		double sabs = 0;
#pragma unroll
		for (int g = 0; g < NG; ++g) {
			sabs += fabs(cooling_tend[g]);
		}
		const double compare = Egas_guess + cscale * lambda_gd_times_dt + sabs + CR_heating;
		const double pe_term = PE_heating_energy_derivative * EradVec_guess[NG - 1];
		double x = Egas0;
		const int max_iter_td = 100;
		int it = 0;
		for (; it < max_iter_td; ++it) {
			const double T_gas_ = ec.tgasFromEint(x);
			double sc = 0, sd = 0;
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				sc += (m.cool[g] * T_gas_) * dt;
			}
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				sd += m.cool[g] * dt;
			}
			double the_rhs;
			if (with_PE) {
				the_rhs = x - Egas0 + cscale * lambda_gd_times_dt + sc - pe_term - CR_heating;
			} else {
				the_rhs = x - Egas0 + cscale * lambda_gd_times_dt + sc - CR_heating;
			}
			if (fabs(the_rhs) < 1.0e-8 * compare) {
				break;
			}
			const double dT = -the_rhs / (1.0 + sd);
			x += dT;
			if (it > 0) {
				if (fabs(dT) < 1.0e-6 * fabs(x)) {
					break;
				}
			}
		}
		if (it >= max_iter_td) {
			x = -1.0;
		}
		Egas_guess = x;
	}
#pragma unroll
	for (int g = 0; g < NG; ++g) {
		EradVec_guess[g] = EradVec_guess[g] + (1 / cscale) * cooling_tend[g];
	}

	if (n >= maxIter) {
		fail_newton += 1;
	}
	n_solves += 1;
	n_newton_total += n + 1;
	n_newton_max = max(n_newton_max, n + 1);
	if (dust_model == 2) {
		n_decoupled += 1;
	}

	if (n > 0) {
		kappaFAndDeltaTerms<NG>(r, m, T_d, rho, fourPiBoverC, ot);
	}
	res.Egas = Egas_guess;
	res.T_gas = T_gas;
	res.T_d = T_d;
#pragma unroll
	for (int g = 0; g < NG; ++g) {
		res.EradVec[g] = EradVec_guess[g];
		res.work[g] = work_local[g];
	}
}

// source_terms_multi_group.hpp:522-813 for one cell (UpdateFlux :360-520 in line).  U[6 + 4 NG] in place; counters as in the reference.
// DUST: ISM_Traits::enable_dust_gas_thermal_coupling_model (its own instantiation, as in the single-group kernel)
template <int NG, bool DUST = false, class M>
QK_DEV void radSourceCellMG(Rad const &r, M &m, Eos const &eos, double U[RAD0 + NRAD * NG], const double srcval[NG], double dt_radiation, int stage,
			    int &n_newton_total, int &n_newton_max, int &n_solves, int &fail_newton, int &fail_outer, int *n_decoupled = nullptr,
			    int *fail_dust = nullptr)
{
	double dt = dt_radiation;
	if (stage == 2) {
		dt = (1.0 - IMEX_a32) * dt_radiation;
	}
	const double c = r.c;
	const double chat = r.chat;
	const bool gamma_ne_1 = !eos.isothermal;
	const int beta_order = r.beta_order;
	const double rho = U[RHO];
	const double x1GasMom0 = U[MX], x2GasMom0 = U[MY], x3GasMom0 = U[MZ];
	const double gasMtm0[3] = {x1GasMom0, x2GasMom0, x3GasMom0};
	const double Egastot0 = U[ENE];

	double Erad0Vec[NG], Src[NG];
#pragma unroll
	for (int g = 0; g < NG; ++g) {
		Erad0Vec[g] = U[RAD0 + NRAD * g];
		Src[g] = dt * (chat * srcval[g]);
	}
	double Egas0 = __builtin_nan(""), Ekin0 = __builtin_nan(""), Egas_guess = __builtin_nan("");
	double work[NG], work_prev[NG];
#pragma unroll
	for (int g = 0; g < NG; ++g) {
		work[g] = 0.;
		work_prev[g] = 0.;
	}
	if (gamma_ne_1) {
		Egas0 = eintFromEgas(rho, x1GasMom0, x2GasMom0, x3GasMom0, Egastot0);
		Ekin0 = Egastot0 - Egas0;
	}
	double gas_update_factor = 1.0;
	if (stage == 1) {
		gas_update_factor = IMEX_a32;
	}
	const Recip Rcc = recipOf(c * chat);

	// the converged outer iteration stores these (:760-775); an unconverged cell keeps its momenta and radiation state (:787-790)
	double newMom[3] = {U[MX], U[MY], U[MZ]};
	NewtonResultMG<NG> en;

	const int max_iter = 5;
	int iter = 0;
	for (; iter < max_iter; ++iter) {
		if (gamma_ne_1) {
			double vel_times_F[NG];
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				vel_times_F[g] = 0.;
				if (iter == 0) {
					vel_times_F[g] = (x1GasMom0 * U[RAD0 + NRAD * g + 1] + x2GasMom0 * U[RAD0 + NRAD * g + 2] + x3GasMom0 * U[RAD0 + NRAD * g + 3]);
				}
			}
			if constexpr (DUST) { // :612-617, :706-723
				const double H_num_den = rho / r.mean_molecular_mass;
				const double coeff_n = dt * r.dust_coeff * H_num_den * H_num_den / (c / chat);
				solveGasDustRadiationEnergyExchange<NG>(r, m, eos, Egas0, Erad0Vec, rho, coeff_n, dt, iter, work, vel_times_F, Src, en, n_newton_total,
									n_newton_max, n_solves, *n_decoupled, fail_newton, *fail_dust);
			} else {
				solveGasRadiationEnergyExchange<NG>(r, m, eos, Egas0, Erad0Vec, rho, dt, iter, work, vel_times_F, Src, en, n_newton_total, n_newton_max,
								    n_solves, fail_newton);
			}
			Egas_guess = en.Egas;
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				work_prev[g] = en.work[g];
			}
		} else {
			double lower[NG + 1], ratios[NG], am1[NG];
			m.at(rho, __builtin_nan("")); // :733
#pragma unroll
			for (int g = 0; g < NG + 1; ++g) {
				lower[g] = m.lower(g, rho, __builtin_nan(""));
			}
			if (m.model == MG_PIECEWISE_CONSTANT) {
#pragma unroll
				for (int g = 0; g < NG; ++g) {
					en.ot.kappaF[g] = lower[g];
				}
			} else {
#pragma unroll
				for (int g = 0; g < NG; ++g) {
					ratios[g] = m.bnd[g + 1] / m.bnd[g];
					am1[g] = -1.0;
				}
				groupMeanOpacity<NG>(m, lower, ratios, am1, en.ot.kappaF);
			}
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				en.EradVec[g] = Erad0Vec[g];
				en.work[g] = 0.;
			}
			en.T_d = __builtin_nan("");
			en.Egas = __builtin_nan("");
		}

		// UpdateFlux
		double dMomentum[3] = {0., 0., 0.};
		double Frad_t1[3][NG];
		if (!gamma_ne_1 || beta_order == 0) {
#pragma unroll
			for (int g = 0; g < NG; ++g) {
#pragma unroll
				for (int n = 0; n < 3; ++n) {
					const double Frad_t0 = U[RAD0 + NRAD * g + 1 + n];
					Frad_t1[n][g] = Frad_t0 / (1.0 + rho * en.ot.kappaF[g] * chat * dt);
					dMomentum[n] += divBy(-(Frad_t1[n][g] - Frad_t0), Rcc);
				}
			}
		} else {
			double frac[NG], fourPiBoverC[NG];
			m.at(rho, en.T_d); // :428 (the exponents of the pressure and work terms below)
			planckEnergyFractions<NG>(m, en.T_d, frac);
			thermalRadiationMG<NG, DUST>(r, frac, en.T_d, fourPiBoverC);
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				const double Frad_t0[3] = {U[RAD0 + NRAD * g + 1], U[RAD0 + NRAD * g + 2], U[RAD0 + NRAD * g + 3]};
				const double erad = en.EradVec[g];
				double v_terms[3];
				const Recip RcE = recipOf(r.c * erad);
				const double fx = divBy(Frad_t0[0], RcE);
				const double fy = divBy(Frad_t0[1], RcE);
				const double fz = divBy(Frad_t0[2], RcE);
				const double F_coeff = chat * rho * en.ot.kappaF[g] * dt;
				double Tedd[3][3];
				eddingtonTensor(r, fx, fy, fz, Tedd);
#pragma unroll
				for (int n = 0; n < 3; ++n) {
					double Planck_term = en.ot.kappaP[g] * fourPiBoverC[g] - 1.0 / 3.0 * en.ot.delta_nu_kappa_B_at_edge[g]; // include_delta_B
					Planck_term *= chat * dt * gasMtm0[n];
					double pressure_term = 0.0;
#pragma unroll
					for (int z = 0; z < 3; ++z) {
						pressure_term += gasMtm0[z] * Tedd[n][z] * erad;
					}
					if (m.model == MG_PIECEWISE_CONSTANT) {
						pressure_term *= chat * dt * en.ot.kappaE[g];
					} else {
						pressure_term *= chat * dt * (1.0 + m.kexp[g]) * en.ot.kappaE[g];
					}
					v_terms[n] = Planck_term + pressure_term;
				}
				const Recip R1F = recipOf(1.0 + F_coeff);
#pragma unroll
				for (int n = 0; n < 3; ++n) {
					Frad_t1[n][g] = divBy(Frad_t0[n] + v_terms[n], R1F);
					dMomentum[n] += divBy(-(Frad_t1[n][g] - Frad_t0[n]), Rcc);
				}
			}
		}
		double x1GasMom1 = U[MX] + dMomentum[0];
		double x2GasMom1 = U[MY] + dMomentum[1];
		double x3GasMom1 = U[MZ] + dMomentum[2];

		if (gamma_ne_1 && beta_order == 1) {
			const double Egastot1 = egasFromEint(rho, x1GasMom1, x2GasMom1, x3GasMom1, en.Egas);
			const double Ekin1 = Egastot1 - en.Egas;
			const double dEkin_work = Ekin1 - Ekin0;
			en.Egas -= dEkin_work;
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				if (m.model == MG_PIECEWISE_CONSTANT) {
					en.work[g] = (x1GasMom1 * Frad_t1[0][g] + x2GasMom1 * Frad_t1[1][g] + x3GasMom1 * Frad_t1[2][g]) * en.ot.kappaF[g] * chat / (c * c) * dt;
				} else {
					en.work[g] = (x1GasMom1 * Frad_t1[0][g] + x2GasMom1 * Frad_t1[1][g] + x3GasMom1 * Frad_t1[2][g]) * (1.0 + m.kexp[g]) * en.ot.kappaF[g] *
						     chat / (c * c) * dt;
				}
			}
		}
		x1GasMom1 = U[MX] + dMomentum[0] * gas_update_factor;
		x2GasMom1 = U[MY] + dMomentum[1] * gas_update_factor;
		x3GasMom1 = U[MZ] + dMomentum[2] * gas_update_factor;

		bool work_converged = true;
		if (!((beta_order == 0) || !gamma_ne_1)) {
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				work[g] = en.work[g];
			}
			const double Egastot1 = egasFromEint(rho, x1GasMom1, x2GasMom1, x3GasMom1, Egas_guess);
			const double rel_lag_tol = 1.0e-8;
			const double lag_tol = 1.0e-13;
			double sabs = 0, sdiff = 0;
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				sabs += fabs(work[g]);
			}
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				sdiff += fabs(work[g] - work_prev[g]);
			}
			double ref_work = rel_lag_tol * sabs;
			ref_work = smax(ref_work, lag_tol * Egastot1 / (c / chat));
			if (sdiff > ref_work) {
				work_converged = false;
			}
		}
		if (work_converged) {
			newMom[0] = x1GasMom1;
			newMom[1] = x2GasMom1;
			newMom[2] = x3GasMom1;
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				U[RAD0 + NRAD * g] = en.EradVec[g];
				U[RAD0 + NRAD * g + 1] = Frad_t1[0][g];
				U[RAD0 + NRAD * g + 2] = Frad_t1[1][g];
				U[RAD0 + NRAD * g + 3] = Frad_t1[2][g];
			}
			if (gamma_ne_1) {
				Egas_guess = en.Egas;
			}
			break;
		}
	}
	if (iter >= max_iter) {
		fail_outer += 1;
	}
	U[MX] = newMom[0];
	U[MY] = newMom[1];
	U[MZ] = newMom[2];
	if (gamma_ne_1) {
		Egas_guess = Egas0 + (Egas_guess - Egas0) * gas_update_factor;
		U[EINT] = Egas_guess;
		U[ENE] = egasFromEint(rho, newMom[0], newMom[1], newMom[2], Egas_guess);
	}
}

} // namespace qk

#endif // QK_RAD_MG_DEVICE_HPP_
