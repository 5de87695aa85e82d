// ORACLE — TEST INFRASTRUCTURE ONLY (see grid.hpp header).
//
// radiation_multigroup.hpp: restatement of the multigroup matter-radiation exchange
//   reference src/radiation/source_terms_multi_group.hpp   (ComputeModelDependentKappaEAndKappaP, ComputeModelDependentKappaFAndDeltaTerms,
//                                                           ComputeJacobianForGas, SolveGasRadiationEnergyExchange, UpdateFlux,
//                                                           AddSourceTermsMultiGroup)
//   reference src/radiation/radiation_system.hpp           (ComputePlanckEnergyFractions :430-461, ComputeThermalRadiationMultiGroup :483-497,
//                                                           ComputeThermalRadiationTempDerivativeMultiGroup :505-513, SolveLinearEqs :547-558,
//                                                           ComputeRadQuantityExponents :1169-1250, ComputeGroupMeanOpacity :1252-1287,
//                                                           PlanckFunction :1311-1326, ComputeDiffusionFluxMeanOpacity :1328-1352)
//   reference src/radiation/planck_integral.hpp            (interpolate_planck_integral, integrate_planck_from_0_to_x)
//   reference src/radiation/radiation_dust_system.hpp      (ComputeJacobianForGasAndDust :22-83, ...Decoupled :85-128, SolveGasDustRadiationEnergyExchange :228-576;
//                                                           ComputeDustTemperatureBateKeto, radiation_system.hpp:1420-1483, nGroups > 1 branch)
// Gas + radiation, and gas + dust + radiation (ISM_Traits::enable_dust_gas_thermal_coupling_model).  No photoelectric heating / line cooling /
// cosmic-ray heating (their hooks default to zero and are written as zeros; SolveGasDustRadiationEnergyExchangeWithPE is not restated).
//
// The 1000-point table of the incomplete Planck integral is computed by the oracle itself (planck_table.hpp: series in 113-bit arithmetic, each sample
// rounded once to double); the product ships the same function as a data file computed by another program with another method
// (tools/make_planck_table.py: quadrature at 50 digits) — tests/test_multigroup_oracle.py holds the two equal in every bit.  The reference lists the
// function to 15 digits; it agrees with both to <= 5e-14 relative, which is the size of the difference this restatement can show against the
// reference itself in the Planck fractions.
#ifndef ORACLE_RADIATION_MULTIGROUP_HPP_
#define ORACLE_RADIATION_MULTIGROUP_HPP_

#include <array>
#include <cassert>
#include <cmath>
#include <limits>

#include "planck_table.hpp"
#include "radiation.hpp"

namespace oracle
{

// ------------------------------------------------------------------ planck_integral.hpp
namespace planck
{
constexpr int INTERP_SIZE = 1000;
constexpr double LOG_X_MIN = -3.;
constexpr double LOG_X_MAX = 2.;
// (computed by the oracle itself when the library loads — planck_table.hpp: series in 113-bit arithmetic — NOT read from the product's data file)
inline const std::array<double, INTERP_SIZE> Y_interp = computePlanckTable();
constexpr double PI = M_PI;
constexpr double gInf = PI * PI * PI * PI / 15.0;

// planck_integral.hpp:22-232 (USE_SECOND_ORDER = false: linear interpolation in log10 x)
inline auto interpolate_planck_integral(double logx) -> double
{
	const int arr_len = INTERP_SIZE;
	const int j = static_cast<int>((logx - LOG_X_MIN) / (LOG_X_MAX - LOG_X_MIN) * (arr_len - 1));
	const double gap = (LOG_X_MAX - LOG_X_MIN) / (arr_len - 1);
	if (j < 0) {
		return 0.0;
	}
	if (j >= arr_len - 1) {
		return 1.0;
	}
	const double slope = (Y_interp[j + 1] - Y_interp[j]) / gap;
	return slope * (logx - (LOG_X_MIN + j * gap)) + Y_interp[j];
}

// planck_integral.hpp:234-262
inline auto integrate_planck_from_0_to_x(const double x) -> double
{
	if (x <= 0.) {
		return 0.;
	}
	const double Y_INTERP_MIN = Y_interp[0]; // "= Y_interp[0]" (:20)
	const double logx = std::log10(x);
	double y = NAN;
	if (logx < LOG_X_MIN) {
		y = (-4 + x) * x + 8 * std::log((2 + x) / 2); // 2nd order
		if (y > Y_INTERP_MIN) {
			y = Y_INTERP_MIN;
		} else if (y < 0.) {
			y = 0.;
		}
	} else if (logx >= LOG_X_MAX) {
		return 1.0;
	} else {
		y = interpolate_planck_integral(logx);
	}
	return y;
}
} // namespace planck

namespace mg
{

// quokka::valarray<double, nGroups_> / amrex::GpuArray<double, nGroups_> (src/util/valarray.hpp): element-wise arithmetic, sum() accumulates
// from 0 in index order (:241-247)
struct VA {
	std::array<double, kMaxGroups + 1> v{};
	int n = 0;
	VA() = default;
	explicit VA(int n_) : n(n_) {}
	auto operator[](int i) -> double & { return v[i]; }
	auto operator[](int i) const -> double { return v[i]; }
	void fillin(double s)
	{
		for (int i = 0; i < n; ++i) {
			v[i] = s;
		}
	}
};
#define ORACLE_VA_OP(OP)                                                                                                                              \
	inline auto operator OP(VA const &a, VA const &b) -> VA                                                                                       \
	{                                                                                                                                             \
		VA r(a.n);                                                                                                                            \
		for (int i = 0; i < a.n; ++i) {                                                                                                       \
			r[i] = a[i] OP b[i];                                                                                                          \
		}                                                                                                                                     \
		return r;                                                                                                                             \
	}                                                                                                                                             \
	inline auto operator OP(VA const &a, double s) -> VA                                                                                          \
	{                                                                                                                                             \
		VA r(a.n);                                                                                                                            \
		for (int i = 0; i < a.n; ++i) {                                                                                                       \
			r[i] = a[i] OP s;                                                                                                             \
		}                                                                                                                                     \
		return r;                                                                                                                             \
	}                                                                                                                                             \
	inline auto operator OP(double s, VA const &a) -> VA                                                                                          \
	{                                                                                                                                             \
		VA r(a.n);                                                                                                                            \
		for (int i = 0; i < a.n; ++i) {                                                                                                       \
			r[i] = s OP a[i];                                                                                                             \
		}                                                                                                                                     \
		return r;                                                                                                                             \
	}
ORACLE_VA_OP(+)
ORACLE_VA_OP(-)
ORACLE_VA_OP(*)
ORACLE_VA_OP(/)
#undef ORACLE_VA_OP
inline auto sum(VA const &a) -> double
{
	double s = 0;
	for (int i = 0; i < a.n; ++i) {
		s += a[i];
	}
	return s;
}
inline auto abs(VA const &a) -> VA
{
	VA r(a.n);
	for (int i = 0; i < a.n; ++i) {
		r[i] = std::abs(a[i]);
	}
	return r;
}

// radiation_system.hpp:36-45, :61
constexpr bool include_delta_B = true;
constexpr bool use_diffuse_flux_mean_opacity = true;
constexpr bool special_edge_bin_slopes = false;
constexpr int max_iter_to_update_alpha_E = 5;
constexpr bool use_D_as_base = false;
constexpr bool PPL_free_slope_st_total = false;
constexpr double inf = std::numeric_limits<double>::max();

// src/math/math_impl.hpp:18, radiation_system.hpp:142-145
inline auto sgn(double val) -> int { return static_cast<int>(0. < val) - static_cast<int>(val < 0.); }
inline auto minmod_func(double a, double b) -> double { return 0.5 * (sgn(a) + sgn(b)) * std::min(std::abs(a), std::abs(b)); }

// kappa_expo_and_lower_value: [0] exponents, [1] lower values, nGroups + 1 entries each
struct KappaExpoLower {
	VA expo, lower;
};

// radiation_system.hpp:100-108 / :112-119 / :123-132 / :136-140
struct OpacityTerms {
	VA kappaE, kappaP, kappaF, kappaPoverE;
	VA delta_nu_kappa_B_at_edge;
	VA alpha_P, alpha_E;
};
struct NewtonIterationResult {
	double Egas = NAN, T_gas = NAN, T_d = NAN;
	VA EradVec, work;
	OpacityTerms opacity_terms;
};
struct JacobianResult {
	double J00 = NAN, F0 = NAN, Fg_abs_sum = NAN;
	VA J0g, Jg0, Jgg, Fg;
	VA Jg1; // photoelectric heating: the column of the last (FUV) group
};
struct FluxUpdateResult {
	VA Erad;
	std::array<double, 3> gasMomentum{};
	std::array<VA, 3> Frad;
};

struct MG {
	RadSystem const &rs;
	int nGroups_;
	explicit MG(RadSystem const &r) : rs(r), nGroups_(r.rt.nGroups) {}

	// DefineNetCoolingRate / ...TempDerivative (radiation_system.hpp:348-351; defaults :524-540: zero)
	[[nodiscard]] auto netCoolingRate(double T, double num_den) const -> VA
	{
		VA v(nGroups_);
		if (rs.DefineNetCoolingRate) {
			rs.DefineNetCoolingRate(T, num_den, v.v.data());
		}
		return v;
	}
	[[nodiscard]] auto netCoolingRateTempDerivative(double T, double num_den) const -> VA
	{
		VA v(nGroups_);
		if (rs.DefineNetCoolingRateTempDerivative) {
			rs.DefineNetCoolingRateTempDerivative(T, num_den, v.v.data());
		}
		return v;
	}

	[[nodiscard]] auto boundaries() const -> VA
	{
		VA b(nGroups_ + 1);
		for (int g = 0; g < nGroups_ + 1; ++g) {
			b[g] = rs.rt.radBoundaries[g];
		}
		return b;
	}

	[[nodiscard]] auto DefineOpacityExponentsAndLowerValues(VA const &rad_boundaries, double rho, double Tgas) const -> KappaExpoLower
	{
		KappaExpoLower r;
		r.expo = VA(nGroups_ + 1);
		r.lower = VA(nGroups_ + 1);
		r.expo.fillin(NAN); // the default hook (radiation_system.hpp:1155-1167)
		r.lower.fillin(NAN);
		if (rs.DefineOpacityExponentsAndLowerValues) {
			rs.DefineOpacityExponentsAndLowerValues(rad_boundaries.v.data(), rho, Tgas, r.expo.v.data(), r.lower.v.data());
		}
		return r;
	}

	// radiation_system.hpp:430-461
	[[nodiscard]] auto ComputePlanckEnergyFractions(VA const &bnd, double temperature) const -> VA
	{
		VA radEnergyFractions(nGroups_);
		if (nGroups_ == 1) {
			radEnergyFractions[0] = 1.0;
			return radEnergyFractions;
		}
		double const energy_unit_over_kT = rs.rt.energy_unit / (rs.eos.tr.boltzmann_constant * temperature);
		double y = NAN;
		double previous = 0.0;
		for (int g = 0; g < nGroups_ - 1; ++g) {
			const double x = bnd[g + 1] * energy_unit_over_kT;
			if (x >= 100.) {
				y = 1.0;
			} else {
				y = planck::integrate_planck_from_0_to_x(x);
			}
			radEnergyFractions[g] = y - previous;
			previous = y;
		}
		y = 1.0;
		radEnergyFractions[nGroups_ - 1] = y - previous;
		return radEnergyFractions;
	}

	// :483-497
	[[nodiscard]] auto ComputeThermalRadiationMultiGroup(double temperature, VA const &bnd) const -> VA
	{
		if (rs.rt.thermal_model == 1) { // RadDustMG's specialisation (src/problems/RadDustMG/test_rad_dust_MG.cpp:83-93): a T, no floor
			auto radEnergyFractions = ComputePlanckEnergyFractions(bnd, temperature);
			const double power = rs.rt.radiation_constant * temperature;
			return power * radEnergyFractions;
		}
		const double power = rs.rt.radiation_constant * rs.pow4(temperature);
		const auto radEnergyFractions = ComputePlanckEnergyFractions(bnd, temperature);
		auto Erad_g = power * radEnergyFractions;
		for (int g = 0; g < nGroups_; ++g) {
			if (Erad_g[g] < rs.Erad_floor_()) {
				Erad_g[g] = rs.Erad_floor_();
			}
		}
		return Erad_g;
	}

	// :505-513
	[[nodiscard]] auto ComputeThermalRadiationTempDerivativeMultiGroup(double temperature, VA const &bnd) const -> VA
	{
		auto radEnergyFractions = ComputePlanckEnergyFractions(bnd, temperature);
		if (rs.rt.thermal_model == 1) { // test_rad_dust_MG.cpp:95-104
			const double d_power_dt = rs.rt.radiation_constant;
			return d_power_dt * radEnergyFractions;
		}
		double d_power_dt = 4. * rs.rt.radiation_constant * rs.pow3(temperature);
		return d_power_dt * radEnergyFractions;
	}

	// :547-558
	static void SolveLinearEqs(JacobianResult const &jacobian, double &x0, VA &xi)
	{
		auto ratios = jacobian.J0g / jacobian.Jgg;
		x0 = (sum(ratios * jacobian.Fg) - jacobian.F0) / (-sum(ratios * jacobian.Jg0) + jacobian.J00);
		xi = (-1.0 * jacobian.Fg - jacobian.Jg0 * x0) / jacobian.Jgg;
	}

	// :1169-1250
	[[nodiscard]] auto ComputeRadQuantityExponents(VA const &quant, VA const &bnd) const -> VA
	{
		VA bin_center(nGroups_), quant_mean(nGroups_), logslopes(nGroups_), exponents(nGroups_);
		for (int g = 0; g < nGroups_; ++g) {
			bin_center[g] = std::sqrt(bnd[g] * bnd[g + 1]);
			quant_mean[g] = quant[g] / (bnd[g + 1] - bnd[g]);
			if (g > 0) {
				if (quant_mean[g] == 0.0 && quant_mean[g - 1] == 0.0) {
					logslopes[g - 1] = 0.0;
				} else if (quant_mean[g - 1] * quant_mean[g] <= 0.0) {
					if (quant_mean[g] > quant_mean[g - 1]) {
						logslopes[g - 1] = inf;
					} else {
						logslopes[g - 1] = -inf;
					}
				} else {
					logslopes[g - 1] = std::log(std::abs(quant_mean[g] / quant_mean[g - 1])) / std::log(bin_center[g] / bin_center[g - 1]);
				}
			}
		}
		for (int g = 0; g < nGroups_; ++g) {
			if (g == 0) {
				exponents[g] = special_edge_bin_slopes ? 2.0 : -1.0;
			} else if (g == nGroups_ - 1) {
				exponents[g] = special_edge_bin_slopes ? -4.0 : -1.0;
			} else {
				exponents[g] = minmod_func(logslopes[g - 1], logslopes[g]);
			}
		}
		static_assert(!PPL_free_slope_st_total);
		return exponents;
	}

	// :1252-1287
	[[nodiscard]] auto ComputeGroupMeanOpacity(KappaExpoLower const &kel, VA const &radBoundaryRatios, VA const &alpha_quant) const -> VA
	{
		VA const &alpha_kappa = kel.expo;
		VA const &kappa_lower = kel.lower;
		VA kappa(nGroups_);
		for (int g = 0; g < nGroups_; ++g) {
			double alpha = alpha_quant[g] + 1.0;
			if (alpha > 100.) {
				kappa[g] = kappa_lower[g] * std::pow(radBoundaryRatios[g], kel.expo[g]);
				continue;
			}
			if (alpha < -100.) {
				kappa[g] = kappa_lower[g];
				continue;
			}
			double part1 = 0.0;
			if (std::abs(alpha) < 1e-8) {
				part1 = std::log(radBoundaryRatios[g]);
			} else {
				part1 = (std::pow(radBoundaryRatios[g], alpha) - 1.0) / alpha;
			}
			alpha += alpha_kappa[g];
			double part2 = 0.0;
			if (std::abs(alpha) < 1e-8) {
				part2 = std::log(radBoundaryRatios[g]);
			} else {
				part2 = (std::pow(radBoundaryRatios[g], alpha) - 1.0) / alpha;
			}
			kappa[g] = kappa_lower[g] / part1 * part2;
		}
		return kappa;
	}

	// :1311-1326 (returns 4 pi B(nu) / c)
	[[nodiscard]] auto PlanckFunction(const double nu, const double T) const -> double
	{
		double const coeff = rs.rt.energy_unit / (rs.eos.tr.boltzmann_constant * T);
		double const x = coeff * nu;
		if (x > 100.) {
			return 0.0;
		}
		double planck_integral = NAN;
		if (x <= 1.0e-10) {
			planck_integral = x * x - x * x * x / 2.;
		} else {
			planck_integral = std::pow(x, 3) / (std::exp(x) - 1.0);
		}
		return coeff / (std::pow(planck::PI, 4) / 15.0) * (rs.rt.radiation_constant * std::pow(T, 4)) * planck_integral;
	}

	// :1328-1352
	[[nodiscard]] auto ComputeDiffusionFluxMeanOpacity(VA const &kappaPVec, VA const &kappaEVec, VA const &fourPiBoverC, VA const &delta_nu_kappa_B_at_edge,
							   VA const &delta_nu_B_at_edge, VA const &kappa_slope) const -> VA
	{
		VA kappaF(nGroups_);
		for (int g = 0; g < nGroups_; ++g) {
			kappaF[g] = (kappaPVec[g] + 1. / 3. * kappaEVec[g]) * fourPiBoverC[g] +
				    1. / 3. * (kappa_slope[g] * kappaEVec[g] * fourPiBoverC[g] - delta_nu_kappa_B_at_edge[g]);
			auto const denom = 4. / 3. * fourPiBoverC[g] - 1. / 3. * delta_nu_B_at_edge[g];
			if (denom <= 0.0) {
				kappaF[g] = 0.0;
			} else {
				kappaF[g] /= denom;
			}
		}
		return kappaF;
	}

	// :1354-1365
	[[nodiscard]] auto ComputeBinCenterOpacity(VA const &rad_boundaries, KappaExpoLower const &kel) const -> VA
	{
		VA kappa_center(nGroups_);
		for (int g = 0; g < nGroups_; ++g) {
			kappa_center[g] = kel.lower[g] * std::pow(rad_boundaries[g + 1] / rad_boundaries[g], 0.5 * kel.expo[g]);
		}
		return kappa_center;
	}

	// source_terms_multi_group.hpp:7-60
	[[nodiscard]] auto ComputeModelDependentKappaEAndKappaP(double const T, double const rho, VA const &rad_boundaries, VA const &rad_boundary_ratios,
								VA const &fourPiBoverC, VA const &Erad, int const n_iter, VA const &alpha_E, VA const &alpha_P) const
	    -> OpacityTerms
	{
		OpacityTerms result;
		result.kappaE = VA(nGroups_);
		result.kappaP = VA(nGroups_);
		result.kappaF = VA(nGroups_);
		result.kappaPoverE = VA(nGroups_);
		result.delta_nu_kappa_B_at_edge = VA(nGroups_);
		result.alpha_E = VA(nGroups_);
		result.alpha_P = VA(nGroups_);
		// (the reference leaves kappaF / delta_nu_kappa_B_at_edge / unused alphas of `result` uninitialised: they are always recomputed by
		// ComputeModelDependentKappaFAndDeltaTerms before they are read, :203-207 and :338-342)
		result.kappaF.fillin(NAN);
		result.delta_nu_kappa_B_at_edge.fillin(NAN);

		const auto kappa_expo_and_lower_value = DefineOpacityExponentsAndLowerValues(rad_boundaries, rho, T);

		if (rs.rt.opacity_model == piecewise_constant_opacity) {
			for (int g = 0; g < nGroups_; ++g) {
				result.kappaP[g] = kappa_expo_and_lower_value.lower[g];
				result.kappaE[g] = kappa_expo_and_lower_value.lower[g];
			}
		} else if (rs.rt.opacity_model == PPL_opacity_fixed_slope_spectrum) {
			VA alpha_quant_minus_one(nGroups_);
			alpha_quant_minus_one.fillin(-1.0); // special_edge_bin_slopes = false
			result.kappaP = ComputeGroupMeanOpacity(kappa_expo_and_lower_value, rad_boundary_ratios, alpha_quant_minus_one);
			result.kappaE = result.kappaP;
		} else if (rs.rt.opacity_model == PPL_opacity_full_spectrum) {
			if (n_iter < max_iter_to_update_alpha_E) {
				result.alpha_E = ComputeRadQuantityExponents(Erad, rad_boundaries);
				result.alpha_P = ComputeRadQuantityExponents(fourPiBoverC, rad_boundaries);
			} else {
				result.alpha_E = alpha_E;
				result.alpha_P = alpha_P;
			}
			result.kappaE = ComputeGroupMeanOpacity(kappa_expo_and_lower_value, rad_boundary_ratios, result.alpha_E);
			result.kappaP = ComputeGroupMeanOpacity(kappa_expo_and_lower_value, rad_boundary_ratios, result.alpha_P);
		}
		for (int g = 0; g < nGroups_; ++g) {
			if (result.kappaE[g] > 0.0) {
				result.kappaPoverE[g] = result.kappaP[g] / result.kappaE[g];
			} else {
				result.kappaPoverE[g] = 1.0;
			}
		}
		return result;
	}

	// source_terms_multi_group.hpp:62-96
	void ComputeModelDependentKappaFAndDeltaTerms(double const T, double const rho, VA const &rad_boundaries, VA const &fourPiBoverC,
						      OpacityTerms &opacity_terms) const
	{
		VA delta_nu_B_at_edge(nGroups_);
		const auto kappa_expo_and_lower_value = DefineOpacityExponentsAndLowerValues(rad_boundaries, rho, T);
		for (int g = 0; g < nGroups_; ++g) {
			auto const nu_L = rad_boundaries[g];
			auto const nu_R = rad_boundaries[g + 1];
			auto const B_L = PlanckFunction(nu_L, T);
			auto const B_R = PlanckFunction(nu_R, T);
			auto const kappa_L = kappa_expo_and_lower_value.lower[g];
			auto const kappa_R = kappa_L * std::pow(nu_R / nu_L, kappa_expo_and_lower_value.expo[g]);
			opacity_terms.delta_nu_kappa_B_at_edge[g] = nu_R * kappa_R * B_R - nu_L * kappa_L * B_L;
			delta_nu_B_at_edge[g] = nu_R * B_R - nu_L * B_L;
		}
		if (rs.rt.opacity_model == piecewise_constant_opacity) {
			opacity_terms.kappaF = opacity_terms.kappaP;
		} else {
			static_assert(use_diffuse_flux_mean_opacity);
			opacity_terms.kappaF =
			    ComputeDiffusionFluxMeanOpacity(opacity_terms.kappaP, opacity_terms.kappaE, fourPiBoverC, opacity_terms.delta_nu_kappa_B_at_edge,
							    delta_nu_B_at_edge, kappa_expo_and_lower_value.expo);
		}
	}

	// source_terms_multi_group.hpp:98-147
	[[nodiscard]] auto ComputeJacobianForGas(double /*T_d*/, double Egas_diff, VA const &Erad_diff, VA const &Rvec, VA const &Src, VA const &tau, double c_v,
						 VA const &kappaPoverE, VA const &d_fourpiboverc_d_t, double const num_den, double const dt) const -> JacobianResult
	{
		JacobianResult result;
		const double cscale = rs.rt.c_light / rs.rt.c_hat;
		const double CR_heating = rs.crHeatingRate(num_den) * dt;
		result.F0 = Egas_diff + cscale * sum(Rvec) - CR_heating;
		result.Fg = Erad_diff - (Rvec + Src);
		result.Fg_abs_sum = 0.0;
		for (int g = 0; g < nGroups_; ++g) {
			if (tau[g] > 0.0) {
				result.Fg_abs_sum += std::abs(result.Fg[g]);
			}
		}
		auto dEg_dT = kappaPoverE * d_fourpiboverc_d_t;
		result.J00 = 1.0;
		result.J0g = VA(nGroups_);
		result.J0g.fillin(cscale);
		result.Jg0 = 1.0 / c_v * dEg_dT;
		result.Jgg = VA(nGroups_);
		for (int g = 0; g < nGroups_; ++g) {
			if (tau[g] <= 0.0) {
				result.Jgg[g] = -std::numeric_limits<double>::infinity();
			} else {
				result.Jgg[g] = -1.0 * kappaPoverE[g] / tau[g] - 1.0;
			}
		}
		return result;
	}

	// source_terms_multi_group.hpp:149-358
	[[nodiscard]] auto SolveGasRadiationEnergyExchange(double const Egas0, VA const &Erad0Vec, double const rho, double const dt, int const n_outer_iter,
							   VA const &work, VA const &vel_times_F, VA const &Src, VA const &rad_boundaries, int *p_iteration_counter,
							   int *p_iteration_failure_counter) const -> NewtonIterationResult
	{
		const double c = rs.rt.c_light;
		const double chat = rs.rt.c_hat;
		const double cscale = c / chat;
		const double H_num_den = rho / rs.eos.tr.mean_molecular_weight; // ComputeNumberDensityH (:463-467)

		double Etot0 = Egas0 + cscale * (sum(Erad0Vec) + sum(Src));

		double T_gas = NAN;
		double T_d = NAN;
		double delta_x = NAN;
		VA delta_R(nGroups_), Rvec(nGroups_), tau0(nGroups_), tau(nGroups_), work_local(nGroups_), fourPiBoverC(nGroups_);
		VA rad_boundary_ratios(nGroups_);
		KappaExpoLower kappa_expo_and_lower_value;
		OpacityTerms opacity_terms{};
		opacity_terms.alpha_E = VA(nGroups_); // value-initialised (:186)
		opacity_terms.alpha_P = VA(nGroups_);

		if (rs.rt.opacity_model != piecewise_constant_opacity) {
			for (int g = 0; g < nGroups_; ++g) {
				rad_boundary_ratios[g] = rad_boundaries[g + 1] / rad_boundaries[g];
			}
		}

		double Egas_guess = Egas0;
		auto EradVec_guess = Erad0Vec;

		const double resid_tol = 1.0e-11;
		const int maxIter = 100;
		int n = 0;
		for (; n < maxIter; ++n) {
			T_gas = rs.eos.ComputeTgasFromEint(rho, Egas_guess);
			T_d = T_gas;

			fourPiBoverC = ComputeThermalRadiationMultiGroup(T_d, rad_boundaries);

			opacity_terms = ComputeModelDependentKappaEAndKappaP(T_d, rho, rad_boundaries, rad_boundary_ratios, fourPiBoverC, EradVec_guess, n,
									     opacity_terms.alpha_E, opacity_terms.alpha_P);

			if (n == 0) {
				ComputeModelDependentKappaFAndDeltaTerms(T_d, rho, rad_boundaries, fourPiBoverC, opacity_terms);
			}

			if (n == 0) {
				if ((rs.rt.beta_order == 1) && (include_work_term_in_source)) {
					if (n_outer_iter == 0) {
						for (int g = 0; g < nGroups_; ++g) {
							if (rs.rt.opacity_model == piecewise_constant_opacity) {
								work_local[g] = vel_times_F[g] * opacity_terms.kappaF[g] * chat / (c * c) * dt;
							} else {
								kappa_expo_and_lower_value = DefineOpacityExponentsAndLowerValues(rad_boundaries, rho, T_d);
								work_local[g] = vel_times_F[g] * opacity_terms.kappaF[g] * chat / (c * c) * dt *
										(1.0 + kappa_expo_and_lower_value.expo[g]);
							}
						}
					} else {
						work_local = work;
					}
				} else {
					work_local.fillin(0.0);
				}

				tau0 = dt * rho * opacity_terms.kappaP * chat;
				tau = tau0;
				Rvec = (fourPiBoverC - EradVec_guess / opacity_terms.kappaPoverE) * tau0 + work_local;
				static_assert(!use_D_as_base);
			} else {
				tau = dt * rho * opacity_terms.kappaP * chat;
				for (int g = 0; g < nGroups_; ++g) {
					if (tau[g] > 0.0) {
						EradVec_guess[g] = opacity_terms.kappaPoverE[g] * (fourPiBoverC[g] - (Rvec[g] - work_local[g]) / tau[g]);
						static_assert(!force_rad_floor_in_iteration);
					}
				}
			}

			const auto d_fourpiboverc_d_t = ComputeThermalRadiationTempDerivativeMultiGroup(T_d, rad_boundaries);
			const double c_v = rs.eos.ComputeEintTempDerivative(rho, T_gas);

			const auto Egas_diff = Egas_guess - Egas0;
			const auto Erad_diff = EradVec_guess - Erad0Vec;

			auto jacobian =
			    ComputeJacobianForGas(T_d, Egas_diff, Erad_diff, Rvec, Src, tau, c_v, opacity_terms.kappaPoverE, d_fourpiboverc_d_t, H_num_den, dt);

			if ((std::abs(jacobian.F0 / Etot0) < resid_tol) && (cscale * jacobian.Fg_abs_sum / Etot0 < resid_tol)) {
				break;
			}

			SolveLinearEqs(jacobian, delta_x, delta_R);

			const double T_rad = std::sqrt(std::sqrt(sum(EradVec_guess) / rs.rt.radiation_constant));
			if (enable_dE_constrain && delta_x / c_v > std::max(T_gas, T_rad)) {
				Egas_guess = rs.eos.ComputeEintFromTgas(rho, T_rad);
			} else {
				Egas_guess += delta_x;
				Rvec = Rvec + delta_R;
			}
		}

		if (n >= maxIter) {
			p_iteration_failure_counter[0] += 1;
		}
		p_iteration_counter[0] += 1;
		p_iteration_counter[1] += n + 1;
		p_iteration_counter[2] = std::max(p_iteration_counter[2], n + 1);

		NewtonIterationResult result;
		if (n > 0) {
			ComputeModelDependentKappaFAndDeltaTerms(T_d, rho, rad_boundaries, fourPiBoverC, opacity_terms);
		}
		result.Egas = Egas_guess;
		result.EradVec = EradVec_guess;
		result.work = work_local;
		result.T_gas = T_gas;
		result.T_d = T_d;
		result.opacity_terms = opacity_terms;
		return result;
	}

	// radiation_system.hpp:1420-1483, nGroups_ > 1
	[[nodiscard]] auto ComputeDustTemperatureBateKeto(double const T_gas, double const T_d_init, double const rho, VA const &Erad, double N_d, double dt,
							  double R_sum, int n_step, VA const &rad_boundaries) const -> double
	{
		if (n_step > 0) {
			return T_gas - R_sum / (N_d * std::sqrt(T_gas));
		}
		VA rad_boundary_ratios(nGroups_);
		if (rs.rt.opacity_model != piecewise_constant_opacity) {
			for (int g = 0; g < nGroups_; ++g) {
				rad_boundary_ratios[g] = rad_boundaries[g + 1] / rad_boundaries[g];
			}
		}
		const double c_hat_ = rs.rt.c_hat;
		VA const zero(nGroups_);
		auto rhs = [&](double T_d) -> double {
			const auto fourPiBoverC = ComputeThermalRadiationMultiGroup(T_d, rad_boundaries);
			const auto opacity_terms = ComputeModelDependentKappaEAndKappaP(T_d, rho, rad_boundaries, rad_boundary_ratios, fourPiBoverC, Erad, 0, zero, zero);
			return c_hat_ * dt * rho * sum(opacity_terms.kappaE * Erad - opacity_terms.kappaP * fourPiBoverC) + N_d * std::sqrt(T_gas) * (T_gas - T_d);
		};
		auto jac = [&](double T_d) -> double {
			const auto fourPiBoverC = ComputeThermalRadiationMultiGroup(T_d, rad_boundaries);
			const auto opacity_terms = ComputeModelDependentKappaEAndKappaP(T_d, rho, rad_boundaries, rad_boundary_ratios, fourPiBoverC, Erad, 0, zero, zero);
			const auto d_fourpib_over_c_d_t = ComputeThermalRadiationTempDerivativeMultiGroup(T_d, rad_boundaries);
			return -c_hat_ * dt * rho * sum(opacity_terms.kappaP * d_fourpib_over_c_d_t) - N_d * std::sqrt(T_gas);
		};
		const double Lambda_compare = N_d * std::sqrt(T_gas) * T_gas;
		return RadSystem::BackwardEulerOneVariable(rhs, jac, T_d_init, Lambda_compare);
	}

	// radiation_dust_system.hpp:22-83
	[[nodiscard]] auto ComputeJacobianForGasAndDust(double T_gas, double T_d, double Egas_diff, VA const &Erad_diff, VA const &Rvec, VA const &Src, double coeff_n,
							VA const &tau, double c_v, VA const &kappaPoverE, VA const &d_fourpiboverc_d_t, const double num_den,
							const double dt) const -> JacobianResult
	{
		JacobianResult result;
		const double cscale = rs.rt.c_light / rs.rt.c_hat;
		const auto cooling = netCoolingRate(T_gas, num_den) * dt;
		const auto cooling_derivative = netCoolingRateTempDerivative(T_gas, num_den) * dt;
		const double CR_heating = rs.crHeatingRate(num_den) * dt;
		result.F0 = Egas_diff + cscale * sum(Rvec) + sum(cooling) - CR_heating;
		result.Fg = Erad_diff - (Rvec + Src);
		result.Fg_abs_sum = 0.0;
		for (int g = 0; g < nGroups_; ++g) {
			if (tau[g] > 0.0) {
				result.Fg_abs_sum += std::abs(result.Fg[g]);
			} else {
				result.Fg_abs_sum += std::abs(result.Fg[g] + Rvec[g]);
			}
		}
		auto dEg_dT = kappaPoverE * d_fourpiboverc_d_t;
		result.J00 = 1.0 + sum(cooling_derivative) / c_v;
		result.J0g = VA(nGroups_);
		result.J0g.fillin(cscale);
		const double d_Td_d_T = 3. / 2. - T_d / (2. * T_gas);
		dEg_dT = dEg_dT * d_Td_d_T;
		const double dTd_dRg = -1.0 / (coeff_n * std::sqrt(T_gas));
		const auto rg = kappaPoverE * d_fourpiboverc_d_t * dTd_dRg;
		result.Jg0 = 1.0 / c_v * dEg_dT - (1 / cscale) * cooling_derivative - 1.0 / cscale * rg * result.J00;
		result.Fg = result.Fg - 1.0 / cscale * rg * result.F0;
		result.Jgg = VA(nGroups_);
		for (int g = 0; g < nGroups_; ++g) {
			if (tau[g] <= 0.0) {
				result.Jgg[g] = -std::numeric_limits<double>::infinity();
			} else {
				result.Jgg[g] = -1.0 * kappaPoverE[g] / tau[g] - 1.0;
			}
		}
		return result;
	}

	// radiation_dust_system.hpp:130-196: with photoelectric heating proportional to the energy density of the LAST group (FUV)
	[[nodiscard]] auto ComputeJacobianForGasAndDustWithPE(double T_gas, double T_d, double Egas_diff, VA const &Erad, VA const &Erad0,
							      double PE_heating_energy_derivative, VA const &Rvec, VA const &Src, double coeff_n, VA const &tau, double c_v,
							      VA const &kappaPoverE, VA const &d_fourpiboverc_d_t, double const num_den, double const dt) const
	    -> JacobianResult
	{
		constexpr double LARGE = 1.0e100; // radiation_system.hpp:48
		JacobianResult result;
		const double cscale = rs.rt.c_light / rs.rt.c_hat;
		const auto cooling = netCoolingRate(T_gas, num_den) * dt;
		const auto cooling_derivative = netCoolingRateTempDerivative(T_gas, num_den) * dt;
		const double CR_heating = rs.crHeatingRate(num_den) * dt;
		result.F0 = Egas_diff + cscale * sum(Rvec) + sum(cooling) - PE_heating_energy_derivative * Erad[nGroups_ - 1] - CR_heating;
		result.Fg = Erad - Erad0 - (Rvec + Src);
		result.Fg_abs_sum = 0.0;
		for (int g = 0; g < nGroups_; ++g) {
			if (tau[g] > 0.0) {
				result.Fg_abs_sum += std::abs(result.Fg[g]);
			} else {
				result.Fg_abs_sum += std::abs(result.Fg[g] + Rvec[g]);
			}
		}
		auto d_Eg_d_Rg = -1.0 * kappaPoverE;
		for (int g = 0; g < nGroups_; ++g) {
			if (tau[g] <= 0.0) {
				d_Eg_d_Rg[g] = -LARGE;
			} else {
				d_Eg_d_Rg[g] /= tau[g];
			}
		}
		result.J00 = 1.0 + sum(cooling_derivative) / c_v;
		result.J0g = VA(nGroups_);
		result.J0g.fillin(cscale);
		result.J0g[nGroups_ - 1] -= PE_heating_energy_derivative * d_Eg_d_Rg[nGroups_ - 1];
		const double d_Td_d_T = 3. / 2. - T_d / (2. * T_gas);
		const auto dEg_dT = kappaPoverE * d_fourpiboverc_d_t * d_Td_d_T;
		const double dTd_dRg = -1.0 / (coeff_n * std::sqrt(T_gas));
		const auto rg = kappaPoverE * d_fourpiboverc_d_t * dTd_dRg;
		result.Jg0 = 1.0 / c_v * dEg_dT - (1 / cscale) * cooling_derivative - 1.0 / cscale * rg * result.J00;
		result.Fg = result.Fg - 1.0 / cscale * rg * result.F0;
		result.Jgg = d_Eg_d_Rg + (-1.0);
		result.Jgg[nGroups_ - 1] += rg[nGroups_ - 1] - (rg[nGroups_ - 1] / cscale) * PE_heating_energy_derivative * d_Eg_d_Rg[nGroups_ - 1];
		result.Jg1 = rg - 1.0 / cscale * rg * result.J0g[nGroups_ - 1];
		return result;
	}

	// radiation_dust_system.hpp:198-226: first row, first column, diagonal and the column of the last group
	static void SolveLinearEqsWithLastColumn(JacobianResult const &jacobian, double &x0, VA &xi)
	{
		const int nG = jacobian.Jgg.n;
		const int pe_index = nG - 1;
		const auto ratios = jacobian.J0g / jacobian.Jgg;
		const auto a00_new = jacobian.J00 - sum(ratios * jacobian.Jg0);
		const auto y0_new = jacobian.F0 - sum(ratios * jacobian.Fg);
		auto a01_new = jacobian.J0g[pe_index] - sum(ratios * jacobian.Jg1);
		a01_new = a01_new + ratios[pe_index] * jacobian.Jg1[pe_index] - ratios[pe_index] * jacobian.Jgg[pe_index];
		const auto a10 = jacobian.Jg0[pe_index];
		const auto a11 = jacobian.Jgg[pe_index];
		const auto y1 = jacobian.Fg[pe_index];
		x0 = (y0_new - a01_new / a11 * y1) / (a00_new - a01_new / a11 * a10);
		const auto x1 = (y1 - a10 * x0) / a11;
		xi = VA(nG);
		xi[pe_index] = x1;
		for (int g = 0; g < pe_index; ++g) {
			xi[g] = (jacobian.Fg[g] - jacobian.Jg0[g] * x0 - jacobian.Jg1[g] * x1) / jacobian.Jgg[g];
		}
		x0 *= -1.0;
		xi = xi * -1.0;
	}

	// radiation_dust_system.hpp:85-128
	[[nodiscard]] auto ComputeJacobianForGasAndDustDecoupled(VA const &Erad_diff, VA const &Rvec, VA const &Src, VA const &tau, double lambda_gd_time_dt,
								 VA const &kappaPoverE, VA const &d_fourpiboverc_d_t) const -> JacobianResult
	{
		JacobianResult result;
		result.F0 = -lambda_gd_time_dt + sum(Rvec);
		result.Fg = Erad_diff - (Rvec + Src);
		result.Fg_abs_sum = 0.0;
		for (int g = 0; g < nGroups_; ++g) {
			if (tau[g] > 0.0) {
				result.Fg_abs_sum += std::abs(result.Fg[g]);
			}
		}
		auto dEg_dT = kappaPoverE * d_fourpiboverc_d_t;
		result.J00 = 0.0;
		result.J0g = VA(nGroups_);
		result.J0g.fillin(1.0);
		result.Jg0 = dEg_dT;
		result.Jgg = VA(nGroups_);
		for (int g = 0; g < nGroups_; ++g) {
			if (tau[g] <= 0.0) {
				result.Jgg[g] = -std::numeric_limits<double>::infinity();
			} else {
				result.Jgg[g] = -1.0 * kappaPoverE[g] / tau[g] - 1.0;
			}
		}
		return result;
	}

	// radiation_dust_system.hpp:228-576, and with `with_PE` the photoelectric-heating variant :578-933 (SolveGasDustRadiationEnergyExchangeWithPE: the same
	// text except for the five places marked PE).  p_iteration_counter[3] counts the decoupled solves
	[[nodiscard]] auto SolveGasDustRadiationEnergyExchange(double const Egas0, VA const &Erad0Vec, double const rho, double const coeff_n, double const dt,
							       int const n_outer_iter, VA const &work, VA const &vel_times_F, VA const &Src, VA const &rad_boundaries,
							       int *p_iteration_counter, int *p_iteration_failure_counter, bool const with_PE = false) const
	    -> NewtonIterationResult
	{
		const double c = rs.rt.c_light;
		const double chat = rs.rt.c_hat;
		const double cscale = c / chat;

		int dust_model = 1;
		double T_d0 = NAN;
		double lambda_gd_times_dt = NAN;
		const double T_gas0 = rs.eos.ComputeTgasFromEint(rho, Egas0);
		T_d0 = ComputeDustTemperatureBateKeto(T_gas0, T_gas0, rho, Erad0Vec, coeff_n, dt, NAN, 0, rad_boundaries);
		if (T_d0 < 0.0) {
			p_iteration_failure_counter[1] += 1;
		}
		const double max_Gamma_gd = coeff_n * std::max(std::sqrt(T_gas0) * T_gas0, std::sqrt(T_d0) * T_d0);
		if (cscale * max_Gamma_gd < rs.rt.gas_dust_coupling_threshold * Egas0) {
			dust_model = 2;
			lambda_gd_times_dt = coeff_n * std::sqrt(T_gas0) * (T_gas0 - T_d0);
		}
		double Etot0 = NAN;
		if (dust_model == 1) {
			Etot0 = Egas0 + cscale * (sum(Erad0Vec) + sum(Src));
		} else if (!with_PE) {
			const double fourPiBoverC = sum(ComputeThermalRadiationMultiGroup(T_d0, rad_boundaries));
			Etot0 = std::abs(lambda_gd_times_dt) + fourPiBoverC + (sum(Erad0Vec) + sum(Src));
		} else { // PE (:631)
			Etot0 = std::abs(lambda_gd_times_dt) + (sum(Erad0Vec) + sum(Src));
		}

		double T_gas = NAN;
		double T_d = NAN;
		double delta_x = NAN;
		VA delta_R(nGroups_), Rvec(nGroups_), tau0(nGroups_), tau(nGroups_), work_local(nGroups_), fourPiBoverC(nGroups_);
		VA rad_boundary_ratios(nGroups_);
		KappaExpoLower kappa_expo_and_lower_value;
		OpacityTerms opacity_terms{};
		opacity_terms.alpha_E = VA(nGroups_);
		opacity_terms.alpha_P = VA(nGroups_);
		if (rs.rt.opacity_model != piecewise_constant_opacity) {
			for (int g = 0; g < nGroups_; ++g) {
				rad_boundary_ratios[g] = rad_boundaries[g + 1] / rad_boundaries[g];
			}
		}
		double Egas_guess = Egas0;
		auto EradVec_guess = Erad0Vec;
		T_gas = rs.eos.ComputeTgasFromEint(rho, Egas_guess);
		const double H_num_den = rho / rs.eos.tr.mean_molecular_weight; // ComputeNumberDensityH
		// PE (:683-685): evaluated once, at the initial gas temperature
		const double PE_heating_energy_derivative = with_PE ? dt * rs.peHeatingE1Derivative(T_gas, H_num_den) : 0.0;

		const double resid_tol = 1.0e-11;
		const int maxIter = 100;
		int n = 0;
		for (; n < maxIter; ++n) {
			if (n > 0) {
				T_gas = rs.eos.ComputeTgasFromEint(rho, Egas_guess);
			}
			if (dust_model == 1) {
				if (n == 0) {
					T_d = T_d0;
				} else {
					T_d = T_gas - sum(Rvec) / (coeff_n * std::sqrt(T_gas));
				}
			} else {
				if (n == 0) {
					T_d = T_d0;
				}
			}
			if (T_d < 0.0) {
				p_iteration_failure_counter[1] += 1;
			}

			fourPiBoverC = ComputeThermalRadiationMultiGroup(T_d, rad_boundaries);
			opacity_terms = ComputeModelDependentKappaEAndKappaP(T_d, rho, rad_boundaries, rad_boundary_ratios, fourPiBoverC, EradVec_guess, n,
									     opacity_terms.alpha_E, opacity_terms.alpha_P);
			if (n == 0) {
				ComputeModelDependentKappaFAndDeltaTerms(T_d, rho, rad_boundaries, fourPiBoverC, opacity_terms);
			}
			if (n == 0) {
				if ((rs.rt.beta_order == 1) && (include_work_term_in_source)) {
					if (n_outer_iter == 0) {
						for (int g = 0; g < nGroups_; ++g) {
							if (rs.rt.opacity_model == piecewise_constant_opacity) {
								work_local[g] = vel_times_F[g] * opacity_terms.kappaF[g] * chat / (c * c) * dt;
							} else {
								kappa_expo_and_lower_value = DefineOpacityExponentsAndLowerValues(rad_boundaries, rho, T_d);
								work_local[g] = vel_times_F[g] * opacity_terms.kappaF[g] * chat / (c * c) * dt *
										(1.0 + kappa_expo_and_lower_value.expo[g]);
							}
						}
					} else {
						work_local = work;
					}
				} else {
					work_local.fillin(0.0);
				}
				tau0 = dt * rho * opacity_terms.kappaP * chat;
				tau = tau0;
				Rvec = (fourPiBoverC - EradVec_guess / opacity_terms.kappaPoverE) * tau0 + work_local;
			} else {
				tau = dt * rho * opacity_terms.kappaP * chat;
				for (int g = 0; g < nGroups_; ++g) {
					if (tau[g] > 0.0) {
						EradVec_guess[g] = opacity_terms.kappaPoverE[g] * (fourPiBoverC[g] - (Rvec[g] - work_local[g]) / tau[g]);
					}
				}
			}

			const auto d_fourpiboverc_d_t = ComputeThermalRadiationTempDerivativeMultiGroup(T_d, rad_boundaries);
			const double c_v = rs.eos.ComputeEintTempDerivative(rho, T_gas);
			const auto Egas_diff = Egas_guess - Egas0;
			const auto Erad_diff = EradVec_guess - Erad0Vec;

			JacobianResult jacobian;
			if (dust_model == 1) {
				if (!with_PE) {
					jacobian = ComputeJacobianForGasAndDust(T_gas, T_d, Egas_diff, Erad_diff, Rvec, Src, coeff_n, tau, c_v, opacity_terms.kappaPoverE,
										d_fourpiboverc_d_t, H_num_den, dt);
				} else { // PE (:790-792)
					jacobian = ComputeJacobianForGasAndDustWithPE(T_gas, T_d, Egas_diff, EradVec_guess, Erad0Vec, PE_heating_energy_derivative, Rvec, Src,
										      coeff_n, tau, c_v, opacity_terms.kappaPoverE, d_fourpiboverc_d_t, H_num_den, dt);
				}
			} else {
				jacobian = ComputeJacobianForGasAndDustDecoupled(Erad_diff, Rvec, Src, tau, lambda_gd_times_dt, opacity_terms.kappaPoverE, d_fourpiboverc_d_t);
			}
			if ((std::abs(jacobian.F0 / Etot0) < resid_tol) && (cscale * jacobian.Fg_abs_sum / Etot0 < resid_tol)) {
				break;
			}
			if (with_PE) { // PE (:836)
				SolveLinearEqsWithLastColumn(jacobian, delta_x, delta_R);
			} else {
				SolveLinearEqs(jacobian, delta_x, delta_R);
			}
			if (dust_model == 2) {
				T_d += delta_x;
				Rvec = Rvec + delta_R;
			} else {
				const double T_rad = std::sqrt(std::sqrt(sum(EradVec_guess) / rs.rt.radiation_constant));
				if (enable_dE_constrain && delta_x / c_v > std::max(T_gas, T_rad)) {
					Egas_guess = rs.eos.ComputeEintFromTgas(rho, T_rad);
				} else {
					Egas_guess += delta_x;
					Rvec = Rvec + delta_R;
				}
			}
		}

		const auto cooling_tend = netCoolingRate(T_gas, H_num_den) * dt; // :515
		if (dust_model == 2) { // :516-537 (PE :868-891): line cooling / heating and cosmic-ray heating update the gas energy implicitly
			const double CR_heating = rs.crHeatingRate(H_num_den) * dt;
			const double compare = Egas_guess + cscale * lambda_gd_times_dt + sum(abs(cooling_tend)) + CR_heating;
			const double pe_term = with_PE ? PE_heating_energy_derivative * EradVec_guess[nGroups_ - 1] : 0.0;
			auto rhs = [&](double Egas_) -> double {
				const double T_gas_ = rs.eos.ComputeTgasFromEint(rho, Egas_);
				const auto cooling_ = netCoolingRate(T_gas_, H_num_den) * dt;
				if (with_PE) {
					return Egas_ - Egas0 + cscale * lambda_gd_times_dt + sum(cooling_) - pe_term - CR_heating;
				}
				return Egas_ - Egas0 + cscale * lambda_gd_times_dt + sum(cooling_) - CR_heating;
			};
			auto jac = [&](double Egas_) -> double {
				const double T_gas_ = rs.eos.ComputeTgasFromEint(rho, Egas_);
				const auto d_cooling_d_Tgas_ = netCoolingRateTempDerivative(T_gas_, H_num_den) * dt;
				return 1.0 + sum(d_cooling_d_Tgas_);
			};
			Egas_guess = RadSystem::BackwardEulerOneVariable(rhs, jac, Egas0, compare);
		}
		EradVec_guess = EradVec_guess + (1 / cscale) * cooling_tend; // :539-543

		if (n >= maxIter) {
			p_iteration_failure_counter[0] += 1;
		}
		p_iteration_counter[0] += 1;
		p_iteration_counter[1] += n + 1;
		p_iteration_counter[2] = std::max(p_iteration_counter[2], n + 1);
		if (dust_model == 2) {
			p_iteration_counter[3] += 1;
		}

		NewtonIterationResult result;
		if (n > 0) {
			ComputeModelDependentKappaFAndDeltaTerms(T_d, rho, rad_boundaries, fourPiBoverC, opacity_terms);
		}
		result.Egas = Egas_guess;
		result.EradVec = EradVec_guess;
		result.work = work_local;
		result.T_gas = T_gas;
		result.T_d = T_d;
		result.opacity_terms = opacity_terms;
		return result;
	}

	// source_terms_multi_group.hpp:360-520
	[[nodiscard]] auto UpdateFlux(int const i, int const j, int const k, Array4<const double> const &consPrev, NewtonIterationResult &energy, double const dt,
				      double const gas_update_factor, double const Ekin0) const -> FluxUpdateResult
	{
		std::array<double, 3> Frad_t0{};
		std::array<double, 3> dMomentum{0., 0., 0.};
		std::array<VA, 3> Frad_t1 = {VA(nGroups_), VA(nGroups_), VA(nGroups_)};
		const double gamma_ = rs.eos.tr.gamma;
		const int beta_order_ = rs.rt.beta_order;
		const double c_light_ = rs.rt.c_light;
		const double c_hat_ = rs.rt.c_hat;

		VA const radBoundaries_g = boundaries();
		double const rho = consPrev(i, j, k, 0);
		const double x1GasMom0 = consPrev(i, j, k, 1);
		const double x2GasMom0 = consPrev(i, j, k, 2);
		const double x3GasMom0 = consPrev(i, j, k, 3);
		const std::array<double, 3> gasMtm0 = {x1GasMom0, x2GasMom0, x3GasMom0};

		auto const fourPiBoverC = ComputeThermalRadiationMultiGroup(energy.T_d, radBoundaries_g);
		auto const kappa_expo_and_lower_value = DefineOpacityExponentsAndLowerValues(radBoundaries_g, rho, energy.T_d);

		const double chat = c_hat_;

		for (int g = 0; g < nGroups_; ++g) {
			Frad_t0[0] = consPrev(i, j, k, rs.x1RadFlux_index() + kNumRadVars * g);
			Frad_t0[1] = consPrev(i, j, k, rs.x2RadFlux_index() + kNumRadVars * g);
			Frad_t0[2] = consPrev(i, j, k, rs.x3RadFlux_index() + kNumRadVars * g);

			if ((gamma_ == 1.0) || (beta_order_ == 0)) {
				for (int n = 0; n < 3; ++n) {
					Frad_t1[n][g] = Frad_t0[n] / (1.0 + rho * energy.opacity_terms.kappaF[g] * chat * dt);
					dMomentum[n] += -(Frad_t1[n][g] - Frad_t0[n]) / (c_light_ * chat);
				}
			} else {
				const auto erad = energy.EradVec[g];
				std::array<double, 3> v_terms{};
				auto fx = Frad_t0[0] / (c_light_ * erad);
				auto fy = Frad_t0[1] / (c_light_ * erad);
				auto fz = Frad_t0[2] / (c_light_ * erad);
				double F_coeff = chat * rho * energy.opacity_terms.kappaF[g] * dt;
				auto Tedd = rs.ComputeEddingtonTensor(fx, fy, fz);

				for (int n = 0; n < 3; ++n) {
					double Planck_term = NAN;
					if (include_delta_B) {
						Planck_term =
						    energy.opacity_terms.kappaP[g] * fourPiBoverC[g] - 1.0 / 3.0 * energy.opacity_terms.delta_nu_kappa_B_at_edge[g];
					} else {
						Planck_term = energy.opacity_terms.kappaP[g] * fourPiBoverC[g];
					}
					Planck_term *= chat * dt * gasMtm0[n];

					double pressure_term = 0.0;
					for (int z = 0; z < 3; ++z) {
						pressure_term += gasMtm0[z] * Tedd[n][z] * erad;
					}
					if (rs.rt.opacity_model == piecewise_constant_opacity) {
						pressure_term *= chat * dt * energy.opacity_terms.kappaE[g];
					} else {
						pressure_term *= chat * dt * (1.0 + kappa_expo_and_lower_value.expo[g]) * energy.opacity_terms.kappaE[g];
					}
					v_terms[n] = Planck_term + pressure_term;
				}

				for (int n = 0; n < 3; ++n) {
					Frad_t1[n][g] = (Frad_t0[n] + v_terms[n]) / (1.0 + F_coeff);
					dMomentum[n] += -(Frad_t1[n][g] - Frad_t0[n]) / (c_light_ * chat);
				}
			}
		}

		double x1GasMom1 = consPrev(i, j, k, 1) + dMomentum[0];
		double x2GasMom1 = consPrev(i, j, k, 2) + dMomentum[1];
		double x3GasMom1 = consPrev(i, j, k, 3) + dMomentum[2];

		FluxUpdateResult updated_flux;
		updated_flux.Erad = VA(nGroups_);
		for (int g = 0; g < nGroups_; ++g) {
			updated_flux.Erad[g] = energy.EradVec[g];
		}

		// 3. work term
		if ((gamma_ != 1.0) && (beta_order_ == 1)) {
			double const Egastot1 = RadSystem::ComputeEgasFromEint(rho, x1GasMom1, x2GasMom1, x3GasMom1, energy.Egas);
			double const Ekin1 = Egastot1 - energy.Egas;
			double const dEkin_work = Ekin1 - Ekin0;
			static_assert(include_work_term_in_source);
			energy.Egas -= dEkin_work;
			for (int g = 0; g < nGroups_; ++g) {
				if (rs.rt.opacity_model == piecewise_constant_opacity) {
					energy.work[g] = (x1GasMom1 * Frad_t1[0][g] + x2GasMom1 * Frad_t1[1][g] + x3GasMom1 * Frad_t1[2][g]) *
							 energy.opacity_terms.kappaF[g] * chat / (c_light_ * c_light_) * dt;
				} else {
					energy.work[g] = (x1GasMom1 * Frad_t1[0][g] + x2GasMom1 * Frad_t1[1][g] + x3GasMom1 * Frad_t1[2][g]) *
							 (1.0 + kappa_expo_and_lower_value.expo[g]) * energy.opacity_terms.kappaF[g] * chat / (c_light_ * c_light_) *
							 dt;
				}
			}
		}

		x1GasMom1 = consPrev(i, j, k, 1) + dMomentum[0] * gas_update_factor;
		x2GasMom1 = consPrev(i, j, k, 2) + dMomentum[1] * gas_update_factor;
		x3GasMom1 = consPrev(i, j, k, 3) + dMomentum[2] * gas_update_factor;
		updated_flux.gasMomentum = {x1GasMom1, x2GasMom1, x3GasMom1};
		updated_flux.Frad = Frad_t1;
		return updated_flux;
	}

	// source_terms_multi_group.hpp:522-813.  counters: p_iteration_counter[4], p_iteration_failure_counter[3]
	void AddSourceTermsMultiGroup(Array4<double> const &consVar, Array4<const double> const &radEnergySource, Box const &indexRange, double dt_radiation,
				      const int stage, int *p_iteration_counter, int *p_iteration_failure_counter) const
	{
		Array4<const double> consPrev(consVar.p, consVar.box(), consVar.ncomp);
		Array4<double> const &consNew = consVar;
		auto dt = dt_radiation;
		if (stage == 2) {
			dt = (1.0 - IMEX_a32) * dt_radiation;
		}
		const double gamma_ = rs.eos.tr.gamma;
		const int beta_order_ = rs.rt.beta_order;
		const double c_light_ = rs.rt.c_light;
		const double c_hat_ = rs.rt.c_hat;
		VA const radBoundaries_g = boundaries();

		for (int k = indexRange.lo[2]; k <= indexRange.hi[2]; ++k) {
			for (int j = indexRange.lo[1]; j <= indexRange.hi[1]; ++j) {
				for (int i = indexRange.lo[0]; i <= indexRange.hi[0]; ++i) {
					const double c = c_light_;
					const double chat = c_hat_;

					const double rho = consPrev(i, j, k, 0);
					const double x1GasMom0 = consPrev(i, j, k, 1);
					const double x2GasMom0 = consPrev(i, j, k, 2);
					const double x3GasMom0 = consPrev(i, j, k, 3);
					const double Egastot0 = consPrev(i, j, k, 4);

					VA Erad0Vec(nGroups_);
					for (int g = 0; g < nGroups_; ++g) {
						Erad0Vec[g] = consPrev(i, j, k, rs.radEnergy_index() + kNumRadVars * g);
					}
					const double Erad0 = sum(Erad0Vec);

					VA Src(nGroups_);
					for (int g = 0; g < nGroups_; ++g) {
						Src[g] = dt * (chat * radEnergySource(i, j, k, g));
					}

					double Egas0 = NAN;
					double Ekin0 = NAN;
					double Etot0 = NAN;
					double Egas_guess = NAN;
					VA work(nGroups_), work_prev(nGroups_);

					if (gamma_ != 1.0) {
						Egas0 = RadSystem::ComputeEintFromEgas(rho, x1GasMom0, x2GasMom0, x3GasMom0, Egastot0);
						Etot0 = Egas0 + (c / chat) * (Erad0 + sum(Src));
						Ekin0 = Egastot0 - Egas0;
					}
					(void)Etot0;

					VA radBoundaries_g_copy = radBoundaries_g;
					VA radBoundaryRatios_copy(nGroups_);
					for (int g = 0; g < nGroups_; ++g) {
						radBoundaryRatios_copy[g] = radBoundaries_g_copy[g + 1] / radBoundaries_g_copy[g];
					}
					VA alpha_quant_minus_one(nGroups_);
					alpha_quant_minus_one.fillin(-1.0); // special_edge_bin_slopes = false (:591-606; other models: unused)

					double gas_update_factor = 1.0;
					if (stage == 1) {
						gas_update_factor = IMEX_a32;
					}
					// :612-617
					const double H_num_den = rho / rs.eos.tr.mean_molecular_weight;
					const double cscale = c / chat;
					double coeff_n = NAN;
					if (rs.rt.enable_dust_gas_thermal_coupling_model) {
						coeff_n = dt * rs.rt.dustGasInteractionCoeff * H_num_den * H_num_den / cscale;
					}

					const int max_iter = 5;
					int iter = 0;
					for (; iter < max_iter; ++iter) {
						KappaExpoLower kappa_expo_and_lower_value;
						NewtonIterationResult updated_energy;
						updated_energy.opacity_terms.kappaF = VA(nGroups_);

						if (gamma_ != 1.0) {
							VA vel_times_F(nGroups_);
							if (include_work_term_in_source) {
								if (iter == 0) {
									for (int g = 0; g < nGroups_; ++g) {
										const double frad0 = consPrev(i, j, k, rs.x1RadFlux_index() + kNumRadVars * g);
										const double frad1 = consPrev(i, j, k, rs.x2RadFlux_index() + kNumRadVars * g);
										const double frad2 = consPrev(i, j, k, rs.x3RadFlux_index() + kNumRadVars * g);
										vel_times_F[g] = (x1GasMom0 * frad0 + x2GasMom0 * frad1 + x3GasMom0 * frad2);
									}
								}
							}

							if (!rs.rt.enable_dust_gas_thermal_coupling_model) { // :706-723
								updated_energy = SolveGasRadiationEnergyExchange(Egas0, Erad0Vec, rho, dt, iter, work, vel_times_F, Src,
														 radBoundaries_g_copy, p_iteration_counter,
														 p_iteration_failure_counter);
							} else {
								updated_energy = SolveGasDustRadiationEnergyExchange(Egas0, Erad0Vec, rho, coeff_n, dt, iter, work, vel_times_F, Src,
														     radBoundaries_g_copy, p_iteration_counter,
														     p_iteration_failure_counter, rs.rt.enable_photoelectric_heating);
							}

							Egas_guess = updated_energy.Egas;
							for (int g = 0; g < nGroups_; ++g) {
								work_prev[g] = updated_energy.work[g];
							}
							kappa_expo_and_lower_value = DefineOpacityExponentsAndLowerValues(radBoundaries_g_copy, rho, updated_energy.T_d);
						} else {
							kappa_expo_and_lower_value = DefineOpacityExponentsAndLowerValues(radBoundaries_g_copy, rho, NAN);
							if (rs.rt.opacity_model == piecewise_constant_opacity) {
								for (int g = 0; g < nGroups_; ++g) {
									updated_energy.opacity_terms.kappaF[g] = kappa_expo_and_lower_value.lower[g];
								}
							} else {
								updated_energy.opacity_terms.kappaF =
								    ComputeGroupMeanOpacity(kappa_expo_and_lower_value, radBoundaryRatios_copy, alpha_quant_minus_one);
							}
							// (gamma == 1: UpdateFlux reads EradVec only through updated_flux.Erad, which the reference stores unconditionally
							// at :776 from an uninitialised NewtonIterationResult; restated with the unchanged radiation energy)
							updated_energy.EradVec = Erad0Vec;
							updated_energy.work = VA(nGroups_);
						}

						auto updated_flux = UpdateFlux(i, j, k, consPrev, updated_energy, dt, gas_update_factor, Ekin0);

						bool work_converged = true;
						if ((beta_order_ == 0) || (gamma_ == 1.0) || (!include_work_term_in_source)) {
							// pass
						} else {
							work = updated_energy.work;
							auto const Egastot1 = RadSystem::ComputeEgasFromEint(rho, updated_flux.gasMomentum[0], updated_flux.gasMomentum[1],
													     updated_flux.gasMomentum[2], Egas_guess);
							const double rel_lag_tol = 1.0e-8;
							const double lag_tol = 1.0e-13;
							double ref_work = rel_lag_tol * sum(abs(work));
							ref_work = std::max(ref_work, lag_tol * Egastot1 / (c_light_ / c_hat_));
							if (sum(abs(work - work_prev)) > ref_work) {
								work_converged = false;
							}
						}

						if (work_converged) {
							consNew(i, j, k, 1) = updated_flux.gasMomentum[0];
							consNew(i, j, k, 2) = updated_flux.gasMomentum[1];
							consNew(i, j, k, 3) = updated_flux.gasMomentum[2];
							for (int g = 0; g < nGroups_; ++g) {
								consNew(i, j, k, rs.radEnergy_index() + kNumRadVars * g) = updated_flux.Erad[g];
								consNew(i, j, k, rs.x1RadFlux_index() + kNumRadVars * g) = updated_flux.Frad[0][g];
								consNew(i, j, k, rs.x2RadFlux_index() + kNumRadVars * g) = updated_flux.Frad[1][g];
								consNew(i, j, k, rs.x3RadFlux_index() + kNumRadVars * g) = updated_flux.Frad[2][g];
							}
							if (gamma_ != 1.0) {
								Egas_guess = updated_energy.Egas;
							}
							break;
						}
					} // end full-step iteration

					if (iter >= max_iter) {
						p_iteration_failure_counter[2] += 1;
					}

					// 4b. (:796-811)
					const auto x1GasMom1 = consNew(i, j, k, 1);
					const auto x2GasMom1 = consNew(i, j, k, 2);
					const auto x3GasMom1 = consNew(i, j, k, 3);
					if (gamma_ != 1.0) {
						Egas_guess = Egas0 + (Egas_guess - Egas0) * gas_update_factor;
						consNew(i, j, k, 5) = Egas_guess;
						consNew(i, j, k, 4) = RadSystem::ComputeEgasFromEint(rho, x1GasMom1, x2GasMom1, x3GasMom1, Egas_guess);
					}
				}
			}
		}
	}
};

} // namespace mg

} // namespace oracle

#endif // ORACLE_RADIATION_MULTIGROUP_HPP_
