// ORACLE — TEST INFRASTRUCTURE ONLY (see grid.hpp header).
//
// radiation.hpp: restatement of the two-moment (M1) radiation operators
//   reference src/radiation/radiation_system.hpp            (RadSystem<problem_t>, line refs per function)
//   reference src/radiation/source_terms_single_group.hpp   (AddSourceTermsSingleGroup)
// The transport operators (ConservedToPrimitive, ComputeFluxes, PredictStep, AddFluxesRK2) run over rt.nGroups photon groups, group g
// occupying components nstartHyperbolic_ + 4 g .. + 4 g + 3 (Physics_Indices, physics_info.hpp:20-47).  The single-group source term is
// here; the multigroup one (source_terms_multi_group.hpp) is in radiation_multigroup.hpp.  No dust / photoelectric / line-cooling models
// (ISM_Traits defaults).  The device hooks a problem specialises (ComputePlanckOpacity / ComputeEnergyMeanOpacity /
// ComputeFluxMeanOpacity / DefineOpacityExponentsAndLowerValues) are std::function members here.
#ifndef ORACLE_RADIATION_HPP_
#define ORACLE_RADIATION_HPP_

#include <array>
#include <cmath>
#include <functional>
#include <limits>
#include <vector>

#include "eos.hpp"
#include "grid.hpp"
#include "hyperbolic.hpp"

namespace oracle
{

// radiation_system.hpp:49-52 (IMEX PD-ARS)
constexpr double IMEX_a22 = 1.0;
constexpr double IMEX_a32 = 0.5;
// radiation_system.hpp:34-40 hyper-parameters
constexpr bool add_line_cooling_to_radiation_in_jac = false;
constexpr bool force_rad_floor_in_iteration = false;
constexpr bool include_work_term_in_source = true;
constexpr bool enable_dE_constrain = true; // :44

constexpr int kNumRadVars = 4; // physics_numVars.hpp:9
constexpr int kMaxGroups = 64;  // (oracle storage bound; the reference's nGroups is a compile-time constant)

// OpacityModel (radiation_system.hpp:64-71)
enum OpacityModel { single_group = 0, piecewise_constant_opacity = 1, PPL_opacity_fixed_slope_spectrum = 2, PPL_opacity_full_spectrum = 3 };

// runtime stand-in for RadSystem_Traits<problem_t> (radiation_system.hpp:73-82)
struct RadTraits {
	double c_light = C::c_light;
	double c_hat = C::c_light;
	double radiation_constant = C::a_rad;
	double Erad_floor = 0.;
	int beta_order = 1;
	// 0: std::pow(T, 4) / std::pow(T, 3) as in the reference; 1: repeated multiplication (used by tests that want
	// bit-level agreement with the device build, whose libm differs from glibc by <= 1 ulp in pow)
	int pow_mode = 0;
	// the ComputeEddingtonFactor hook (radiation_system.hpp:773-790 is the default, Levermore's closure); problems that specialise it
	// use the Eddington approximation chi = 1/3 (e.g. src/problems/RadhydroShockCGS/test_radhydro_shock_cgs.cpp:88-91)
	int eddington_model = 0; // 0: Levermore, 1: chi = 1/3
	// ISM_Traits<problem_t>::enable_dust_gas_thermal_coupling_model (radiation_system.hpp:84-98) and QuokkaSimulation::dustGasInteractionCoeff_
	// (QuokkaSimulation.hpp:127, deck key radiation.dust_gas_interaction_coeff)
	bool enable_dust_gas_thermal_coupling_model = false;
	double dustGasInteractionCoeff = 2.5e-34;
	bool enable_photoelectric_heating = false; // ISM_Traits (multigroup dust model: SolveGasDustRadiationEnergyExchangeWithPE)
	double gas_dust_coupling_threshold = 1.0e-6; // ISM_Traits (multigroup: below it gas and dust are treated as decoupled)
	// the ComputeThermalRadiationSingleGroup / ...TempDerivativeSingleGroup hooks: 0 the default a T^4 / 4 a T^3 (:471-479, :499-503);
	// 1: a T / a, the linearised emission of RadDust (src/problems/RadDust/test_rad_dust.cpp:86-97)
	int thermal_model = 0;
	// multigroup (Physics_Traits::nGroups, RadSystem_Traits::radBoundaries / energy_unit / opacity_model; radiation_system.hpp:201-223)
	int nGroups = 1;
	std::vector<double> radBoundaries; // nGroups + 1 group edges, in units of energy_unit / h ... (energy = energy_unit * boundary)
	double energy_unit = C::hplanck;
	int opacity_model = single_group;
};

struct RadSystem {
	RadTraits rt;
	EOS eos;
	int nstartHyperbolic_ = 6; // Physics_Indices::radFirstIndex (no passive scalars)
	int ndim = 3;
	std::function<double(double, double)> ComputePlanckOpacity;	 // radiation_system.hpp:1141
	std::function<double(double, double)> ComputeFluxMeanOpacity;	 // :1146 (default: Planck)
	std::function<double(double, double)> ComputeEnergyMeanOpacity; // :1151 (default: Planck)
	// :1155-1167 DefineOpacityExponentsAndLowerValues(rad_boundaries, rho, Tgas) -> (exponents[nGroups+1], lower values[nGroups+1])
	std::function<void(double const *rad_boundaries, double rho, double Tgas, double *exponents, double *lower_values)> DefineOpacityExponentsAndLowerValues;
	// the ISM heating / cooling hooks (radiation_system.hpp:344-353; defaults :524-545 and radiation_dust_system.hpp:7-12: zero)
	std::function<void(double temperature, double num_density, double *cooling_per_group)> DefineNetCoolingRate;
	std::function<void(double temperature, double num_density, double *dcooling_dT_per_group)> DefineNetCoolingRateTempDerivative;
	std::function<double(double num_density)> DefineCosmicRayHeatingRate;
	std::function<double(double temperature, double num_density)> DefinePhotoelectricHeatingE1Derivative;
	[[nodiscard]] auto crHeatingRate(double n) const -> double { return DefineCosmicRayHeatingRate ? DefineCosmicRayHeatingRate(n) : 0.0; }
	[[nodiscard]] auto peHeatingE1Derivative(double T, double n) const -> double
	{
		return DefinePhotoelectricHeatingE1Derivative ? DefinePhotoelectricHeatingE1Derivative(T, n) : 0.0;
	}
	// group 0 of the two cooling hooks (the single-group scheme reads [0])
	[[nodiscard]] auto netCoolingRate0(double T, double n) const -> double
	{
		double v[kMaxGroups] = {};
		if (DefineNetCoolingRate) {
			DefineNetCoolingRate(T, n, v);
		}
		return v[0];
	}
	[[nodiscard]] auto netCoolingRateTempDerivative0(double T, double n) const -> double
	{
		double v[kMaxGroups] = {};
		if (DefineNetCoolingRateTempDerivative) {
			DefineNetCoolingRateTempDerivative(T, n, v);
		}
		return v[0];
	}

	[[nodiscard]] auto nGroups_() const -> int { return rt.nGroups; }
	[[nodiscard]] auto nRadComps() const -> int { return kNumRadVars * rt.nGroups; }
	// :211
	[[nodiscard]] auto Erad_floor_() const -> double { return rt.Erad_floor / rt.nGroups; }

	// radVarIndex (radiation_system.hpp:183)
	[[nodiscard]] auto radEnergy_index() const -> int { return nstartHyperbolic_; }
	[[nodiscard]] auto x1RadFlux_index() const -> int { return nstartHyperbolic_ + 1; }
	[[nodiscard]] auto x2RadFlux_index() const -> int { return nstartHyperbolic_ + 2; }
	[[nodiscard]] auto x3RadFlux_index() const -> int { return nstartHyperbolic_ + 3; }

	[[nodiscard]] auto pow4(double T) const -> double { return (rt.pow_mode == 0) ? std::pow(T, 4) : (T * T) * (T * T); }
	[[nodiscard]] auto pow3(double T) const -> double { return (rt.pow_mode == 0) ? std::pow(T, 3) : (T * T) * T; }

	// radiation_system.hpp:471-479
	[[nodiscard]] auto ComputeThermalRadiationSingleGroup(double temperature) const -> double
	{
		if (rt.thermal_model == 1) {
			return rt.radiation_constant * temperature;
		}
		double power = rt.radiation_constant * pow4(temperature);
		if (power < Erad_floor_()) {
			power = Erad_floor_();
		}
		return power;
	}
	// :499-503
	[[nodiscard]] auto ComputeThermalRadiationTempDerivativeSingleGroup(double temperature) const -> double
	{
		if (rt.thermal_model == 1) {
			return rt.radiation_constant;
		}
		return 4. * rt.radiation_constant * pow3(temperature);
	}

	// :1387-1418 BackwardEulerOneVariable
	template <typename RHSFunction, typename JacFunction>
	[[nodiscard]] static auto BackwardEulerOneVariable(RHSFunction const &rhs, JacFunction const &jac, const double x0, const double compare) -> double
	{
		double x = x0;
		const double rel_tol = 1.0e-8;
		const double rel_change_tol = 1.0e-6;
		const int max_iter_td = 100;
		int iter_Td = 0;
		for (; iter_Td < max_iter_td; ++iter_Td) {
			const auto the_rhs = rhs(x);
			if (std::abs(the_rhs) < rel_tol * compare) {
				break;
			}
			const double dT = -the_rhs / jac(x);
			x += dT;
			if (iter_Td > 0) {
				if (std::abs(dT) < rel_change_tol * std::abs(x)) {
					break;
				}
			}
		}
		if (iter_Td >= max_iter_td) {
			x = -1.0;
		}
		return x;
	}

	// :1420-1483 ComputeDustTemperatureBateKeto, nGroups_ == 1
	[[nodiscard]] auto ComputeDustTemperatureBateKeto(double const T_gas, double const T_d_init, double const rho, double const Erad0, double N_d, double dt,
							  double R_sum, int n_step) const -> double
	{
		if (n_step > 0) {
			const auto T_d = T_gas - R_sum / (N_d * std::sqrt(T_gas));
			return T_d;
		}
		const double c_hat_ = rt.c_hat;
		auto rhs = [=](double T_d) -> double {
			const auto fourPiBoverC = ComputeThermalRadiationSingleGroup(T_d);
			const auto kappaE = ComputeEnergyMeanOpacity(rho, T_d);
			const auto kappaP = ComputePlanckOpacity(rho, T_d);
			return c_hat_ * dt * rho * (kappaE * Erad0 - kappaP * fourPiBoverC) + N_d * std::sqrt(T_gas) * (T_gas - T_d);
		};
		auto jac = [=](double T_d) -> double {
			const auto kappaP = ComputePlanckOpacity(rho, T_d);
			const auto d_fourpib_over_c_d_t = ComputeThermalRadiationTempDerivativeSingleGroup(T_d);
			return -c_hat_ * dt * rho * (kappaP * d_fourpib_over_c_d_t) - N_d * std::sqrt(T_gas);
		};
		const double Lambda_compare = N_d * std::sqrt(T_gas) * T_gas;
		return BackwardEulerOneVariable(rhs, jac, T_d_init, Lambda_compare);
	}
	// :1289-1308
	[[nodiscard]] static auto ComputeEintFromEgas(double density, double X1GasMom, double X2GasMom, double X3GasMom, double Etot) -> double
	{
		const double p_sq = X1GasMom * X1GasMom + X2GasMom * X2GasMom + X3GasMom * X3GasMom;
		const double Ekin = p_sq / (2.0 * density);
		return Etot - Ekin;
	}
	[[nodiscard]] static auto ComputeEgasFromEint(double density, double X1GasMom, double X2GasMom, double X3GasMom, double Eint) -> double
	{
		const double p_sq = X1GasMom * X1GasMom + X2GasMom * X2GasMom + X3GasMom * X3GasMom;
		const double Ekin = p_sq / (2.0 * density);
		return Eint + Ekin;
	}

	// :773-790 Levermore closure
	[[nodiscard]] auto ComputeEddingtonFactor(double f_in) const -> double
	{
		if (rt.eddington_model == 1) {
			return (1. / 3.);
		}
		const double f = clamp(f_in, 0., 1.);
		const double f_fac = std::sqrt(4.0 - 3.0 * (f * f));
		const double chi = (3.0 + 4.0 * (f * f)) / (5.0 + 2.0 * f_fac);
		return chi;
	}

	// :873-916
	[[nodiscard]] auto ComputeEddingtonTensor(const double fx, const double fy, const double fz) const -> std::array<std::array<double, 3>, 3>
	{
		auto f = std::sqrt(fx * fx + fy * fy + fz * fz);
		std::array<double, 3> fvec = {fx, fy, fz};
		std::array<double, 3> n{};
		for (int ii = 0; ii < 3; ++ii) {
			n[ii] = (f > 0.) ? (fvec[ii] / f) : 0.;
		}
		const double chi = ComputeEddingtonFactor(f);
		const double Tdiag = (1.0 - chi) / 2.0;
		const double Tf = (3.0 * chi - 1.0) / 2.0;
		std::array<std::array<double, 3>, 3> T{};
		for (int ii = 0; ii < 3; ++ii) {
			for (int jj = 0; jj < 3; ++jj) {
				const double delta_ij = (ii == jj) ? 1 : 0;
				T[ii][jj] = Tdiag * delta_ij + Tf * (n[ii] * n[jj]);
			}
		}
		return T;
	}

	struct RadPressureResult {
		std::array<double, 4> F;
		double S;
	};

	// :918-983
	[[nodiscard]] auto ComputeRadPressure(int dir, const double erad, const double Fx, const double Fy, const double Fz, const double fx,
					      const double fy, const double fz) const -> RadPressureResult
	{
		auto T = ComputeEddingtonTensor(fx, fy, fz);
		const double Tnormal = T[dir][dir];
		const double Fn = (dir == 0) ? Fx : (dir == 1) ? Fy : Fz;
		const double Tnx = T[dir][0];
		const double Tny = T[dir][1];
		const double Tnz = T[dir][2];
		RadPressureResult result{};
		result.F = {Fn, Tnx * erad, Tny * erad, Tnz * erad};
		result.S = std::max(0.1, std::sqrt(Tnormal));
		return result;
	}

	// :589-614
	void ConservedToPrimitive(Array4<const double> const &cons, Array4<double> const &primVar, Box const &indexRange) const
	{
		for (int k = indexRange.lo[2]; k <= indexRange.hi[2]; ++k) {
			for (int j = indexRange.lo[1]; j <= indexRange.hi[1]; ++j) {
				for (int i = indexRange.lo[0]; i <= indexRange.hi[0]; ++i) {
					for (int g = 0; g < nGroups_(); ++g) {
						const auto E_r = cons(i, j, k, radEnergy_index() + kNumRadVars * g);
						const auto Fx = cons(i, j, k, x1RadFlux_index() + kNumRadVars * g);
						const auto Fy = cons(i, j, k, x2RadFlux_index() + kNumRadVars * g);
						const auto Fz = cons(i, j, k, x3RadFlux_index() + kNumRadVars * g);
						primVar(i, j, k, 0 + kNumRadVars * g) = E_r;
						primVar(i, j, k, 1 + kNumRadVars * g) = Fx / (rt.c_light * E_r);
						primVar(i, j, k, 2 + kNumRadVars * g) = Fy / (rt.c_light * E_r);
						primVar(i, j, k, 3 + kNumRadVars * g) = Fz / (rt.c_light * E_r);
					}
				}
			}
		}
	}

	using RadCons = std::array<double, kNumRadVars * kMaxGroups>; // std::array<Real, nvarHyperbolic_>

	// :626-644
	[[nodiscard]] auto isStateValid(RadCons const &cons) const -> bool
	{
		bool isValid = true;
		for (int g = 0; g < nGroups_(); ++g) {
			const auto E_r = cons[0 + kNumRadVars * g];
			const auto Fx = cons[1 + kNumRadVars * g];
			const auto Fy = cons[2 + kNumRadVars * g];
			const auto Fz = cons[3 + kNumRadVars * g];
			const auto Fnorm = std::sqrt(Fx * Fx + Fy * Fy + Fz * Fz);
			const auto f = Fnorm / (rt.c_light * E_r);
			bool isNonNegative = (E_r > 0.);
			bool isFluxCausal = (f <= 1.);
			isValid = (isValid && isNonNegative && isFluxCausal);
		}
		return isValid;
	}

	// :646-665
	void amendRadState(RadCons &cons) const
	{
		for (int g = 0; g < nGroups_(); ++g) {
			auto E_r = cons[0 + kNumRadVars * g];
			if (E_r < Erad_floor_()) {
				E_r = Erad_floor_();
				cons[0 + kNumRadVars * g] = Erad_floor_();
			}
			const auto Fx = cons[1 + kNumRadVars * g];
			const auto Fy = cons[2 + kNumRadVars * g];
			const auto Fz = cons[3 + kNumRadVars * g];
			if (Fx * Fx + Fy * Fy + Fz * Fz > rt.c_light * rt.c_light * E_r * E_r) {
				const auto Fnorm = std::sqrt(Fx * Fx + Fy * Fy + Fz * Fz);
				cons[1 + kNumRadVars * g] = Fx / Fnorm * rt.c_light * E_r;
				cons[2 + kNumRadVars * g] = Fy / Fnorm * rt.c_light * E_r;
				cons[3 + kNumRadVars * g] = Fz / Fnorm * rt.c_light * E_r;
			}
		}
	}

	// :803-871  interface-averaged cell optical depth of every photon group at the left face of cell (i, j, k) of the view
	void ComputeCellOpticalDepth(int dir, View<const double> const &consVar, double const dx[3], int i, int j, int k, double *optical_depths) const
	{
		// piecewise-constant reconstruction (gas components 0..4: Physics_Indices::hydroFirstIndex = 0)
		const double rho_L = consVar(i - 1, j, k, 0);
		const double rho_R = consVar(i, j, k, 0);
		const double x1GasMom_L = consVar(i - 1, j, k, 1);
		const double x1GasMom_R = consVar(i, j, k, 1);
		const double x2GasMom_L = consVar(i - 1, j, k, 2);
		const double x2GasMom_R = consVar(i, j, k, 2);
		const double x3GasMom_L = consVar(i - 1, j, k, 3);
		const double x3GasMom_R = consVar(i, j, k, 3);
		const double Egas_L = consVar(i - 1, j, k, 4);
		const double Egas_R = consVar(i, j, k, 4);

		double Eint_L = NAN;
		double Eint_R = NAN;
		double Tgas_L = NAN;
		double Tgas_R = NAN;
		if (eos.tr.gamma != 1.0) { // :842-847
			Eint_L = ComputeEintFromEgas(rho_L, x1GasMom_L, x2GasMom_L, x3GasMom_L, Egas_L);
			Eint_R = ComputeEintFromEgas(rho_R, x1GasMom_R, x2GasMom_R, x3GasMom_R, Egas_R);
			Tgas_L = eos.ComputeTgasFromEint(rho_L, Eint_L);
			Tgas_R = eos.ComputeTgasFromEint(rho_R, Eint_R);
		}
		const double dl = dx[dir]; // :849-856

		if (nGroups_() == 1) { // :859-862
			const auto &fluxMean = ComputeFluxMeanOpacity ? ComputeFluxMeanOpacity : ComputePlanckOpacity; // (:1146: the default hook is the Planck mean)
			const double tau_L = dl * rho_L * fluxMean(rho_L, Tgas_L);
			const double tau_R = dl * rho_R * fluxMean(rho_R, Tgas_R);
			optical_depths[0] = (tau_L * tau_R * 2.) / (tau_L + tau_R); // harmonic mean
		} else { // :863-868  DefineOpacityExponentsAndLowerValues + ComputeBinCenterOpacity (:1354-1365) either side
			const int ng = nGroups_();
			std::vector<double> expo_L(ng + 1, NAN), lower_L(ng + 1, NAN), expo_R(ng + 1, NAN), lower_R(ng + 1, NAN); // (default hook: NaN, :1155-1167)
			if (DefineOpacityExponentsAndLowerValues) {
				DefineOpacityExponentsAndLowerValues(rt.radBoundaries.data(), rho_L, Tgas_L, expo_L.data(), lower_L.data());
				DefineOpacityExponentsAndLowerValues(rt.radBoundaries.data(), rho_R, Tgas_R, expo_R.data(), lower_R.data());
			}
			for (int g = 0; g < ng; ++g) {
				const double ratio = rt.radBoundaries[g + 1] / rt.radBoundaries[g];
				const double kappa_L = lower_L[g] * std::pow(ratio, 0.5 * expo_L[g]);
				const double kappa_R = lower_R[g] * std::pow(ratio, 0.5 * expo_R[g]);
				const double tau_L = dl * rho_L * kappa_L;
				const double tau_R = dl * rho_R * kappa_R;
				optical_depths[g] = (tau_L * tau_R * 2.) / (tau_L + tau_R);
			}
		}
	}

	// :985-1139; use_wavespeed_correction (QuokkaSimulation.hpp:133, default false) with the cell widths it needs
	void ComputeFluxes(int dir, Array4<double> const &x1Flux_in, Array4<double> const &x1FluxDiffusive_in, Array4<const double> const &x1LeftState_in,
			   Array4<const double> const &x1RightState_in, Box const &indexRange, Array4<const double> const &consVar_in, double const *dx = nullptr,
			   bool const use_wavespeed_correction = false) const
	{
		View<const double> x1LeftState(x1LeftState_in, dir);
		View<const double> x1RightState(x1RightState_in, dir);
		View<double> x1Flux(x1Flux_in, dir);
		View<double> x1FluxDiffusive(x1FluxDiffusive_in, dir);
		View<const double> consVar(consVar_in, dir);
		const double c_light_ = rt.c_light;
		const double c_hat_ = rt.c_hat;

		for (int k_in = indexRange.lo[2]; k_in <= indexRange.hi[2]; ++k_in) {
			for (int j_in = indexRange.lo[1]; j_in <= indexRange.hi[1]; ++j_in) {
				for (int i_in = indexRange.lo[0]; i_in <= indexRange.hi[0]; ++i_in) {
					auto [i, j, k] = reorderMultiIndex(dir, i_in, j_in, k_in);

					// :1019-1022
					double tau_cell[kMaxGroups] = {};
					if (use_wavespeed_correction) {
						ComputeCellOpticalDepth(dir, consVar, dx, i, j, k, tau_cell);
					}

					for (int g = 0; g < nGroups_(); ++g) {
						const int pg = kNumRadVars * g; // component offset of group g

						double erad_L = x1LeftState(i, j, k, pg + 0);
						double erad_R = x1RightState(i, j, k, pg + 0);
						double fx_L = x1LeftState(i, j, k, pg + 1);
						double fx_R = x1RightState(i, j, k, pg + 1);
						double fy_L = x1LeftState(i, j, k, pg + 2);
						double fy_R = x1RightState(i, j, k, pg + 2);
						double fz_L = x1LeftState(i, j, k, pg + 3);
						double fz_R = x1RightState(i, j, k, pg + 3);

						double f_L = std::sqrt(fx_L * fx_L + fy_L * fy_L + fz_L * fz_L);
						double f_R = std::sqrt(fx_R * fx_R + fy_R * fy_R + fz_R * fz_R);

						double Fx_L = fx_L * (c_light_ * erad_L);
						double Fx_R = fx_R * (c_light_ * erad_R);
						double Fy_L = fy_L * (c_light_ * erad_L);
						double Fy_R = fy_R * (c_light_ * erad_R);
						double Fz_L = fz_L * (c_light_ * erad_L);
						double Fz_R = fz_R * (c_light_ * erad_R);

						// :1054-1079 first-order fallback
						if ((erad_L <= 0.) || (erad_R <= 0.) || (f_L >= 1.) || (f_R >= 1.)) {
							erad_L = consVar(i - 1, j, k, radEnergy_index() + pg);
							erad_R = consVar(i, j, k, radEnergy_index() + pg);
							Fx_L = consVar(i - 1, j, k, x1RadFlux_index() + pg);
							Fx_R = consVar(i, j, k, x1RadFlux_index() + pg);
							Fy_L = consVar(i - 1, j, k, x2RadFlux_index() + pg);
							Fy_R = consVar(i, j, k, x2RadFlux_index() + pg);
							Fz_L = consVar(i - 1, j, k, x3RadFlux_index() + pg);
							Fz_R = consVar(i, j, k, x3RadFlux_index() + pg);
							fx_L = Fx_L / (c_light_ * erad_L);
							fx_R = Fx_R / (c_light_ * erad_R);
							fy_L = Fy_L / (c_light_ * erad_L);
							fy_R = Fy_R / (c_light_ * erad_R);
							fz_L = Fz_L / (c_light_ * erad_L);
							fz_R = Fz_R / (c_light_ * erad_R);
							f_L = std::sqrt(fx_L * fx_L + fy_L * fy_L + fz_L * fz_L);
							f_R = std::sqrt(fx_R * fx_R + fy_R * fy_R + fz_R * fz_R);
						}

						auto [F_L, S_L] = ComputeRadPressure(dir, erad_L, Fx_L, Fy_L, Fz_L, fx_L, fy_L, fz_L);
						S_L *= -1.;
						auto [F_R, S_R] = ComputeRadPressure(dir, erad_R, Fx_R, Fy_R, Fz_R, fx_R, fy_R, fz_R);

						// :1087-1094
						F_L[0] *= c_hat_ / c_light_;
						F_R[0] *= c_hat_ / c_light_;
						for (int n = 1; n < kNumRadVars; ++n) {
							F_L[n] *= c_hat_ * c_light_;
							F_R[n] *= c_hat_ * c_light_;
						}
						S_L *= c_hat_;
						S_R *= c_hat_;

						const std::array<double, 4> U_L = {erad_L, Fx_L, Fy_L, Fz_L};
						const std::array<double, 4> U_R = {erad_R, Fx_R, Fy_R, Fz_R};
						std::array<double, 4> epsilon = {1.0, 1.0, 1.0, 1.0};
						if (use_wavespeed_correction) { // :1103-1109
							// no correction for odd zones
							if ((i + j + k) % 2 == 0) {
								const double S_corr = std::min(1.0, 1.0 / tau_cell[g]); // Skinner et al.
								epsilon = {S_corr, 1.0, 1.0, 1.0};
							}
						}

						// :1116-1117, :1130-1131
						for (int n = 0; n < kNumRadVars; ++n) {
							const double F = (S_R / (S_R - S_L)) * F_L[n] - (S_L / (S_R - S_L)) * F_R[n] +
									 epsilon[n] * (S_R * S_L / (S_R - S_L)) * (U_R[n] - U_L[n]);
							const double diffusiveF =
							    (S_R / (S_R - S_L)) * F_L[n] - (S_L / (S_R - S_L)) * F_R[n] + (S_R * S_L / (S_R - S_L)) * (U_R[n] - U_L[n]);
							x1Flux(i, j, k, pg + n) = F;
							x1FluxDiffusive(i, j, k, pg + n) = diffusiveF;
						}
					}
				}
			}
		}
	}

	// :667-710
	void PredictStep(Array4<const double> const &consVarOld, Array4<double> const &consVarNew, std::array<Array4<const double>, 3> const &fluxArray,
			 const double dt, double const dx_in[3], Box const &indexRange) const
	{
		for (int k = indexRange.lo[2]; k <= indexRange.hi[2]; ++k) {
			for (int j = indexRange.lo[1]; j <= indexRange.hi[1]; ++j) {
				for (int i = indexRange.lo[0]; i <= indexRange.hi[0]; ++i) {
					RadCons cons{};
					for (int n = 0; n < nRadComps(); ++n) {
						double d = (dt / dx_in[0]) * (fluxArray[0](i, j, k, n) - fluxArray[0](i + 1, j, k, n));
						if (ndim >= 2) {
							d = d + (dt / dx_in[1]) * (fluxArray[1](i, j, k, n) - fluxArray[1](i, j + 1, k, n));
						}
						if (ndim == 3) {
							d = d + (dt / dx_in[2]) * (fluxArray[2](i, j, k, n) - fluxArray[2](i, j, k + 1, n));
						}
						cons[n] = consVarOld(i, j, k, nstartHyperbolic_ + n) + d;
					}
					if (!isStateValid(cons)) {
						amendRadState(cons);
					}
					for (int n = 0; n < nRadComps(); ++n) {
						consVarNew(i, j, k, nstartHyperbolic_ + n) = cons[n];
					}
				}
			}
		}
	}

	// :712-771
	void AddFluxesRK2(Array4<double> const &U_new, Array4<const double> const &U0, Array4<const double> const &U1,
			  std::array<Array4<const double>, 3> const &fluxArrayOld, std::array<Array4<const double>, 3> const &fluxArray, const double dt,
			  double const dx_in[3], Box const &indexRange) const
	{
		for (int k = indexRange.lo[2]; k <= indexRange.hi[2]; ++k) {
			for (int j = indexRange.lo[1]; j <= indexRange.hi[1]; ++j) {
				for (int i = indexRange.lo[0]; i <= indexRange.hi[0]; ++i) {
					RadCons cons_new{};
					for (int n = 0; n < nRadComps(); ++n) {
						const double U_0 = U0(i, j, k, nstartHyperbolic_ + n);
						const double U_1 = U1(i, j, k, nstartHyperbolic_ + n);
						double s0 = (dt / dx_in[0]) * (fluxArrayOld[0](i, j, k, n) - fluxArrayOld[0](i + 1, j, k, n));
						double s1 = (dt / dx_in[0]) * (fluxArray[0](i, j, k, n) - fluxArray[0](i + 1, j, k, n));
						if (ndim >= 2) {
							s0 = s0 + (dt / dx_in[1]) * (fluxArrayOld[1](i, j, k, n) - fluxArrayOld[1](i, j + 1, k, n));
							s1 = s1 + (dt / dx_in[1]) * (fluxArray[1](i, j, k, n) - fluxArray[1](i, j + 1, k, n));
						}
						if (ndim == 3) {
							s0 = s0 + (dt / dx_in[2]) * (fluxArrayOld[2](i, j, k, n) - fluxArrayOld[2](i, j, k + 1, n));
							s1 = s1 + (dt / dx_in[2]) * (fluxArray[2](i, j, k, n) - fluxArray[2](i, j, k + 1, n));
						}
						// :758-759
						cons_new[n] = (1.0 - IMEX_a32) * U_0 + IMEX_a32 * U_1 + ((0.5 - IMEX_a32) * (s0)) + (0.5 * (s1));
					}
					if (!isStateValid(cons_new)) {
						amendRadState(cons_new);
					}
					for (int n = 0; n < nRadComps(); ++n) {
						U_new(i, j, k, nstartHyperbolic_ + n) = cons_new[n];
					}
				}
			}
		}
	}

	// :560-579
	static void Solve3x3matrix(const double C00, const double C01, const double C02, const double C10, const double C11, const double C12, const double C20,
				   const double C21, const double C22, const double Y0, const double Y1, const double Y2, double &X0, double &X1, double &X2)
	{
		auto E11 = C11 - C01 * C10 / C00;
		auto E12 = C12 - C02 * C10 / C00;
		auto E21 = C21 - C01 * C20 / C00;
		auto E22 = C22 - C02 * C20 / C00;
		auto Z1 = Y1 - Y0 * C10 / C00;
		auto Z2 = Y2 - Y0 * C20 / C00;
		X2 = (Z2 - Z1 * E21 / E11) / (E22 - E12 * E21 / E11);
		X1 = (Z1 - E12 * X2) / E11;
		X0 = (Y0 - C01 * X1 - C02 * X2) / C00;
	}

	// source_terms_single_group.hpp:10-564.  counters: p_iteration_counter[4], p_iteration_failure_counter[3]
	void AddSourceTermsSingleGroup(Array4<double> const &consVar, Array4<const double> const &radEnergySource, Box const &indexRange, double dt_radiation,
				       const int stage, int *p_iteration_counter, int *p_iteration_failure_counter) const
	{
		Array4<const double> consPrev(consVar.p, consVar.box(), consVar.ncomp);
		Array4<double> const &consNew = consVar;
		auto dt = dt_radiation;
		if (stage == 2) {
			dt = (1.0 - IMEX_a32) * dt_radiation;
		}
		const double gamma_ = eos.tr.gamma;
		const int beta_order_ = rt.beta_order;
		const double Erad_floor_ = this->Erad_floor_();
		const double radiation_constant_ = rt.radiation_constant;
		const double c_light_ = rt.c_light;

		for (int k = indexRange.lo[2]; k <= indexRange.hi[2]; ++k) {
			for (int j = indexRange.lo[1]; j <= indexRange.hi[1]; ++j) {
				for (int i = indexRange.lo[0]; i <= indexRange.hi[0]; ++i) {
					const double c = rt.c_light;
					const double chat = rt.c_hat;

					const double rho = consPrev(i, j, k, 0);
					const double x1GasMom0 = consPrev(i, j, k, 1);
					const double x2GasMom0 = consPrev(i, j, k, 2);
					const double x3GasMom0 = consPrev(i, j, k, 3);
					const std::array<double, 3> gasMtm0 = {x1GasMom0, x2GasMom0, x3GasMom0};
					const double Egastot0 = consPrev(i, j, k, 4);
					const double Erad0 = consPrev(i, j, k, radEnergy_index());
					const double Src = radEnergySource(i, j, k, 0) * dt * chat;

					double Egas0 = NAN, Ekin0 = NAN, Etot0 = NAN, Egas_guess = NAN, T_gas = NAN, T_d = NAN;
					double lorentz_factor = NAN, lorentz_factor_v = NAN, lorentz_factor_v_v = NAN;
					double fourPiBoverC = NAN, Erad_guess = NAN, kappaP = NAN, kappaE = NAN, kappaF = NAN, kappaPoverE = NAN;
					double work = 0.0;
					double work_prev = 0.0;
					std::array<double, 3> dMomentum{};
					std::array<double, 3> Frad_t1{};

					const double cscale = c / chat;

					if (gamma_ != 1.0) {
						Egas0 = ComputeEintFromEgas(rho, x1GasMom0, x2GasMom0, x3GasMom0, Egastot0);
						Etot0 = Egas0 + cscale * (Erad0 + Src);
					}

					double gas_update_factor = 1.0;
					if (stage == 1) {
						gas_update_factor = IMEX_a32;
					}

					// :89-98
					double coeff_n = NAN;
					const double H_num_den = rho / eos.tr.mean_molecular_weight; // ComputeNumberDensityH (:463-467)
					if (rt.enable_dust_gas_thermal_coupling_model) {
						coeff_n = dt * rt.dustGasInteractionCoeff * H_num_den * H_num_den / cscale;
					}

					const int max_ite = 5;
					int ite = 0;
					for (; ite < max_ite; ++ite) {
						double R = NAN;
						Erad_guess = Erad0;

						if (gamma_ != 1.0) {
							double tau0 = NAN;
							double tau = NAN;
							Egas_guess = Egas0;
							Ekin0 = Egastot0 - Egas0;

							const double betaSqr =
							    (x1GasMom0 * x1GasMom0 + x2GasMom0 * x2GasMom0 + x3GasMom0 * x3GasMom0) / (rho * rho * c * c);

							if ((beta_order_ == 0) || (beta_order_ == 1)) {
								lorentz_factor = 1.0;
								lorentz_factor_v = 1.0;
							} else if (beta_order_ == 2) {
								lorentz_factor = 1.0 + 0.5 * betaSqr;
								lorentz_factor_v = 1.0;
								lorentz_factor_v_v = 1.0;
							} else if (beta_order_ == 3) {
								lorentz_factor = 1.0 + 0.5 * betaSqr;
								lorentz_factor_v = 1.0 + 0.5 * betaSqr;
								lorentz_factor_v_v = 1.0;
							} else {
								lorentz_factor = 1.0 / std::sqrt(1.0 - betaSqr);
								lorentz_factor_v = lorentz_factor;
								lorentz_factor_v_v = lorentz_factor;
							}

							double F_G = NAN, deltaEgas = NAN, deltaR = NAN, F_D = NAN;
							const double resid_tol = 1.0e-11;
							const int maxIter = 100;
							int n = 0;
							for (; n < maxIter; ++n) {
								T_gas = eos.ComputeTgasFromEint(rho, Egas_guess);
								// dust temperature (:165-175)
								if (!rt.enable_dust_gas_thermal_coupling_model) {
									T_d = T_gas;
								} else {
									T_d = ComputeDustTemperatureBateKeto(T_gas, T_gas, rho, Erad_guess, coeff_n, dt, R, n);
									if (T_d < 0.0) {
										p_iteration_failure_counter[1] += 1;
									}
								}
								fourPiBoverC = ComputeThermalRadiationSingleGroup(T_d);
								kappaP = ComputePlanckOpacity(rho, T_d);
								kappaE = ComputeEnergyMeanOpacity(rho, T_d);
								if (kappaE > 0.0) {
									kappaPoverE = kappaP / kappaE;
								} else {
									kappaPoverE = 1.0;
								}

								if (n == 0) {
									kappaF = ComputeFluxMeanOpacity(rho, T_d);
									if ((beta_order_ != 0) && (include_work_term_in_source)) {
										if (ite == 0) {
											const double frad0 = consPrev(i, j, k, x1RadFlux_index());
											const double frad1 = consPrev(i, j, k, x2RadFlux_index());
											const double frad2 = consPrev(i, j, k, x3RadFlux_index());
											work = (x1GasMom0 * frad0 + x2GasMom0 * frad1 + x3GasMom0 * frad2) *
											       (2.0 * kappaE - kappaF) * chat / (c * c) * lorentz_factor_v * dt;
										}
									}
									tau0 = dt * rho * kappaP * chat * lorentz_factor;
									tau = tau0;
									R = (fourPiBoverC - Erad_guess / kappaPoverE) * tau0 + work;
									tau0 = std::max(tau0, 1.0);
								} else {
									tau = dt * rho * kappaP * chat * lorentz_factor;
									if (tau > 0.0) {
										Erad_guess = kappaPoverE * (fourPiBoverC - (R - work) / tau);
										if (force_rad_floor_in_iteration) {
											if (Erad_guess <= 0.0) {
												Egas_guess -= (c / chat) * (Erad_floor_ - Erad_guess);
												Erad_guess = Erad_floor_;
											}
										}
									}
								}

								double cooling = 0.0;
								double cooling_derivative = 0.0;
								const double CR_heating = crHeatingRate(H_num_den) * dt;
								if (rt.enable_dust_gas_thermal_coupling_model) { // :234-237
									cooling = netCoolingRate0(T_gas, H_num_den);
									cooling_derivative = netCoolingRateTempDerivative0(T_gas, H_num_den);
								}

								F_G = Egas_guess - Egas0 + cscale * R + cooling * dt - CR_heating;
								F_D = Erad_guess - Erad0 - (R + Src);
								double F_D_abs = 0.0;
								if (tau > 0.0) {
									F_D_abs = std::abs(F_D);
								} else {
									F_D_abs = std::abs(F_D + R);
								}

								if ((std::abs(F_G) < resid_tol * Etot0) && (cscale * F_D_abs < resid_tol * Etot0)) {
									break;
								}

								const double c_v = eos.ComputeEintTempDerivative(rho, T_gas);
								const auto d_fourpiboverc_d_t = ComputeThermalRadiationTempDerivativeSingleGroup(T_d);
								auto dEg_dT = kappaPoverE * d_fourpiboverc_d_t;

								double J00 = NAN;
								double J01 = NAN;
								double J10 = NAN;
								double J11 = NAN;
								if (!rt.enable_dust_gas_thermal_coupling_model) {
									J00 = 1.0 + cooling_derivative * dt / c_v;
									J01 = cscale;
									J10 = 1.0 / c_v * dEg_dT - (1 / cscale) * cooling_derivative * dt;
									if (tau <= 0.0) {
										J11 = -std::numeric_limits<double>::infinity();
									} else {
										J11 = -1.0 * kappaPoverE / tau - 1.0;
									}
								} else { // :293-305
									const double LARGE = 1.0e100; // :7
									const double d_Td_d_T = 3. / 2. - T_d / (2. * T_gas);
									dEg_dT *= d_Td_d_T;
									const double dTd_dRg = -1.0 / (coeff_n * std::sqrt(T_gas));
									J00 = 1.0;
									J01 = cscale;
									J10 = 1.0 / c_v * dEg_dT;
									if (tau <= 0.0) {
										J11 = -LARGE;
									} else {
										J11 = kappaPoverE * d_fourpiboverc_d_t * dTd_dRg - kappaPoverE / tau - 1.0;
									}
								}

								const double y0 = -F_G;
								const auto y1 = -1. * F_D;
								const double det = J00 * J11 - J01 * J10;
								deltaEgas = (J11 * y0 - J01 * y1) / det;
								deltaR = (J00 * y1 - J10 * y0) / det;

								if (!enable_dE_constrain) {
									Egas_guess += deltaEgas;
									R += deltaR;
								} else {
									double T_rad = std::sqrt(std::sqrt(Erad_guess / radiation_constant_));
									if (deltaEgas / c_v > std::max(T_gas, T_rad)) {
										Egas_guess = eos.ComputeEintFromTgas(rho, T_rad);
									} else {
										Egas_guess += deltaEgas;
										R += deltaR;
									}
								}
							} // END NEWTON-RAPHSON LOOP

							if (n >= maxIter) {
								p_iteration_failure_counter[0] += 1;
							}
							p_iteration_counter[0] += 1;
							p_iteration_counter[1] += n + 1;
							p_iteration_counter[2] = std::max(p_iteration_counter[2], n + 1);

							if (!add_line_cooling_to_radiation_in_jac) {
								const auto cooling_tend = netCoolingRate0(T_gas, H_num_den) * dt; // :351-356
								Erad_guess += (1 / cscale) * cooling_tend;
							}
							if (n > 0) {
								kappaF = ComputeFluxMeanOpacity(rho, T_d);
							}
						} else { // gamma_ == 1.0
							T_d = T_gas;
							kappaF = ComputeFluxMeanOpacity(rho, T_d);
						}

						// 2. radiation flux update (:385-481)
						std::array<double, 3> Frad_t0{};
						dMomentum = {0., 0., 0.};
						Frad_t0[0] = consPrev(i, j, k, x1RadFlux_index());
						Frad_t0[1] = consPrev(i, j, k, x2RadFlux_index());
						Frad_t0[2] = consPrev(i, j, k, x3RadFlux_index());

						if ((gamma_ != 1.0) && (beta_order_ != 0)) {
							auto erad = Erad_guess;
							std::array<double, 3> gasVel{};
							std::array<double, 3> v_terms{};
							auto fx = Frad_t0[0] / (c_light_ * erad);
							auto fy = Frad_t0[1] / (c_light_ * erad);
							auto fz = Frad_t0[2] / (c_light_ * erad);
							const double F_coeff = chat * rho * kappaF * dt * lorentz_factor;
							auto Tedd = ComputeEddingtonTensor(fx, fy, fz);

							for (int n = 0; n < 3; ++n) {
								double Planck_term = kappaP * fourPiBoverC * lorentz_factor_v;
								if (kappaF != kappaE) {
									Planck_term += (kappaF - kappaE) * erad * std::pow(lorentz_factor_v, 3);
								}
								Planck_term *= chat * dt * gasMtm0[n];
								double pressure_term = 0.0;
								for (int z = 0; z < 3; ++z) {
									pressure_term += gasMtm0[z] * Tedd[n][z] * erad;
								}
								pressure_term *= chat * dt * kappaF * lorentz_factor_v;
								v_terms[n] = Planck_term + pressure_term;
							}

							if (beta_order_ == 1) {
								for (int n = 0; n < 3; ++n) {
									Frad_t1[n] = (Frad_t0[n] + v_terms[n]) / (1.0 + F_coeff);
									dMomentum[n] += -(Frad_t1[n] - Frad_t0[n]) / (c * chat);
								}
							} else {
								if (kappaF == kappaE) {
									for (int n = 0; n < 3; ++n) {
										Frad_t1[n] = (Frad_t0[n] + v_terms[n]) / (1.0 + F_coeff);
										dMomentum[n] += -(Frad_t1[n] - Frad_t0[n]) / (c * chat);
									}
								} else {
									const double K0 =
									    2.0 * rho * chat * dt * (kappaF - kappaE) / c / c * std::pow(lorentz_factor_v_v, 3);
									const double A00 = 1.0 + F_coeff + K0 * gasVel[0] * gasVel[0];
									const double A01 = K0 * gasVel[0] * gasVel[1];
									const double A02 = K0 * gasVel[0] * gasVel[2];
									const double A10 = K0 * gasVel[1] * gasVel[0];
									const double A11 = 1.0 + F_coeff + K0 * gasVel[1] * gasVel[1];
									const double A12 = K0 * gasVel[1] * gasVel[2];
									const double A20 = K0 * gasVel[2] * gasVel[0];
									const double A21 = K0 * gasVel[2] * gasVel[1];
									const double A22 = 1.0 + F_coeff + K0 * gasVel[2] * gasVel[2];
									const double B0 = v_terms[0] + Frad_t0[0];
									const double B1 = v_terms[1] + Frad_t0[1];
									const double B2 = v_terms[2] + Frad_t0[2];
									double sol0 = NAN, sol1 = NAN, sol2 = NAN;
									Solve3x3matrix(A00, A01, A02, A10, A11, A12, A20, A21, A22, B0, B1, B2, sol0, sol1, sol2);
									Frad_t1[0] = sol0;
									Frad_t1[1] = sol1;
									Frad_t1[2] = sol2;
									for (int n = 0; n < 3; ++n) {
										dMomentum[n] += -(Frad_t1[n] - Frad_t0[n]) / (c * chat);
									}
								}
							}
						} else {
							for (int n = 0; n < 3; ++n) {
								Frad_t1[n] = Frad_t0[n] / (1.0 + rho * kappaF * chat * dt);
								dMomentum[n] += -(Frad_t1[n] - Frad_t0[n]) / (c * chat);
							}
						}

						double const x1GasMom1 = consPrev(i, j, k, 1) + dMomentum[0];
						double const x2GasMom1 = consPrev(i, j, k, 2) + dMomentum[1];
						double const x3GasMom1 = consPrev(i, j, k, 3) + dMomentum[2];

						// 3. work term (:488-514)
						if ((gamma_ != 1.0) && (beta_order_ != 0)) {
							double const Egastot1 = ComputeEgasFromEint(rho, x1GasMom1, x2GasMom1, x3GasMom1, Egas_guess);
							double const Ekin1 = Egastot1 - Egas_guess;
							double const dEkin_work = Ekin1 - Ekin0;
							if (include_work_term_in_source) {
								Egas_guess -= dEkin_work;
							}
						}

						if ((beta_order_ == 0) || (gamma_ == 1.0) || (!include_work_term_in_source)) {
							break;
						}
						work_prev = work;
						work = (x1GasMom1 * Frad_t1[0] + x2GasMom1 * Frad_t1[1] + x3GasMom1 * Frad_t1[2]) * chat / (c * c) * lorentz_factor_v *
						       (2.0 * kappaE - kappaF) * dt;
						const double lag_tol = 1.0e-13;
						if ((std::abs(work) == 0.0) || (cscale * std::abs(work - work_prev) < lag_tol * Etot0) ||
						    (std::abs(work - work_prev) <= lag_tol * R) || (std::abs(work - work_prev) <= 1.0e-8 * std::abs(work))) {
							break;
						}
					} // end full-step iteration

					if (ite >= max_ite) {
						p_iteration_failure_counter[2] += 1;
					}

					// 4b. store (:534-563)
					const auto x1GasMom1 = consPrev(i, j, k, 1) + dMomentum[0] * gas_update_factor;
					const auto x2GasMom1 = consPrev(i, j, k, 2) + dMomentum[1] * gas_update_factor;
					const auto x3GasMom1 = consPrev(i, j, k, 3) + dMomentum[2] * gas_update_factor;
					consNew(i, j, k, 1) = x1GasMom1;
					consNew(i, j, k, 2) = x2GasMom1;
					consNew(i, j, k, 3) = x3GasMom1;
					if (gamma_ != 1.0) {
						Egas_guess = Egas0 + (Egas_guess - Egas0) * gas_update_factor;
						consNew(i, j, k, 5) = Egas_guess;
						consNew(i, j, k, 4) = ComputeEgasFromEint(rho, x1GasMom1, x2GasMom1, x3GasMom1, Egas_guess);
						consNew(i, j, k, radEnergy_index()) = Erad_guess;
					}
					consNew(i, j, k, x1RadFlux_index()) = Frad_t1[0];
					consNew(i, j, k, x2RadFlux_index()) = Frad_t1[1];
					consNew(i, j, k, x3RadFlux_index()) = Frad_t1[2];
				}
			}
		}
	}
};

} // namespace oracle

#endif // ORACLE_RADIATION_HPP_
