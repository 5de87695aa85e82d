// quokka_rad_system.hpp — part 2 of the C++17 host mirror: RadSystem_Traits / ISM_Traits, RadSystem<problem_t> — indices, constants, the opacity / emission /
//   closure hooks a problem specialises, the radiation operators (reference src/radiation/radiation_system.hpp); the single- and multigroup source-term kernels
//   are instantiated in the problem's translation unit with its compiled hooks (qk_problem_kernels.hpp).
#ifndef QK_HOST_QUOKKA_RAD_SYSTEM_HPP_
#define QK_HOST_QUOKKA_RAD_SYSTEM_HPP_

#include "quokka_hydro_system.hpp"

// physical constants in CGS units, as problem files name them (reference src/radiation/radiation_system.hpp:58-61)
static constexpr double c_light_cgs_ = C::c_light;
static constexpr double radiation_constant_cgs_ = C::a_rad;
static constexpr double inf = std::numeric_limits<double>::max();

// this struct is specialized by the user application code (reference src/radiation/radiation_system.hpp:73-82)
// radiation_system.hpp:63-70
enum class OpacityModel { single_group = 0, piecewise_constant_opacity, PPL_opacity_fixed_slope_spectrum, PPL_opacity_full_spectrum };

// radiation_system.hpp:86-90
template <typename problem_t> struct ISM_Traits {
	static constexpr bool enable_dust_gas_thermal_coupling_model = false;
	static constexpr bool enable_photoelectric_heating = false;
	static constexpr double gas_dust_coupling_threshold = 1.0e-6;
};

template <typename problem_t> struct RadSystem_Traits {
	static constexpr double c_light = c_light_cgs_;
	static constexpr double c_hat = c_light_cgs_;
	static constexpr double radiation_constant = radiation_constant_cgs_;
	static constexpr double Erad_floor = 0.;
	static constexpr double energy_unit = C::ev2erg;
	static constexpr amrex::GpuArray<double, Physics_Traits<problem_t>::nGroups + 1> radBoundaries = {0., inf};
	static constexpr double beta_order = 1;
	static constexpr OpacityModel opacity_model = OpacityModel::single_group;
};

// members a specialisation of RadSystem_Traits may leave out (reference radiation_system.hpp:147-154 does this for opacity_model)
namespace qkhost
{
template <typename P, typename = void> struct RadHasOpacityModel : std::false_type {
};
template <typename P> struct RadHasOpacityModel<P, std::void_t<decltype(RadSystem_Traits<P>::opacity_model)>> : std::true_type {
};
template <typename P, typename = void> struct RadHasEnergyUnit : std::false_type {
};
template <typename P> struct RadHasEnergyUnit<P, std::void_t<decltype(RadSystem_Traits<P>::energy_unit)>> : std::true_type {
};
template <typename P> constexpr auto radEnergyUnit() -> double
{
	if constexpr (RadHasEnergyUnit<P>::value) {
		return RadSystem_Traits<P>::energy_unit;
	} else {
		return C::ev2erg;
	}
}
} // namespace qkhost
template <typename problem_t> using RadSystem_Has_Opacity_Model = qkhost::RadHasOpacityModel<problem_t>;

// RadSystem<problem_t>: indices, constants, the problem's device hooks and the operators of the radiation update, each ONE
// call into the C-ABI (reference src/radiation/radiation_system.hpp:150-330).  Single group, OpacityModel::single_group.
template <typename problem_t> class RadSystem;
#include "qk_problem_kernels.hpp" // the source-term kernel instantiated with this problem's compiled hooks

template <typename problem_t> class RadSystem : public HyperbolicSystem<problem_t>
{
      public:
	using array_t = amrex::Array4<amrex::Real>;
	using arrayconst_t = amrex::Array4<const amrex::Real>;
	enum gasVarIndex { gasDensity_index = 0, x1GasMomentum_index, x2GasMomentum_index, x3GasMomentum_index, gasEnergy_index, gasInternalEnergy_index, scalar0_index };
	static constexpr int nvarHyperbolic_ = Physics_NumVars::numRadVars * Physics_Traits<problem_t>::nGroups;
	static constexpr int nstartHyperbolic_ = Physics_Indices<problem_t>::radFirstIndex;
	static constexpr int nvar_ = nstartHyperbolic_ + nvarHyperbolic_;
	enum radVarIndex { radEnergy_index = nstartHyperbolic_, x1RadFlux_index, x2RadFlux_index, x3RadFlux_index };

	static constexpr double c_light_ = RadSystem_Traits<problem_t>::c_light;
	static constexpr double c_hat_ = RadSystem_Traits<problem_t>::c_hat;
	static constexpr double radiation_constant_ = RadSystem_Traits<problem_t>::radiation_constant;
	static constexpr int beta_order_ = static_cast<int>(RadSystem_Traits<problem_t>::beta_order);
	static constexpr int numRadVars_ = Physics_NumVars::numRadVars;
	static constexpr int nmscalars_ = Physics_Traits<problem_t>::numMassScalars;
	enum primVarIndex { primRadEnergy_index = 0, x1ReducedFlux_index, x2ReducedFlux_index, x3ReducedFlux_index };

	// :195-231
	static constexpr bool enable_dust_gas_thermal_coupling_model_ = ISM_Traits<problem_t>::enable_dust_gas_thermal_coupling_model;
	static constexpr bool enable_photoelectric_heating_ = ISM_Traits<problem_t>::enable_photoelectric_heating;
	static constexpr int nGroups_ = Physics_Traits<problem_t>::nGroups;
	static constexpr amrex::GpuArray<double, nGroups_ + 1> radBoundaries_ = []() constexpr {
		if constexpr (nGroups_ > 1) {
			return RadSystem_Traits<problem_t>::radBoundaries;
		} else {
			amrex::GpuArray<double, 2> boundaries{0., inf};
			return boundaries;
		}
	}();
	static constexpr double Erad_floor_ = RadSystem_Traits<problem_t>::Erad_floor / nGroups_;
	static constexpr OpacityModel opacity_model_ = []() constexpr {
		if constexpr (RadSystem_Has_Opacity_Model<problem_t>::value) {
			return RadSystem_Traits<problem_t>::opacity_model;
		} else {
			return OpacityModel::single_group;
		}
	}();
	static_assert(((nGroups_ > 1 && opacity_model_ != OpacityModel::single_group) || (nGroups_ == 1 && opacity_model_ == OpacityModel::single_group)),
		      "OpacityModel::single_group MUST be used when nGroups_ == 1. If nGroups_ > 1, you MUST set opacity_model.");
	static_assert(!(nGroups_ < 3 && opacity_model_ == OpacityModel::PPL_opacity_full_spectrum), "PPL_opacity_full_spectrum requires at least 3 photon groups.");
	static constexpr double mean_molecular_mass_ = quokka::EOS_Traits<problem_t>::mean_molecular_weight;
	static constexpr double boltzmann_constant_ = quokka::EOS_Traits<problem_t>::boltzmann_constant;
	static constexpr double gamma_ = quokka::EOS_Traits<problem_t>::gamma;
	static constexpr double energy_unit_ = qkhost::radEnergyUnit<problem_t>();

	// device hooks a problem may specialise (:1141-1167, :471-513, :582-587)
	// multigroup: exponents and lower values of the piecewise power-law opacity at the group edges (default: NaN, :1155-1167)
	AMREX_GPU_HOST_DEVICE static auto DefineOpacityExponentsAndLowerValues(amrex::GpuArray<double, nGroups_ + 1> rad_boundaries, double rho, double Tgas)
	    -> amrex::GpuArray<amrex::GpuArray<double, nGroups_ + 1>, 2>;
	AMREX_GPU_HOST_DEVICE static auto ComputeThermalRadiationSingleGroup(amrex::Real temperature) -> amrex::Real;
	AMREX_GPU_HOST_DEVICE static auto ComputeThermalRadiationTempDerivativeSingleGroup(amrex::Real temperature) -> amrex::Real;
	// The Planck-spectrum members problem files call for initial and boundary states: each forwards to the body the HIP library's multigroup kernels
	// use themselves (csrc/qk_planck.hpp; host + device), with std::pow where the reference's members use it.
	// :430-461: energy fractions of a Planck spectrum in the groups
	AMREX_GPU_HOST_DEVICE static auto ComputePlanckEnergyFractions(amrex::GpuArray<double, nGroups_ + 1> const &boundaries, amrex::Real temperature)
	    -> quokka::valarray<amrex::Real, nGroups_>
	{
		quokka::valarray<amrex::Real, nGroups_> out{};
		qk::planck::groupFractions<nGroups_>(boundaries.data(), energy_unit_ / (boltzmann_constant_ * temperature), qk::planck::LocalTable{}, &out[0]);
		return out;
	}
	// :483-497: a T^4 in the groups, floored
	AMREX_GPU_HOST_DEVICE static auto ComputeThermalRadiationMultiGroup(amrex::Real temperature, amrex::GpuArray<double, nGroups_ + 1> const &boundaries)
	    -> quokka::valarray<amrex::Real, nGroups_>
	{
		auto out = ComputePlanckEnergyFractions(boundaries, temperature);
		qk::planck::scaleFloored<nGroups_>(radiation_constant_ * std::pow(temperature, 4), Erad_floor_, &out[0]);
		return out;
	}
	// :505-513: its temperature derivative
	AMREX_GPU_HOST_DEVICE static auto ComputeThermalRadiationTempDerivativeMultiGroup(amrex::Real temperature,
											  amrex::GpuArray<double, nGroups_ + 1> const &boundaries)
	    -> quokka::valarray<amrex::Real, nGroups_>
	{
		auto out = ComputePlanckEnergyFractions(boundaries, temperature);
		qk::planck::scale<nGroups_>(4. * radiation_constant_ * std::pow(temperature, 3), &out[0]);
		return out;
	}
	// :1311-1326 (4 pi B(nu) / c)
	AMREX_GPU_HOST_DEVICE static auto PlanckFunction(const double nu, const double T) -> double
	{
		return qk::planck::spectralDensity(energy_unit_ / (boltzmann_constant_ * T), nu, std::pow(PI, 4) / 15.0, radiation_constant_ * std::pow(T, 4),
						   [](double x) { return std::pow(x, 3); });
	}
	// :1367-1385: the radiation flux of each group in the diffusion limit for gas moving at `vel`
	AMREX_GPU_HOST_DEVICE static auto ComputeFluxInDiffusionLimit(const amrex::GpuArray<double, nGroups_ + 1> rad_boundaries, const double T, const double vel)
	    -> amrex::GpuArray<double, nGroups_>
	{
		amrex::GpuArray<double, nGroups_> out{};
		qk::planck::diffusionLimitFluxes<nGroups_>(rad_boundaries.data(), energy_unit_ / (boltzmann_constant_ * T), gInf, vel * radiation_constant_, std::pow(T, 4),
							   qk::planck::LocalTable{}, [](double x) { return std::pow(x, 3); }, out.data());
		return out;
	}
	// :1354-1365
	AMREX_GPU_HOST_DEVICE static auto ComputeBinCenterOpacity(amrex::GpuArray<double, nGroups_ + 1> rad_boundaries,
								  amrex::GpuArray<amrex::GpuArray<double, nGroups_ + 1>, 2> kappa_expo_and_lower_value)
	    -> quokka::valarray<double, nGroups_>
	{
		quokka::valarray<double, nGroups_> out{};
		qk::planck::binCentreOpacity<nGroups_>(rad_boundaries.data(), kappa_expo_and_lower_value[0].data(), kappa_expo_and_lower_value[1].data(), &out[0]);
		return out;
	}
	// radiation_system.hpp:1289-1308
	AMREX_GPU_HOST_DEVICE static auto ComputeEintFromEgas(double density, double X1GasMom, double X2GasMom, double X3GasMom, double Etot) -> double
	{
		return Etot - qk::planck::kineticEnergy(density, X1GasMom, X2GasMom, X3GasMom);
	}
	AMREX_GPU_HOST_DEVICE static auto ComputeEgasFromEint(double density, double X1GasMom, double X2GasMom, double X3GasMom, double Eint) -> double
	{
		return Eint + qk::planck::kineticEnergy(density, X1GasMom, X2GasMom, X3GasMom);
	}
	AMREX_GPU_HOST_DEVICE static auto ComputePlanckOpacity(double rho, double Tgas) -> amrex::Real;
	// the ISM heating / cooling hooks (radiation_system.hpp:344-353; defaults zero, :524-545 and radiation_dust_system.hpp:7-12)
	AMREX_GPU_HOST_DEVICE static auto DefinePhotoelectricHeatingE1Derivative(amrex::Real temperature, amrex::Real num_density) -> amrex::Real;
	AMREX_GPU_HOST_DEVICE static auto DefineNetCoolingRate(amrex::Real temperature, amrex::Real num_density) -> quokka::valarray<double, nGroups_>;
	AMREX_GPU_HOST_DEVICE static auto DefineNetCoolingRateTempDerivative(amrex::Real temperature, amrex::Real num_density) -> quokka::valarray<double, nGroups_>;
	AMREX_GPU_HOST_DEVICE static auto DefineCosmicRayHeatingRate(amrex::Real num_density) -> double;
	// ... sampled on the host into the closed set of qk_rad_traits (cooling linear in T, the two heating rates constant); aborts otherwise
	static void ismHooks(qk_rad_traits &rt)
	{
		auto close = [](double a, double b) { return a == b || std::abs(a - b) <= 1e-13 * std::abs(b); };
		bool ok = true, any = false;
		auto const c1 = DefineNetCoolingRate(1.0, 1.0);
		double const cr = DefineCosmicRayHeatingRate(1.0);
		double const pe = DefinePhotoelectricHeatingE1Derivative(1.0, 1.0);
		for (int g = 0; g < nGroups_; ++g) {
			rt.cooling_linear_coeff[g] = c1[g];
			any = any || c1[g] != 0.0;
		}
		for (double T : {0.3, 7.0, 4.0e4}) {
			for (double n : {1.0e-3, 1.0, 5.0e7}) {
				auto const c = DefineNetCoolingRate(T, n);
				auto const d = DefineNetCoolingRateTempDerivative(T, n);
				for (int g = 0; g < nGroups_; ++g) {
					ok = ok && close(c[g], c1[g] * T) && close(d[g], c1[g]);
				}
				ok = ok && DefineCosmicRayHeatingRate(n) == cr && DefinePhotoelectricHeatingE1Derivative(T, n) == pe;
			}
		}
		if (!ok) {
			amrex::Abort("RadSystem: the DefineNetCoolingRate / DefineCosmicRayHeatingRate / DefinePhotoelectricHeatingE1Derivative hooks are not in "
				     "the C-ABI's closed set (cooling linear in T, constant heating rates)");
		}
		rt.cr_heating_rate = cr;
		rt.enable_photoelectric_heating = enable_photoelectric_heating_ ? 1 : 0;
		rt.pe_heating_E1_derivative = enable_photoelectric_heating_ ? pe : 0.0;
		if ((any || cr != 0.0 || enable_photoelectric_heating_) && !enable_dust_gas_thermal_coupling_model_) {
			amrex::Abort("RadSystem: the ISM heating / cooling hooks are carried by the C-ABI together with the dust model only");
		}
		if (enable_photoelectric_heating_ && nGroups_ == 1) {
			amrex::Abort("RadSystem: photoelectric heating is a multigroup model (radiation_dust_system.hpp)");
		}
	}
	AMREX_GPU_HOST_DEVICE static auto ComputeFluxMeanOpacity(double rho, double Tgas) -> amrex::Real;
	AMREX_GPU_HOST_DEVICE static auto ComputeEnergyMeanOpacity(double rho, double Tgas) -> amrex::Real;
	AMREX_GPU_HOST_DEVICE static auto ComputeEddingtonFactor(double f) -> double; // :773-790 (default: Levermore closure)
	static void SetRadEnergySource(array_t &radEnergySource, amrex::Box const &indexRange, amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const &dx,
				       amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const &prob_lo, amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const &prob_hi,
				       amrex::Real time);

	// The opacity and closure hooks run on the device in the reference; the C-ABI carries them as a closed, parametrised set
	// (opacity model 0: constants, model 1: kappa = k0 / rho; closure 0: Levermore, 1: chi = 1/3).  The hooks are sampled on the host:
	// anything outside the set is refused, never approximated.
	// closure hook -> closed set (0: Levermore, 1: chi = 1/3); pow_mode from the deck
	static auto closureAndPowMode(int &eddington_model, int &pow_mode) -> void
	{
		bool lev = true, third = true;
		for (double f : {0.0, 0.3, 0.77, 1.0}) {
			double const ff = std::sqrt(4.0 - 3.0 * (f * f));
			lev = lev && ComputeEddingtonFactor(f) == (3.0 + 4.0 * (f * f)) / (5.0 + 2.0 * ff);
			third = third && ComputeEddingtonFactor(f) == (1. / 3.);
		}
		eddington_model = lev ? 0 : (third ? 1 : -1);
		if (eddington_model < 0) {
			amrex::Abort("RadSystem: ComputeEddingtonFactor is neither the Levermore closure nor the Eddington approximation");
		}
		pow_mode = 0;
		amrex::ParmParse pp("radiation");
		pp.query("pow_mode", pow_mode); // 0: pow(T, 4) like the reference's std::pow; 1: repeated multiplication
	}

	// Multigroup: RadSystem_Traits::radBoundaries / energy_unit / opacity_model and the DefineOpacityExponentsAndLowerValues hook sampled on
	// the host into the C-ABI's closed set (exponent per edge; lower value = k_g rho^a (T / 1 K)^b with a in {0, -1}); anything else is refused.
	static auto multigroupTraits() -> qk_rad_traits
	{
		static_assert(nGroups_ <= QK_MAX_GROUPS, "at most QK_MAX_GROUPS photon groups");
		qk_rad_traits rt{};
		rt.c_light = c_light_;
		rt.c_hat = c_hat_;
		rt.radiation_constant = radiation_constant_;
		rt.Erad_floor = RadSystem_Traits<problem_t>::Erad_floor;
		rt.beta_order = beta_order_;
		closureAndPowMode(rt.eddington_model, rt.pow_mode);
		rt.ngroups = nGroups_;
		rt.mg_opacity_model = static_cast<int>(opacity_model_);
		rt.energy_unit = energy_unit_;
		for (int g = 0; g < nGroups_ + 1; ++g) {
			rt.rad_boundaries[g] = radBoundaries_[g];
		}
		const bool pc = (opacity_model_ == OpacityModel::piecewise_constant_opacity); // (its last edge entry is never read and may be unset)
		const int nedge = pc ? nGroups_ : nGroups_ + 1;
		const double r0 = 1.0, T0 = 1.0e3;
		auto const base = DefineOpacityExponentsAndLowerValues(radBoundaries_, r0, T0);
		auto close = [](double a, double b) { return a == b || std::abs(a - b) <= 1e-12 * std::abs(b); };
		// density exponent: 0 or -1; temperature exponent: nearest multiple of 1/2 of the sampled slope
		auto const r2 = DefineOpacityExponentsAndLowerValues(radBoundaries_, 2.0 * r0, T0);
		auto const T2 = DefineOpacityExponentsAndLowerValues(radBoundaries_, r0, 1.0e6);
		int e = 0; // the probe: the first edge with a non-zero lower value (all zero — a transparent medium — is the constant 0)
		while (e < nedge - 1 && base[1][e] == 0.0) {
			++e;
		}
		double a = std::numeric_limits<double>::quiet_NaN();
		if (close(r2[1][e], base[1][e])) {
			a = 0.0;
		} else if (close(r2[1][e], 0.5 * base[1][e])) {
			a = -1.0;
		}
		const double slope = (base[1][e] == 0.0 && T2[1][e] == 0.0) ? 0.0 : std::log(T2[1][e] / base[1][e]) / std::log(1.0e6 / T0);
		const double b = std::round(2.0 * slope) / 2.0;
		bool ok = std::isfinite(a) && std::isfinite(b) && std::abs(slope - b) < 1e-9;
		rt.mg_kappa_rho_exponent = a;
		rt.mg_kappa_T_ref = 1.0;
		rt.mg_kappa_T_exponent = b;
		for (int g = 0; g < nedge && ok; ++g) {
			rt.mg_kappa_exponent[g] = base[0][g];
			rt.mg_kappa_lower[g] = base[1][g] / (std::pow(r0, a) * std::pow(T0, b));
		}
		for (double r : {1.0, 1.0e-24, 3.7e-19, 2.0e-3}) {
			for (double T : {3.0, 1.1e3, 4.0e7}) {
				auto const v = DefineOpacityExponentsAndLowerValues(radBoundaries_, r, T);
				for (int g = 0; g < nedge && ok; ++g) {
					const double expect = (a == -1.0 && b == 0.0) ? rt.mg_kappa_lower[g] / r
									     : (a == 0.0 && b == 0.0) ? rt.mg_kappa_lower[g]
												       : rt.mg_kappa_lower[g] * std::pow(r, a) * std::pow(T, b);
					ok = ok && v[0][g] == base[0][g] && close(v[1][g], expect);
				}
			}
		}
		int force_compiled = 0; // deck `qk.mg_compiled_hook = 1`: the compiled hook also where the closed set would do (tests: same bits)
		amrex::ParmParse("qk").query("mg_compiled_hook", force_compiled);
		if (!ok || force_compiled != 0) {
			// not in the closed set (e.g. exponents that follow the temperature, RadhydroPulseMGint): the hook itself is compiled into the
			// source-term kernel of this translation unit (qk_problem_kernels.hpp: ProblemRadMG); the library's entry refuses this value
			rt.opacity_model = QK_HOOK_COMPILED;
			rt.mg_kappa_rho_exponent = 0.0;
			rt.mg_kappa_T_exponent = 0.0;
			for (int g = 0; g < nGroups_ + 1; ++g) {
				rt.mg_kappa_exponent[g] = 0.0;
				rt.mg_kappa_lower[g] = 0.0;
			}
		}
		// the thermal-emission hooks (:483-497, :505-513): the defaults, or RadDustMG's linearised a T / a (test_rad_dust_MG.cpp:83-104)
		{
			bool quartic = true, linear = true;
			for (double T : {0.7, 3.0e2, 4.0e6}) {
				auto const e = ComputeThermalRadiationMultiGroup(T, radBoundaries_);
				auto const d = ComputeThermalRadiationTempDerivativeMultiGroup(T, radBoundaries_);
				auto const f = ComputePlanckEnergyFractions(radBoundaries_, T);
				for (int g = 0; g < nGroups_; ++g) {
					quartic = quartic && e[g] == std::max(radiation_constant_ * std::pow(T, 4) * f[g], Erad_floor_) &&
						  d[g] == 4. * radiation_constant_ * std::pow(T, 3) * f[g];
					linear = linear && e[g] == radiation_constant_ * T * f[g] && d[g] == radiation_constant_ * f[g];
				}
			}
			if (!quartic && !linear) {
				amrex::Abort("RadSystem: the ComputeThermalRadiationMultiGroup hooks are neither a T^4 nor RadDustMG's linearised a T");
			}
			rt.thermal_model = quartic ? 0 : 1;
		}
		if (enable_dust_gas_thermal_coupling_model_) { // radiation_dust_system.hpp; the coefficient is QuokkaSimulation::dustGasInteractionCoeff_
			rt.enable_dust_gas_thermal_coupling_model = 1;
			rt.gas_dust_coupling_threshold = ISM_Traits<problem_t>::gas_dust_coupling_threshold;
			rt.dust_gas_interaction_coeff = 2.5e-34;
			amrex::ParmParse rpp("radiation");
			rpp.query("dust_gas_interaction_coeff", rt.dust_gas_interaction_coeff);
		} else if (rt.thermal_model != 0) {
			amrex::Abort("RadSystem: the linearised thermal-emission hook is carried by the C-ABI together with the dust model only");
		}
		ismHooks(rt);
		return rt;
	}

	static auto traits() -> qk_rad_traits
	{
		if constexpr (nGroups_ > 1) {
			return multigroupTraits();
		}
		// Single group: the opacity hooks are compiled into the source-term kernel of this translation unit (qk_problem_kernels.hpp): nothing to
		// describe to the library, whose transport operators never evaluate an opacity.
		int eddington_model = -1, pow_mode = 0;
		closureAndPowMode(eddington_model, pow_mode);
		double const nan = std::numeric_limits<double>::quiet_NaN();
		qk_rad_traits rt{c_light_, c_hat_, radiation_constant_, Erad_floor_, beta_order_, QK_HOOK_COMPILED, nan, nan, nan, pow_mode, eddington_model, 0.0, 0.0, 0.0};
		// the thermal-emission hooks (:471-479, :499-503): a problem that did not specialise them gets the library's a T^4 (floored) / 4 a T^3,
		// which honours radiation.pow_mode — recognised by exact agreement with the defining formula on probe points; anything else is the
		// problem's compiled hook
		{
			bool quartic = true;
			for (double T : {0.7, 3.0e2, 4.0e6}) {
				const double e = ComputeThermalRadiationSingleGroup(T), d = ComputeThermalRadiationTempDerivativeSingleGroup(T);
				quartic = quartic && e == std::max(radiation_constant_ * std::pow(T, 4), Erad_floor_) && d == 4. * radiation_constant_ * std::pow(T, 3);
			}
			rt.thermal_model = quartic ? 0 : QK_HOOK_COMPILED;
		}
		if (enable_dust_gas_thermal_coupling_model_) { // ISM_Traits; the coefficient is QuokkaSimulation::dustGasInteractionCoeff_ (QuokkaSimulation.hpp:127, :392)
			rt.enable_dust_gas_thermal_coupling_model = 1;
			rt.dust_gas_interaction_coeff = 2.5e-34;
			amrex::ParmParse rpp("radiation");
			rpp.query("dust_gas_interaction_coeff", rt.dust_gas_interaction_coeff);
		}
		return rt; // (the ISM heating / cooling hooks of the single-group source term are compiled as well)
	}
	static auto lev() -> qk_level * { return qkhost::Runtime::get().lev; }
	static void flux3(std::array<amrex::MultiFab, AMREX_SPACEDIM> const &f, qk_array4 *out[3])
	{
		for (int d = 0; d < 3; ++d) {
			out[d] = (d < AMREX_SPACEDIM) ? qkhost::tab(f[d]) : nullptr;
		}
	}
	static void dx3(amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const &dx, double out[3])
	{
		for (int d = 0; d < 3; ++d) {
			out[d] = (d < AMREX_SPACEDIM) ? dx[d] : 1.0;
		}
	}

	// computeRadiationFluxes + fluxFunction<DIR> (reference src/QuokkaSimulation.hpp:1884-1961): cons -> prim, reconstruction, HLL
	// `eps`: nullptr (use_wavespeed_correction false) or the factors ComputeWavespeedCorrection(consVar) left
	static void computeRadiationFluxes(amrex::MultiFab const &consVar, std::array<amrex::MultiFab, AMREX_SPACEDIM> &flux, int reconstructionOrder,
					   std::array<amrex::MultiFab, AMREX_SPACEDIM> const *eps = nullptr)
	{
		auto rt = traits();
		qk_array4 *f[3], *e[3] = {nullptr, nullptr, nullptr};
		flux3(flux, f);
		if (eps != nullptr) {
			flux3(*eps, e);
		}
		qkhost::check(qk_rad_computeRadiationFluxes(lev(), nullptr, &rt, AMREX_SPACEDIM, reconstructionOrder, qkhost::tab(consVar), f, eps != nullptr ? e : nullptr),
			      "RadSystem::computeRadiationFluxes");
	}
	// ComputeCellOpticalDepth<DIR> on every face and what the use_wavespeed_correction branch of ComputeFluxes<DIR> makes of it (:803-871, :1019-1022,
	// :1098-1109): eps[d] = min(1, 1 / tau_cell) on the even faces of direction d, 1 on the odd ones.  Instantiated HERE with this problem's compiled
	// opacity and EOS hooks (qk_problem_kernels.hpp).
	static void ComputeWavespeedCorrection(amrex::MultiFab const &consVar, amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> dx,
					       std::array<amrex::MultiFab, AMREX_SPACEDIM> &eps)
	{
		auto rt = traits();
		auto t = qkhost::traits<problem_t>();
		qk_array4 *e[3];
		double d3[3];
		flux3(eps, e);
		dx3(dx, d3);
		qkhost::check(qkhost::computeWavespeedCorrection<problem_t>(lev(), &rt, &t, qkhost::tab(consVar), d3, e), "RadSystem::ComputeCellOpticalDepth");
	}
	// :667-710
	static void PredictStep(amrex::MultiFab const &consVarOld, amrex::MultiFab &consVarNew, std::array<amrex::MultiFab, AMREX_SPACEDIM> const &fluxArray,
				double dt, amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> dx)
	{
		auto rt = traits();
		qk_array4 *f[3];
		double d3[3];
		flux3(fluxArray, f);
		dx3(dx, d3);
		qkhost::check(qk_rad_PredictStep(lev(), nullptr, &rt, AMREX_SPACEDIM, qkhost::tab(consVarOld), qkhost::tab(consVarNew), f, dt, d3),
			      "RadSystem::PredictStep");
	}
	// one transport stage with the flux divergence taken inside the flux kernels (qk_rad_stage_fused): computeRadiationFluxes(U_in) +
	// PredictStep (stage 1) / AddFluxesRK2 (stage 2); `fluxOut`: where the face fluxes are stored, or nullptr when nothing reads them
	static void stageFused(int stage, int order, amrex::MultiFab const &U_in, amrex::MultiFab const &U0, amrex::MultiFab &U_new, amrex::MultiFab &acc,
			       std::array<amrex::MultiFab, AMREX_SPACEDIM> *fluxOut, double dt, amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> dx,
			       std::array<amrex::MultiFab, AMREX_SPACEDIM> const *eps = nullptr)
	{
		auto rt = traits();
		qk_array4 *f[3] = {nullptr, nullptr, nullptr}, *e[3] = {nullptr, nullptr, nullptr};
		double d3[3];
		if (fluxOut != nullptr) {
			flux3(*fluxOut, f);
		}
		if (eps != nullptr) {
			flux3(*eps, e);
		}
		dx3(dx, d3);
		qkhost::check(qk_rad_stage_fused(lev(), nullptr, &rt, order, stage, qkhost::tab(U_in), qkhost::tab(U0), qkhost::tab(U_new), qkhost::tab(acc),
						 fluxOut != nullptr ? f : nullptr, dt, d3, eps != nullptr ? e : nullptr),
			      "RadSystem::stageFused");
	}
	// :712-771
	static void AddFluxesRK2(amrex::MultiFab &U_new, amrex::MultiFab const &U0, amrex::MultiFab const &U1,
				 std::array<amrex::MultiFab, AMREX_SPACEDIM> const &fluxArrayOld, std::array<amrex::MultiFab, AMREX_SPACEDIM> const &fluxArray, double dt,
				 amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> dx)
	{
		auto rt = traits();
		qk_array4 *f0[3], *f1[3];
		double d3[3];
		flux3(fluxArrayOld, f0);
		flux3(fluxArray, f1);
		dx3(dx, d3);
		qkhost::check(qk_rad_AddFluxesRK2(lev(), nullptr, &rt, AMREX_SPACEDIM, qkhost::tab(U_new), qkhost::tab(U0), qkhost::tab(U1), f0, f1, dt, d3),
			      "RadSystem::AddFluxesRK2");
	}
	// src/radiation/source_terms_single_group.hpp:10-564
	// mirror (an extension of this host): the new radiation components of the valid cells are stored there too — the swapRadiationState() of the
	// next substep from the registers of this kernel (include/quokka_amd.h: qk_rad_AddSourceTermsSingleGroupMirror)
	static void AddSourceTermsSingleGroup(amrex::MultiFab &consVar, amrex::MultiFab const &radEnergySource, double dt, int stage, int *p_iteration_counter,
					      int *p_iteration_failure_counter, amrex::MultiFab *mirror = nullptr)
	{
		auto rt = traits();
		auto t = qkhost::traits<problem_t>();
		// the kernel is instantiated HERE, with this problem's compiled opacity / emission / EOS / ISM hooks (qk_problem_kernels.hpp)
		qkhost::check(qkhost::addSourceTermsSingleGroup<problem_t>(lev(), &rt, &t, qkhost::tab(consVar), qkhost::tab(radEnergySource), dt, stage,
									    p_iteration_counter, p_iteration_failure_counter,
									    mirror != nullptr ? qkhost::tab(*mirror) : nullptr),
			      "RadSystem::AddSourceTermsSingleGroup");
	}
	// src/radiation/source_terms_multi_group.hpp:522-813
	static void AddSourceTermsMultiGroup(amrex::MultiFab &consVar, amrex::MultiFab const &radEnergySource, double dt, int stage, int *p_iteration_counter,
					     int *p_iteration_failure_counter)
	{
		auto rt = traits();
		auto t = qkhost::traits<problem_t>();
		if (rt.opacity_model == QK_HOOK_COMPILED) { // the kernel is instantiated HERE, with this problem's DefineOpacityExponentsAndLowerValues
			qkhost::check(qkhost::addSourceTermsMultiGroup<problem_t>(lev(), &rt, &t, qkhost::tab(consVar), qkhost::tab(radEnergySource), dt, stage,
										   p_iteration_counter, p_iteration_failure_counter),
				      "RadSystem::AddSourceTermsMultiGroup (compiled opacity hook)");
			return;
		}
		qkhost::check(qk_rad_AddSourceTermsMultiGroup(lev(), nullptr, &rt, &t, qkhost::tab(consVar), qkhost::tab(radEnergySource), dt, stage,
							      p_iteration_counter, p_iteration_failure_counter),
			      "RadSystem::AddSourceTermsMultiGroup");
	}
};

// the defaults of the hooks a problem may specialise (reference radiation_system.hpp:471-479, :499-503, :524-545, :773-790, :1141-1167), each in terms of the
// bodies the HIP library shares with this mirror (csrc/qk_planck.hpp)
#define QK_RAD_HOOK template <typename problem_t> AMREX_GPU_HOST_DEVICE auto RadSystem<problem_t>::
QK_RAD_HOOK ComputeThermalRadiationSingleGroup(amrex::Real temperature) -> amrex::Real
{
	double e = 1.0; // a T^4, floored: the one-group case of ComputeThermalRadiationMultiGroup
	qk::planck::scaleFloored<1>(radiation_constant_ * std::pow(temperature, 4), Erad_floor_, &e);
	return e;
}
QK_RAD_HOOK ComputeThermalRadiationTempDerivativeSingleGroup(amrex::Real temperature) -> amrex::Real { return 4. * radiation_constant_ * std::pow(temperature, 3); }
QK_RAD_HOOK DefineOpacityExponentsAndLowerValues(amrex::GpuArray<double, nGroups_ + 1> /*rad_boundaries*/, const double /*rho*/, const double /*Tgas*/)
    -> amrex::GpuArray<amrex::GpuArray<double, nGroups_ + 1>, 2>
{
	amrex::GpuArray<amrex::GpuArray<double, nGroups_ + 1>, 2> undefined{}; // a multigroup problem MUST specialise this hook: NaN everywhere until it does
	for (auto &row : undefined.arr) {
		for (auto &v : row.arr) {
			v = std::numeric_limits<double>::quiet_NaN();
		}
	}
	return undefined;
}
QK_RAD_HOOK DefinePhotoelectricHeatingE1Derivative(amrex::Real const /*temperature*/, amrex::Real const /*num_density*/) -> amrex::Real { return 0.0; }
QK_RAD_HOOK DefineNetCoolingRate(amrex::Real const /*temperature*/, amrex::Real const /*num_density*/) -> quokka::valarray<double, nGroups_>
{
	return quokka::valarray<double, nGroups_>{}; // (value-initialised: no line cooling)
}
QK_RAD_HOOK DefineNetCoolingRateTempDerivative(amrex::Real const /*temperature*/, amrex::Real const /*num_density*/) -> quokka::valarray<double, nGroups_>
{
	return quokka::valarray<double, nGroups_>{};
}
QK_RAD_HOOK DefineCosmicRayHeatingRate(amrex::Real const /*num_density*/) -> double { return 0.0; }
QK_RAD_HOOK ComputePlanckOpacity(const double /*rho*/, const double /*Tgas*/) -> amrex::Real { return std::numeric_limits<double>::quiet_NaN(); }
// (the flux-mean and the energy-mean opacity default to the Planck mean)
QK_RAD_HOOK ComputeFluxMeanOpacity(const double rho, const double Tgas) -> amrex::Real { return ComputePlanckOpacity(rho, Tgas); }
QK_RAD_HOOK ComputeEnergyMeanOpacity(const double rho, const double Tgas) -> amrex::Real { return ComputePlanckOpacity(rho, Tgas); }
QK_RAD_HOOK ComputeEddingtonFactor(double f_in) -> double { return qk::planck::levermoreFactor(f_in); }
#undef QK_RAD_HOOK
template <typename problem_t>
void RadSystem<problem_t>::SetRadEnergySource(array_t & /*radEnergySource*/, amrex::Box const & /*indexRange*/,
					      amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const & /*dx*/,
					      amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const & /*prob_lo*/,
					      amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const & /*prob_hi*/, amrex::Real /*time*/)
{
	// do nothing -- user implemented
}


#endif // QK_HOST_QUOKKA_RAD_SYSTEM_HPP_
