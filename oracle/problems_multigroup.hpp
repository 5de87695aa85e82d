// ORACLE — TEST INFRASTRUCTURE ONLY (see grid.hpp header).
//
// problems_multigroup.hpp: the reference's multigroup radiation test problems, restated on the oracle's driver (included by problems.hpp)
//   src/problems/RadhydroShockMultigroup/test_radhydro_shock_multigroup.cpp
//   src/problems/RadTube/test_radiation_tube.cpp
//   src/problems/RadMarshakVaytet/test_radiation_marshak_Vaytet.cpp
//   src/problems/RadhydroPulseMGconst/test_radhydro_pulse_MG_const_kappa.cpp
//   src/problems/RadDust/test_rad_dust.cpp (single group, dust-gas thermal coupling)
// and the 2-D hydro pin src/problems/HydroQuirk/test_quirk.cpp
#ifndef ORACLE_PROBLEMS_MULTIGROUP_HPP_
#define ORACLE_PROBLEMS_MULTIGROUP_HPP_

namespace oracle
{

inline void setRadGroups(HydroSim &sim, std::vector<double> const &boundaries, double energy_unit, int opacity_model)
{
	sim.rad.rt.nGroups = static_cast<int>(boundaries.size()) - 1;
	sim.rad.rt.radBoundaries = boundaries;
	sim.rad.rt.energy_unit = energy_unit;
	sim.rad.rt.opacity_model = opacity_model;
	sim.ncomp_cc = kNumHydroVars + kNumRadVars * sim.rad.rt.nGroups;
}

// ---------------------------------------------------------------- multigroup radiative shock (test_radhydro_shock_multigroup.cpp)
struct RadShockMGConstants { // :19-43
	static constexpr double a_rad = C::a_rad;
	static constexpr double c = C::c_light;
	static constexpr double k_B = C::k_B;
	static constexpr double c_s0 = 1.73e7;
	static constexpr double kappa = 577.0; // rho * kappa [cm^-1]
	static constexpr double gamma_gas = (5. / 3.);
	static constexpr double c_v = k_B / ((C::m_p + C::m_e) * (gamma_gas - 1.0));
	static constexpr double T0 = 2.18e6, rho0 = 5.69, v0 = 5.19e7;
	static constexpr double T1 = 7.98e6, rho1 = 17.1, v1 = 1.73e7;
	static constexpr double chat = 10.0 * (v0 + c_s0);
	static constexpr double Erad0 = a_rad * (T0 * T0 * T0 * T0);
	static constexpr double Erad_floor_ = Erad0 * 1e-12;
	static constexpr double Egas0 = rho0 * c_v * T0;
	static constexpr double Egas1 = rho1 * c_v * T1;
	static constexpr double shock_position = 0.0130;
	static constexpr double Lx = 0.01575;
};

// opacity_model: the problem file selects PPL_opacity_fixed_slope_spectrum (:67); the two alternatives it lists (:66, :68) are accepted
inline void setupRadShockMG(HydroSim &sim, int opacity_model = PPL_opacity_fixed_slope_spectrum)
{
	using S = RadShockMGConstants;
	sim.hydro.tr.eos.tr.gamma = S::gamma_gas; // :71-75
	sim.hydro.tr.eos.tr.mean_molecular_weight = C::m_p + C::m_e;
	sim.hydro.tr.eos.tr.boltzmann_constant = S::k_B;
	sim.hydro.tr.reconstruct_eint = true;
	sim.hydro.tr.nscalars = 0;
	sim.is_radiation_enabled = true;
	sim.rad.rt.c_light = S::c; // :56-69
	sim.rad.rt.c_hat = S::chat;
	sim.rad.rt.radiation_constant = S::a_rad;
	sim.rad.rt.Erad_floor = S::Erad_floor_;
	sim.rad.rt.beta_order = 1;
	sim.rad.rt.eddington_model = 1; // :91-94
	setRadGroups(sim, {1.00000000e+15, 1.00000000e+16, 1.00000000e+17, 1.00000000e+18, 1.00000000e+19, 1.00000000e+20}, C::hplanck, opacity_model);
	sim.rad.eos = sim.hydro.tr.eos;
	sim.rad.ndim = sim.geom.ndim;
	sim.rad.nstartHyperbolic_ = kNumHydroVars;
	const int ng = sim.rad.rt.nGroups;
	// :77-89
	sim.rad.DefineOpacityExponentsAndLowerValues = [ng](double const *, double rho, double, double *expo, double *lower) {
		for (int i = 0; i < ng + 1; ++i) {
			expo[i] = 0.0;
			lower[i] = S::kappa / rho;
		}
	};

	// problem_main :221-247
	sim.BCs_cc.assign(sim.ncomp_cc, BCRec{});
	for (int n = 0; n < sim.ncomp_cc; ++n) {
		sim.BCs_cc[n].lo[0] = ext_dir;
		sim.BCs_cc[n].hi[0] = ext_dir;
	}
	sim.radiationReconstructionOrder_ = 3;
	sim.reconstructionOrder_ = 3;
	sim.cflNumber_ = 0.4;
	sim.radiationCflNumber_ = 0.4;
	sim.maxTimesteps_ = 20000;
	sim.stopTime_ = 1.0e-9;

	HydroSim *const simp = &sim;
	// setCustomBoundaryConditions :96-162
	sim.customBC = [simp, ng](int i, int j, int k, Array4<double> const &consVar, Box const &dom, double /*time*/) {
		mg::MG const m(simp->rad);
		auto fill = [&](double rho, double v, double Egas, double T) {
			const double px = rho * v;
			auto Erad_g = m.ComputeThermalRadiationMultiGroup(T, m.boundaries());
			consVar(i, j, k, density_index) = rho;
			consVar(i, j, k, x1Momentum_index) = px;
			consVar(i, j, k, x2Momentum_index) = 0.;
			consVar(i, j, k, x3Momentum_index) = 0.;
			consVar(i, j, k, energy_index) = Egas + (px * px) / (2 * rho);
			consVar(i, j, k, internalEnergy_index) = Egas;
			for (int g = 0; g < ng; ++g) {
				consVar(i, j, k, kNumHydroVars + 0 + kNumRadVars * g) = Erad_g[g];
				consVar(i, j, k, kNumHydroVars + 1 + kNumRadVars * g) = 0;
				consVar(i, j, k, kNumHydroVars + 2 + kNumRadVars * g) = 0;
				consVar(i, j, k, kNumHydroVars + 3 + kNumRadVars * g) = 0;
			}
		};
		if (i < dom.lo[0]) {
			fill(S::rho0, S::v0, S::Egas0, S::T0);
		} else if (i >= dom.hi[0]) {
			fill(S::rho1, S::v1, S::Egas1, S::T1);
		}
	};

	sim.define();
	// setInitialConditionsOnGrid :164-219
	Geometry const g = sim.geom;
	mg::MG const m(sim.rad);
	forEachValidCell(sim, [&](Array4<double> const &state_cc, int i, int j, int k) {
		double const x = g.prob_lo[0] + (i + 0.5) * g.dx[0];
		double x1RadFlux = NAN, energy = NAN, density = NAN, x1Momentum = NAN, temp = NAN;
		if (x < S::shock_position) {
			x1RadFlux = 0.0;
			energy = S::Egas0 + 0.5 * S::rho0 * (S::v0 * S::v0);
			density = S::rho0;
			x1Momentum = S::rho0 * S::v0;
			temp = S::T0;
		} else {
			x1RadFlux = 0.0;
			energy = S::Egas1 + 0.5 * S::rho1 * (S::v1 * S::v1);
			density = S::rho1;
			x1Momentum = S::rho1 * S::v1;
			temp = S::T1;
		}
		auto Erad_g = m.ComputeThermalRadiationMultiGroup(temp, m.boundaries());
		state_cc(i, j, k, density_index) = density;
		state_cc(i, j, k, x1Momentum_index) = x1Momentum;
		state_cc(i, j, k, x2Momentum_index) = 0;
		state_cc(i, j, k, x3Momentum_index) = 0;
		state_cc(i, j, k, energy_index) = energy;
		state_cc(i, j, k, internalEnergy_index) = energy - (x1Momentum * x1Momentum) / (2 * density);
		for (int gg = 0; gg < m.nGroups_; ++gg) {
			state_cc(i, j, k, kNumHydroVars + 0 + kNumRadVars * gg) = Erad_g[gg];
			state_cc(i, j, k, kNumHydroVars + 1 + kNumRadVars * gg) = x1RadFlux;
			state_cc(i, j, k, kNumHydroVars + 2 + kNumRadVars * gg) = 0;
			state_cc(i, j, k, kNumHydroVars + 3 + kNumRadVars * gg) = 0;
		}
	});
	sim.finishInitialConditions();
}

// ---------------------------------------------------------------- radiation pressure tube (test_radiation_tube.cpp, deck tests/RadTube.in)
struct RadTubeConstants { // :30-41
	static constexpr double kappa0 = 100.;
	static constexpr double mu = 2.33 * C::m_u;
	static constexpr double gamma_gas = 5. / 3.;
	static constexpr double rho0 = 1.0;
	static constexpr double T0 = 2.75e7;
	static constexpr double rho1 = 2.1940476649492044;
	static constexpr double T1 = 2.2609633884436745e7;
	static constexpr double a_rad = C::a_rad;
	static constexpr double a0 = 4.0295519855200705e7;
	static constexpr double Lx = 128.0;
};

// x, rho, Pgas, Erad columns of extern/pressure_tube/initial_conditions.txt (rows = table_len)
inline void setupRadTube(HydroSim &sim, int table_len, double const *x_tab, double const *rho_tab, double const *Pgas_tab, double const *Erad_tab)
{
	using S = RadTubeConstants;
	sim.hydro.tr.eos.tr.gamma = S::gamma_gas; // :43-47
	sim.hydro.tr.eos.tr.mean_molecular_weight = S::mu;
	sim.hydro.tr.eos.tr.boltzmann_constant = C::k_B;
	sim.hydro.tr.reconstruct_eint = true;
	sim.hydro.tr.nscalars = 0;
	sim.is_radiation_enabled = true;
	sim.rad.rt.c_light = C::c_light; // :61-73
	sim.rad.rt.c_hat = 10.0 * S::a0;
	sim.rad.rt.radiation_constant = S::a_rad;
	sim.rad.rt.Erad_floor = 0.;
	sim.rad.rt.beta_order = 1;
	setRadGroups(sim, {0.01 * S::T0, 3.3 * S::T0, 1000. * S::T0}, C::k_B, piecewise_constant_opacity);
	sim.rad.eos = sim.hydro.tr.eos;
	sim.rad.ndim = sim.geom.ndim;
	sim.rad.nstartHyperbolic_ = kNumHydroVars;
	const int ng = sim.rad.rt.nGroups;
	// :75-87
	sim.rad.DefineOpacityExponentsAndLowerValues = [ng](double const *, double, double, double *expo, double *lower) {
		for (int i = 0; i < ng + 1; ++i) {
			expo[i] = 0.0;
			lower[i] = S::kappa0;
		}
	};

	// problem_main :254-285
	sim.BCs_cc.assign(sim.ncomp_cc, BCRec{});
	for (int n = 0; n < sim.ncomp_cc; ++n) {
		sim.BCs_cc[n].lo[0] = ext_dir;
		sim.BCs_cc[n].hi[0] = ext_dir;
	}
	sim.radiationReconstructionOrder_ = 3;
	sim.reconstructionOrder_ = 3;
	sim.stopTime_ = S::Lx / S::a0;
	sim.cflNumber_ = 0.4;
	sim.radiationCflNumber_ = 0.4;
	sim.maxTimesteps_ = 2000;

	HydroSim *const simp = &sim;
	// setCustomBoundaryConditions :184-252: constant gas / radiation energy states; the normal momentum and the normal radiation fluxes follow
	// the first valid cell
	sim.customBC = [simp, ng](int i, int j, int k, Array4<double> const &consVar, Box const &dom, double /*time*/) {
		mg::MG const m(simp->rad);
		auto fill = [&](int iv, double rhoB, double TB) {
			auto radEnergyFractions = m.ComputePlanckEnergyFractions(m.boundaries(), TB);
			const double Erad = simp->rad.rt.radiation_constant * std::pow(TB, 4);
			for (int g = 0; g < ng; ++g) {
				const double Frad = consVar(iv, j, k, kNumHydroVars + 1 + kNumRadVars * g);
				consVar(i, j, k, kNumHydroVars + 0 + kNumRadVars * g) = Erad * radEnergyFractions[g];
				consVar(i, j, k, kNumHydroVars + 1 + kNumRadVars * g) = Frad;
				consVar(i, j, k, kNumHydroVars + 2 + kNumRadVars * g) = 0.;
				consVar(i, j, k, kNumHydroVars + 3 + kNumRadVars * g) = 0.;
			}
			const double Egas = (C::k_B / S::mu) * rhoB * TB / (S::gamma_gas - 1.0);
			const double x1Mom = consVar(iv, j, k, x1Momentum_index);
			const double Ekin = 0.5 * (x1Mom * x1Mom) / rhoB;
			consVar(i, j, k, energy_index) = Egas + Ekin;
			consVar(i, j, k, density_index) = rhoB;
			consVar(i, j, k, internalEnergy_index) = Egas;
			consVar(i, j, k, x1Momentum_index) = x1Mom;
			consVar(i, j, k, x2Momentum_index) = 0.;
			consVar(i, j, k, x3Momentum_index) = 0.;
		};
		if (i < dom.lo[0]) {
			fill(dom.lo[0], S::rho0, S::T0);
		} else if (i > dom.hi[0]) {
			fill(dom.hi[0], S::rho1, S::T1);
		}
	};

	sim.define();
	// setInitialConditionsOnGrid :139-182
	std::vector<double> const x_arr(x_tab, x_tab + table_len), rho_arr(rho_tab, rho_tab + table_len), P_arr(Pgas_tab, Pgas_tab + table_len),
	    E_arr(Erad_tab, Erad_tab + table_len);
	Geometry const g = sim.geom;
	mg::MG const m(sim.rad);
	forEachValidCell(sim, [&](Array4<double> const &state_cc, int i, int j, int k) {
		double const x = g.prob_lo[0] + (i + 0.5) * g.dx[0];
		double const rho = interpolate_value(x, x_arr.data(), rho_arr.data(), table_len);
		double const Pgas = interpolate_value(x, x_arr.data(), P_arr.data(), table_len);
		double const Erad = interpolate_value(x, x_arr.data(), E_arr.data(), table_len);
		double const Tgas = Pgas / C::k_B * S::mu / rho;
		auto radEnergyFractions = m.ComputePlanckEnergyFractions(m.boundaries(), Tgas);
		for (int gg = 0; gg < m.nGroups_; ++gg) {
			state_cc(i, j, k, kNumHydroVars + 0 + kNumRadVars * gg) = Erad * radEnergyFractions[gg];
			state_cc(i, j, k, kNumHydroVars + 1 + kNumRadVars * gg) = 0;
			state_cc(i, j, k, kNumHydroVars + 2 + kNumRadVars * gg) = 0;
			state_cc(i, j, k, kNumHydroVars + 3 + kNumRadVars * gg) = 0;
		}
		state_cc(i, j, k, energy_index) = Pgas / (S::gamma_gas - 1.0);
		state_cc(i, j, k, density_index) = rho;
		state_cc(i, j, k, internalEnergy_index) = Pgas / (S::gamma_gas - 1.0);
		state_cc(i, j, k, x1Momentum_index) = 0.;
		state_cc(i, j, k, x2Momentum_index) = 0.;
		state_cc(i, j, k, x3Momentum_index) = 0.;
	});
	sim.finishInitialConditions();
}

// ---------------------------------------------------------------- Marshak wave with nu^-2 opacity (test_radiation_marshak_Vaytet.cpp, deck
// tests/MarshakVaytet.in: cfl = 0.4, 64 cells on [0, 20] cm)
struct MarshakVaytetConstants { // :23-94
	static constexpr double kappa0 = 2000.0;
	static constexpr double nu_pivot = 4.0e13;
	static constexpr double rho0 = 1.0e-3;
	static constexpr double T_initial = 300.0;
	static constexpr double T_L = 1000.0;
	static constexpr double T_R = 300.0;
	static constexpr double rho_C_V = 1.0e-3;
	static constexpr double c_v = rho_C_V / rho0;
	static constexpr double mu = 1.0 / (5. / 3. - 1.) * C::k_B / c_v;
	static constexpr double a_rad = C::a_rad;
	static constexpr double Erad_floor_ = a_rad * T_initial * T_initial * T_initial * T_initial * 1e-20;
};

// the_model = 10 (:27): kappa(nu) = kappa0 (nu / nu_pivot)^-2; n_groups_ = 4 (:13); opacity_model_ = PPL_opacity_full_spectrum (:21), the two
// alternatives listed at :19-20 accepted
inline void setupMarshakVaytet(HydroSim &sim, int opacity_model = PPL_opacity_full_spectrum)
{
	using S = MarshakVaytetConstants;
	sim.hydro.tr.eos.tr.gamma = 5. / 3.; // :96-100
	sim.hydro.tr.eos.tr.mean_molecular_weight = S::mu;
	sim.hydro.tr.eos.tr.boltzmann_constant = C::k_B;
	sim.hydro.tr.reconstruct_eint = true;
	sim.hydro.tr.nscalars = 0;
	sim.is_radiation_enabled = true; // :102-112
	sim.is_hydro_enabled = false;
	sim.rad.rt.c_light = C::c_light; // :114-123
	sim.rad.rt.c_hat = C::c_light;
	sim.rad.rt.radiation_constant = S::a_rad;
	sim.rad.rt.Erad_floor = S::Erad_floor_;
	sim.rad.rt.beta_order = 0;
	setRadGroups(sim, {6.0e10, 6.0e11, 6.0e12, 6.0e13, 6.0e14}, C::hplanck, opacity_model);
	sim.rad.eos = sim.hydro.tr.eos;
	sim.rad.ndim = sim.geom.ndim;
	sim.rad.nstartHyperbolic_ = kNumHydroVars;
	const int ng = sim.rad.rt.nGroups;
	// :125-165 (the_model == 10)
	sim.rad.DefineOpacityExponentsAndLowerValues = [ng, opacity_model](double const *rad_boundaries, double, double, double *expo, double *lower) {
		for (int i = 0; i < ng + 1; ++i) {
			expo[i] = -2.0;
		}
		if (opacity_model == piecewise_constant_opacity) {
			for (int i = 0; i < ng; ++i) {
				auto const bin_center = std::sqrt(rad_boundaries[i] * rad_boundaries[i + 1]);
				lower[i] = S::kappa0 * std::pow(bin_center / S::nu_pivot, -2.);
			}
		} else {
			for (int i = 0; i < ng + 1; ++i) {
				lower[i] = S::kappa0 * std::pow(rad_boundaries[i] / S::nu_pivot, -2.);
			}
		}
	};

	// problem_main :255-285
	sim.BCs_cc.assign(sim.ncomp_cc, BCRec{});
	for (int n = 0; n < sim.ncomp_cc; ++n) {
		sim.BCs_cc[n].lo[0] = ext_dir;
		sim.BCs_cc[n].hi[0] = foextrap;
	}
	sim.radiationReconstructionOrder_ = 3;
	sim.stopTime_ = 1.36e-7;
	sim.maxDt_ = 1.0;
	sim.radiationCflNumber_ = 0.8;
	sim.cflNumber_ = 0.4; // the deck
	sim.maxTimesteps_ = 1000000;

	HydroSim *const simp = &sim;
	EOS const eos = sim.hydro.tr.eos;
	// setCustomBoundaryConditions :167-219 (the functor runs on every cell outside the domain and does not consult the BCRec: the foextrap
	// record of the upper face is overwritten by the 300 K state)
	sim.customBC = [simp, ng, eos](int i, int j, int k, Array4<double> const &consVar, Box const &dom, double /*time*/) {
		if (i < dom.lo[0] || i >= dom.hi[0]) {
			mg::MG const m(simp->rad);
			double const T_H = (i < dom.lo[0]) ? S::T_L : S::T_R;
			auto Erad_g = m.ComputeThermalRadiationMultiGroup(T_H, m.boundaries());
			const double Egas = eos.ComputeEintFromTgas(S::rho0, S::T_initial);
			for (int g = 0; g < ng; ++g) {
				consVar(i, j, k, kNumHydroVars + 0 + kNumRadVars * g) = Erad_g[g];
				consVar(i, j, k, kNumHydroVars + 1 + kNumRadVars * g) = 0.;
				consVar(i, j, k, kNumHydroVars + 2 + kNumRadVars * g) = 0.;
				consVar(i, j, k, kNumHydroVars + 3 + kNumRadVars * g) = 0.;
			}
			consVar(i, j, k, energy_index) = Egas;
			consVar(i, j, k, density_index) = S::rho0;
			consVar(i, j, k, internalEnergy_index) = Egas;
			consVar(i, j, k, x1Momentum_index) = 0.;
			consVar(i, j, k, x2Momentum_index) = 0.;
			consVar(i, j, k, x3Momentum_index) = 0.;
		}
	};

	sim.define();
	mg::MG const m(sim.rad);
	forEachValidCell(sim, [&](Array4<double> const &state_cc, int i, int j, int k) { // :221-251
		const double Egas = eos.ComputeEintFromTgas(S::rho0, S::T_initial);
		auto Erad_g = m.ComputeThermalRadiationMultiGroup(S::T_initial, m.boundaries());
		for (int g = 0; g < m.nGroups_; ++g) {
			state_cc(i, j, k, kNumHydroVars + 0 + kNumRadVars * g) = Erad_g[g];
			state_cc(i, j, k, kNumHydroVars + 1 + kNumRadVars * g) = 0;
			state_cc(i, j, k, kNumHydroVars + 2 + kNumRadVars * g) = 0;
			state_cc(i, j, k, kNumHydroVars + 3 + kNumRadVars * g) = 0;
		}
		state_cc(i, j, k, density_index) = S::rho0;
		state_cc(i, j, k, energy_index) = Egas;
		state_cc(i, j, k, internalEnergy_index) = Egas;
		state_cc(i, j, k, x1Momentum_index) = 0.;
		state_cc(i, j, k, x2Momentum_index) = 0.;
		state_cc(i, j, k, x3Momentum_index) = 0.;
	});
	sim.finishInitialConditions();
}

// ---------------------------------------------------------------- advecting radiation pulse, constant opacity, grey and multigroup
// (test_radhydro_pulse_MG_const_kappa.cpp, deck tests/RadhydroPulse.in: 64 cells on [-512, 512] cm, periodic)
struct PulseMGConstants { // :24-66
	static constexpr double kappa0 = 100.;
	static constexpr double T0 = 1.0e7;
	static constexpr double T1 = 2.0e7;
	static constexpr double rho0 = 1.2;
	static constexpr double a_rad = C::a_rad;
	static constexpr double c = C::c_light;
	static constexpr double chat = c;
	static constexpr double width = 24.0;
	static constexpr double Erad0 = a_rad * T0 * T0 * T0 * T0;
	static constexpr double erad_floor = Erad0 * 1.0e-14;
	static constexpr double mu = 2.33 * C::m_u;
	static constexpr double v0 = 2.0e8;
	static constexpr double max_time = 4.8e-5;
	static constexpr long max_timesteps = 100;
};
// :73-88
inline auto pulseMG_initial_Tgas(const double x) -> double
{
	using S = PulseMGConstants;
	const double sigma = S::width;
	return S::T0 + (S::T1 - S::T0) * std::exp(-x * x / (2.0 * sigma * sigma));
}
inline auto pulseMG_exact_rho(const double x) -> double
{
	using S = PulseMGConstants;
	auto T = pulseMG_initial_Tgas(x);
	return S::rho0 * S::T0 / T + (S::a_rad * S::mu / 3. / C::k_B) * (std::pow(S::T0, 4) / T - std::pow(T, 3));
}

// multigroup = false: problem 1 of the file (SGProblem: grey, gas at rest); true: problem 2 (MGproblem: 4 groups, PPL_opacity_fixed_slope_spectrum,
// advected at v0)
inline void setupPulseMG(HydroSim &sim, bool multigroup)
{
	using S = PulseMGConstants;
	sim.hydro.tr.eos.tr.gamma = 5. / 3.; // :90-94, :150-154
	sim.hydro.tr.eos.tr.mean_molecular_weight = S::mu;
	sim.hydro.tr.eos.tr.boltzmann_constant = C::k_B;
	sim.hydro.tr.reconstruct_eint = true;
	sim.hydro.tr.nscalars = 0;
	sim.is_radiation_enabled = true;
	sim.rad.rt.c_light = S::c; // :107-113, :167-178
	sim.rad.rt.c_hat = S::chat;
	sim.rad.rt.radiation_constant = S::a_rad;
	sim.rad.rt.Erad_floor = S::erad_floor;
	sim.rad.rt.beta_order = 1;
	if (multigroup) {
		setRadGroups(sim, {1e15, 1e16, 1e17, 1e18, 1e19}, C::hplanck, PPL_opacity_fixed_slope_spectrum);
		const int ng = sim.rad.rt.nGroups;
		sim.rad.DefineOpacityExponentsAndLowerValues = [ng](double const *, double, double, double *expo, double *lower) { // :180-194
			for (int g = 0; g < ng + 1; ++g) {
				expo[g] = 0.;
				lower[g] = S::kappa0;
			}
		};
	} else {
		sim.ncomp_cc = kNumHydroVars + kNumRadVars;
		sim.rad.ComputePlanckOpacity = [](double, double) { return S::kappa0; }; // :115-120
		sim.rad.ComputeFluxMeanOpacity = [](double, double) { return S::kappa0; };
		sim.rad.ComputeEnergyMeanOpacity = [](double, double) { return S::kappa0; };
	}
	sim.rad.eos = sim.hydro.tr.eos;
	sim.rad.ndim = sim.geom.ndim;
	sim.rad.nstartHyperbolic_ = kNumHydroVars;

	// problem_main :252-280 / :311-335 (x is periodic in the deck: the foextrap records are never consulted)
	sim.BCs_cc.assign(sim.ncomp_cc, BCRec{});
	for (int n = 0; n < sim.ncomp_cc; ++n) {
		sim.BCs_cc[n].lo[0] = foextrap;
		sim.BCs_cc[n].hi[0] = foextrap;
	}
	sim.radiationReconstructionOrder_ = 3;
	sim.stopTime_ = S::max_time;
	sim.radiationCflNumber_ = 0.8;
	sim.cflNumber_ = 0.8;
	sim.maxDt_ = 1e-3;
	sim.maxTimesteps_ = S::max_timesteps;

	sim.define();
	EOS const eos = sim.hydro.tr.eos;
	Geometry const g = sim.geom;
	double const x0 = g.prob_lo[0] + 0.5 * (g.prob_hi[0] - g.prob_lo[0]);
	mg::MG const m(sim.rad);
	forEachValidCell(sim, [&](Array4<double> const &state_cc, int i, int j, int k) { // :122-148, :196-232
		double const x = g.prob_lo[0] + (i + 0.5) * g.dx[0];
		const double Trad = pulseMG_initial_Tgas(x - x0);
		const double rho = pulseMG_exact_rho(x - x0);
		const double Egas = eos.ComputeEintFromTgas(rho, Trad);
		if (multigroup) {
			auto Erad_g = m.ComputeThermalRadiationMultiGroup(Trad, m.boundaries());
			for (int gg = 0; gg < m.nGroups_; ++gg) {
				state_cc(i, j, k, kNumHydroVars + 0 + kNumRadVars * gg) = Erad_g[gg];
				state_cc(i, j, k, kNumHydroVars + 1 + kNumRadVars * gg) = 4. / 3. * S::v0 * Erad_g[gg];
				state_cc(i, j, k, kNumHydroVars + 2 + kNumRadVars * gg) = 0;
				state_cc(i, j, k, kNumHydroVars + 3 + kNumRadVars * gg) = 0;
			}
			state_cc(i, j, k, energy_index) = Egas + 0.5 * rho * S::v0 * S::v0;
			state_cc(i, j, k, density_index) = rho;
			state_cc(i, j, k, internalEnergy_index) = Egas;
			state_cc(i, j, k, x1Momentum_index) = S::v0 * rho;
		} else {
			const double Erad = S::a_rad * Trad * Trad * Trad * Trad;
			state_cc(i, j, k, kNumHydroVars + 0) = Erad;
			state_cc(i, j, k, kNumHydroVars + 1) = 0;
			state_cc(i, j, k, kNumHydroVars + 2) = 0;
			state_cc(i, j, k, kNumHydroVars + 3) = 0;
			state_cc(i, j, k, energy_index) = Egas;
			state_cc(i, j, k, density_index) = rho;
			state_cc(i, j, k, internalEnergy_index) = Egas;
			state_cc(i, j, k, x1Momentum_index) = 0.;
		}
		state_cc(i, j, k, x2Momentum_index) = 0.;
		state_cc(i, j, k, x3Momentum_index) = 0.;
	});
	sim.finishInitialConditions();
}

// ---------------------------------------------------------------- gas-dust-radiation relaxation with linearised emission
// (src/problems/RadDust/test_rad_dust.cpp, deck tests/RadDust.in: 8 x 4 x 4 cells, periodic, radiation.cfl = 8, dust_gas_interaction_coeff = 1e6)
struct RadDustConstants { // :19-34
	static constexpr double c = 1.0e8;
	static constexpr double chat = c;
	static constexpr double v0 = 0.0;
	static constexpr double chi0 = 10000.0;
	static constexpr double T0 = 1.0;
	static constexpr double rho0 = 1.0;
	static constexpr double a_rad = 1.0;
	static constexpr double mu = 1.0;
	static constexpr double k_B = 1.0;
	static constexpr double max_time = 1.0e-5;
	static constexpr double delta_time = 1.0e-8;
	static constexpr double Erad0 = a_rad * T0 * T0 * T0 * T0;
	static constexpr double erad_floor = 1.0e-20 * Erad0;
};

// multigroup = true: src/problems/RadDustMG/test_rad_dust_MG.cpp (4 groups over 1e-3 .. 1e3 in units of k_B T, PPL_opacity_fixed_slope_spectrum,
// the same exact solution and tolerance: the emission is linear in T_dust and grey, so the group sum obeys the single-group equations)
inline void setupRadDust(HydroSim &sim, bool multigroup = false)
{
	using S = RadDustConstants;
	sim.hydro.tr.eos.tr.gamma = 5. / 3.; // :42-46
	sim.hydro.tr.eos.tr.mean_molecular_weight = S::mu;
	sim.hydro.tr.eos.tr.boltzmann_constant = S::k_B;
	sim.hydro.tr.reconstruct_eint = true;
	sim.hydro.tr.nscalars = 0;
	sim.ncomp_cc = kNumHydroVars + kNumRadVars;
	sim.is_radiation_enabled = true;
	sim.rad.rt.c_light = S::c; // :48-54
	sim.rad.rt.c_hat = S::chat;
	sim.rad.rt.radiation_constant = S::a_rad;
	sim.rad.rt.Erad_floor = S::erad_floor;
	sim.rad.rt.beta_order = 1;
	sim.rad.rt.enable_dust_gas_thermal_coupling_model = true; // :56-60
	sim.rad.rt.dustGasInteractionCoeff = 1.0e6;		    // the deck
	sim.rad.rt.thermal_model = 1;				    // :86-97
	if (multigroup) { // test_rad_dust_MG.cpp:52-81
		setRadGroups(sim, {1.0e-3, 0.1, 1.0, 10.0, 1.0e3}, 1., PPL_opacity_fixed_slope_spectrum);
		const int ng = sim.rad.rt.nGroups;
		sim.rad.DefineOpacityExponentsAndLowerValues = [ng](double const *, double rho, double, double *expo, double *lower) {
			for (int i = 0; i < ng + 1; ++i) {
				expo[i] = 0.0;
				lower[i] = S::chi0 / rho;
			}
		};
	}
	sim.rad.eos = sim.hydro.tr.eos;
	sim.rad.ndim = sim.geom.ndim;
	sim.rad.nstartHyperbolic_ = kNumHydroVars;
	sim.rad.ComputePlanckOpacity = [](double rho, double) { return S::chi0 / rho; }; // :74-84
	sim.rad.ComputeFluxMeanOpacity = [](double rho, double) { return S::chi0 / rho; };
	sim.rad.ComputeEnergyMeanOpacity = [](double rho, double) { return S::chi0 / rho; };

	sim.BCs_cc.assign(sim.ncomp_cc, BCRec{}); // problem_main :145-170 (periodic)
	sim.radiationReconstructionOrder_ = 3;
	sim.stopTime_ = S::max_time;
	sim.cflNumber_ = 0.8;
	sim.radiationCflNumber_ = 8.0; // the deck
	sim.maxTimesteps_ = 1000000;
	sim.initDt_ = S::delta_time;
	sim.maxDt_ = S::delta_time;

	sim.define();
	EOS const eos = sim.hydro.tr.eos;
	const double Egas = eos.ComputeEintFromTgas(S::rho0, S::T0);
	const int ngroups = sim.rad.rt.nGroups;
	forEachValidCell(sim, [&](Array4<double> const &state_cc, int i, int j, int k) { // :99-122 (MG :106-129: the floor in EVERY group)
		for (int g = 0; g < ngroups; ++g) {
			state_cc(i, j, k, kNumHydroVars + 0 + kNumRadVars * g) = S::erad_floor;
			state_cc(i, j, k, kNumHydroVars + 1 + kNumRadVars * g) = 0;
			state_cc(i, j, k, kNumHydroVars + 2 + kNumRadVars * g) = 0;
			state_cc(i, j, k, kNumHydroVars + 3 + kNumRadVars * g) = 0;
		}
		state_cc(i, j, k, energy_index) = Egas + 0.5 * S::rho0 * S::v0 * S::v0;
		state_cc(i, j, k, density_index) = S::rho0;
		state_cc(i, j, k, internalEnergy_index) = Egas;
		state_cc(i, j, k, x1Momentum_index) = S::v0 * S::rho0;
		state_cc(i, j, k, x2Momentum_index) = 0.;
		state_cc(i, j, k, x3Momentum_index) = 0.;
	});
	sim.finishInitialConditions();
}

// ---------------------------------------------------------------- two-group Marshak wave with dust (src/problems/RadMarshakDust/test_radiation_marshak_dust.cpp,
// deck tests/RadMarshakDust.in: 256 cells, dust_gas_interaction_coeff = 1e-2, kappa1 = 1e10 (IR), kappa2 = 1 (FUV), stop_time = 0.5)
struct MarshakDustConstants { // :19-37
	static constexpr double c = 1.0, chat = 1.0, rho0 = 1.0, CV = 1.0;
	static constexpr double mu = 1.5 / CV;
	static constexpr double initial_T = 1.0;
	static constexpr double a_rad = 1.0e10;
	static constexpr double erad_floor = 1.0e-10;
	static constexpr double initial_Trad = 1.0e-5;
	static constexpr double T_rad_L = 1.0e-2;
	static constexpr double EradL = a_rad * T_rad_L * T_rad_L * T_rad_L * T_rad_L;
	static constexpr double kappa1 = 1.0e10, kappa2 = 1.0; // the deck
};

inline void setupMarshakDust(HydroSim &sim)
{
	using S = MarshakDustConstants;
	sim.hydro.tr.eos.tr.gamma = 5. / 3.; // :41-45
	sim.hydro.tr.eos.tr.mean_molecular_weight = S::mu;
	sim.hydro.tr.eos.tr.boltzmann_constant = 1.0;
	sim.hydro.tr.reconstruct_eint = true;
	sim.hydro.tr.nscalars = 0;
	sim.is_radiation_enabled = true; // :47-57
	sim.is_hydro_enabled = false;
	sim.rad.rt.c_light = S::c; // :59-68
	sim.rad.rt.c_hat = S::chat;
	sim.rad.rt.radiation_constant = S::a_rad;
	sim.rad.rt.Erad_floor = S::erad_floor;
	sim.rad.rt.beta_order = 0;
	sim.rad.rt.enable_dust_gas_thermal_coupling_model = true; // :70-74
	sim.rad.rt.gas_dust_coupling_threshold = 1.0e-5;
	sim.rad.rt.dustGasInteractionCoeff = 1e-2; // the deck
	setRadGroups(sim, {1e-10, 100, 1e4}, 1.0, piecewise_constant_opacity);
	sim.rad.eos = sim.hydro.tr.eos;
	sim.rad.ndim = sim.geom.ndim;
	sim.rad.nstartHyperbolic_ = kNumHydroVars;
	const int ng = sim.rad.rt.nGroups;
	sim.rad.DefineOpacityExponentsAndLowerValues = [ng](double const *, double, double, double *expo, double *lower) { // :86-102
		for (int i = 0; i < ng + 1; ++i) {
			expo[i] = 0.0;
			lower[i] = (i == 0) ? S::kappa1 : S::kappa2;
		}
	};
	// problem_main :186-212
	sim.BCs_cc.assign(sim.ncomp_cc, BCRec{});
	for (int n = 0; n < sim.ncomp_cc; ++n) {
		sim.BCs_cc[n].lo[0] = ext_dir;
		sim.BCs_cc[n].hi[0] = foextrap;
	}
	sim.radiationReconstructionOrder_ = 3;
	sim.radiationCflNumber_ = 0.8;
	sim.maxDt_ = 1;
	sim.maxTimesteps_ = 5000;
	sim.stopTime_ = 0.5; // the deck
	// setCustomBoundaryConditions :126-176: the radiation state beyond the lower face streams in at c (F = c E in the FUV group); the gas state is
	// written on EVERY cell outside the domain (the functor runs beyond the extrapolating upper face as well)
	sim.customBC = [ng](int i, int j, int k, Array4<double> const &consVar, Box const &dom, double /*time*/) {
		const double Erads[2] = {S::erad_floor, S::EradL};
		if (i < dom.lo[0]) {
			for (int g = 0; g < ng; ++g) {
				consVar(i, j, k, kNumHydroVars + 0 + kNumRadVars * g) = Erads[g];
				consVar(i, j, k, kNumHydroVars + 1 + kNumRadVars * g) = Erads[g] * S::c;
				consVar(i, j, k, kNumHydroVars + 2 + kNumRadVars * g) = 0;
				consVar(i, j, k, kNumHydroVars + 3 + kNumRadVars * g) = 0;
			}
		}
		const double Egas = S::initial_T * S::CV;
		consVar(i, j, k, energy_index) = Egas;
		consVar(i, j, k, density_index) = S::rho0;
		consVar(i, j, k, internalEnergy_index) = Egas;
		consVar(i, j, k, x1Momentum_index) = 0.;
		consVar(i, j, k, x2Momentum_index) = 0.;
		consVar(i, j, k, x3Momentum_index) = 0.;
	};
	sim.define();
	mg::MG const m(sim.rad);
	const auto Erads0 = m.ComputeThermalRadiationMultiGroup(S::initial_Trad, m.boundaries());
	forEachValidCell(sim, [&](Array4<double> const &state_cc, int i, int j, int k) { // :104-124
		const double Egas0 = S::initial_T * S::CV;
		for (int g = 0; g < m.nGroups_; ++g) {
			state_cc(i, j, k, kNumHydroVars + 0 + kNumRadVars * g) = Erads0[g];
			state_cc(i, j, k, kNumHydroVars + 1 + kNumRadVars * g) = 0;
			state_cc(i, j, k, kNumHydroVars + 2 + kNumRadVars * g) = 0;
			state_cc(i, j, k, kNumHydroVars + 3 + kNumRadVars * g) = 0;
		}
		state_cc(i, j, k, energy_index) = Egas0;
		state_cc(i, j, k, density_index) = S::rho0;
		state_cc(i, j, k, internalEnergy_index) = Egas0;
		state_cc(i, j, k, x1Momentum_index) = 0.;
		state_cc(i, j, k, x2Momentum_index) = 0.;
		state_cc(i, j, k, x3Momentum_index) = 0.;
	});
	sim.finishInitialConditions();
}

// ---------------------------------------------------------------- line cooling, cosmic-ray and photoelectric heating of a uniform medium
// (src/problems/RadLineCooling/test_rad_line_cooling.cpp: one group; src/problems/RadLineCoolingMG/test_rad_line_cooling_MG.cpp: four groups with
// photoelectric heating by the last one; decks tests/RadLineCooling.in (dust_gas_interaction_coeff = 1e-20) and RadLineCoolingCoupled.in (1e20))
struct LineCoolingConstants { // test_rad_line_cooling.cpp:19-36, ..._MG.cpp:19-42
	static constexpr double cooling_rate = 0.1, CR_heating_rate = 0.03, PE_rate = 0.02;
	static constexpr double c = 1.0, chat = c, v0 = 0.0, kappa0 = 0.0;
	static constexpr double T0 = 1.0, rho0 = 1.0, a_rad = 1.0, mu = 1.5, C_V = 1.0, k_B = 1.0, nu_unit = 1.0;
	static constexpr double erad_floor = a_rad * 1e-20;
	static constexpr double Erad_FUV = a_rad * T0 * T0 * T0 * T0;
	static constexpr double max_time = 10.0, the_dt = 1.0e-2;
	static constexpr int line_index = 0;
};

inline void setupLineCooling(HydroSim &sim, bool multigroup, double dust_coeff)
{
	using S = LineCoolingConstants;
	sim.hydro.tr.eos.tr.gamma = 5. / 3.;
	sim.hydro.tr.eos.tr.mean_molecular_weight = S::mu;
	sim.hydro.tr.eos.tr.boltzmann_constant = S::k_B;
	sim.hydro.tr.reconstruct_eint = true;
	sim.hydro.tr.nscalars = 0;
	sim.is_radiation_enabled = true;
	sim.rad.rt.c_light = S::c;
	sim.rad.rt.c_hat = S::chat;
	sim.rad.rt.radiation_constant = S::a_rad;
	sim.rad.rt.Erad_floor = S::erad_floor;
	sim.rad.rt.beta_order = 0;
	sim.rad.rt.enable_dust_gas_thermal_coupling_model = true; // ISM_Traits
	sim.rad.rt.gas_dust_coupling_threshold = 1.0e-6;
	sim.rad.rt.dustGasInteractionCoeff = dust_coeff; // the deck
	int ng = 1;
	if (multigroup) { // ..._MG.cpp:22-23, :64-83
		setRadGroups(sim, {1.00000000e-03, 1.77827941e-02, 3.16227766e-01, 5.62341325e+00, 1.00000000e+02}, S::nu_unit, piecewise_constant_opacity);
		sim.rad.rt.enable_photoelectric_heating = true;
		ng = sim.rad.rt.nGroups;
		sim.rad.DefinePhotoelectricHeatingE1Derivative = [](double, double) { return S::PE_rate / S::Erad_FUV; }; // :86-91
	} else {
		sim.ncomp_cc = kNumHydroVars + kNumRadVars;
	}
	sim.rad.DefineOpacityExponentsAndLowerValues = [ng](double const *, double, double, double *expo, double *lower) {
		for (int i = 0; i < ng + 1; ++i) {
			expo[i] = 0.0;
			lower[i] = S::kappa0;
		}
	};
	sim.rad.DefineNetCoolingRate = [ng](double temperature, double, double *cooling) { // :77-86 / MG :93-101
		for (int g = 0; g < ng; ++g) {
			cooling[g] = 0.0;
		}
		cooling[S::line_index] = S::cooling_rate * temperature;
	};
	sim.rad.DefineNetCoolingRateTempDerivative = [ng](double, double, double *cooling) {
		for (int g = 0; g < ng; ++g) {
			cooling[g] = 0.0;
		}
		cooling[S::line_index] = S::cooling_rate;
	};
	sim.rad.DefineCosmicRayHeatingRate = [](double) { return S::CR_heating_rate; };
	sim.rad.eos = sim.hydro.tr.eos;
	sim.rad.ndim = sim.geom.ndim;
	sim.rad.nstartHyperbolic_ = kNumHydroVars;
	sim.rad.ComputePlanckOpacity = [](double, double) { return S::kappa0; };
	sim.rad.ComputeFluxMeanOpacity = [](double, double) { return S::kappa0; };
	sim.rad.ComputeEnergyMeanOpacity = [](double, double) { return S::kappa0; };

	sim.BCs_cc.assign(sim.ncomp_cc, BCRec{}); // problem_main: periodic
	sim.radiationReconstructionOrder_ = 3;
	sim.stopTime_ = S::max_time;
	sim.radiationCflNumber_ = 0.8;
	sim.cflNumber_ = 0.8;
	sim.initDt_ = S::the_dt;
	sim.maxDt_ = S::the_dt;
	sim.maxTimesteps_ = 1000000;

	sim.define();
	EOS const eos = sim.hydro.tr.eos;
	const double Egas = eos.ComputeEintFromTgas(S::rho0, S::T0);
	forEachValidCell(sim, [&](Array4<double> const &state_cc, int i, int j, int k) {
		for (int g = 0; g < ng; ++g) {
			// one group: the floor; four groups: E_FUV in the last group, the floor elsewhere (MG :142-148)
			state_cc(i, j, k, kNumHydroVars + 0 + kNumRadVars * g) = (multigroup && g == ng - 1) ? S::Erad_FUV : S::erad_floor;
			state_cc(i, j, k, kNumHydroVars + 1 + kNumRadVars * g) = 0;
			state_cc(i, j, k, kNumHydroVars + 2 + kNumRadVars * g) = 0;
			state_cc(i, j, k, kNumHydroVars + 3 + kNumRadVars * g) = 0;
		}
		state_cc(i, j, k, energy_index) = Egas + 0.5 * S::rho0 * S::v0 * S::v0;
		state_cc(i, j, k, density_index) = S::rho0;
		state_cc(i, j, k, internalEnergy_index) = Egas;
		state_cc(i, j, k, x1Momentum_index) = S::v0 * S::rho0;
		state_cc(i, j, k, x2Momentum_index) = 0.;
		state_cc(i, j, k, x3Momentum_index) = 0.;
	});
	sim.finishInitialConditions();
}

// ---------------------------------------------------------------- FUV front heating the gas photoelectrically (src/problems/RadMarshakDustPE/
// test_radiation_marshak_dust_and_PE.cpp; decks tests/RadMarshakDustPEcoupled.in / ...decoupled.in: kappa1 = kappa2 = 1e-20,
// dust_gas_interaction_coeff = 1e20 / 1e-20, stop_time = 0.5)
struct MarshakDustPEConstants { // :19-36
	static constexpr double PE_rate = 1.0;
	static constexpr double c = 1.0, chat = 1.0, rho0 = 1.0, CV = 1.0;
	static constexpr double mu = 1.5 / CV;
	static constexpr double initial_T = 1.0, a_rad = 1.0, erad_floor = 1.0e-6, T_rad_L = 1.0;
	static constexpr double EradL = a_rad * T_rad_L * T_rad_L * T_rad_L * T_rad_L;
	static constexpr double kappa1 = 1e-20, kappa2 = 1e-20; // the decks
};

inline void setupMarshakDustPE(HydroSim &sim, double dust_coeff)
{
	using S = MarshakDustPEConstants;
	sim.hydro.tr.eos.tr.gamma = 5. / 3.;
	sim.hydro.tr.eos.tr.mean_molecular_weight = S::mu;
	sim.hydro.tr.eos.tr.boltzmann_constant = 1.0;
	sim.hydro.tr.reconstruct_eint = true;
	sim.hydro.tr.nscalars = 0;
	sim.is_radiation_enabled = true;
	sim.is_hydro_enabled = false;
	sim.rad.rt.c_light = S::c;
	sim.rad.rt.c_hat = S::chat;
	sim.rad.rt.radiation_constant = S::a_rad;
	sim.rad.rt.Erad_floor = S::erad_floor;
	sim.rad.rt.beta_order = 1;
	sim.rad.rt.enable_dust_gas_thermal_coupling_model = true; // :70-74
	sim.rad.rt.gas_dust_coupling_threshold = 1.0e-4;
	sim.rad.rt.enable_photoelectric_heating = true;
	sim.rad.rt.dustGasInteractionCoeff = dust_coeff;
	setRadGroups(sim, {1e-10, 30, 1e4}, 1.0, piecewise_constant_opacity);
	sim.rad.eos = sim.hydro.tr.eos;
	sim.rad.ndim = sim.geom.ndim;
	sim.rad.nstartHyperbolic_ = kNumHydroVars;
	const int ng = sim.rad.rt.nGroups;
	sim.rad.DefinePhotoelectricHeatingE1Derivative = [](double, double) { return S::PE_rate; }; // :76-88
	sim.rad.DefineOpacityExponentsAndLowerValues = [ng](double const *, double, double, double *expo, double *lower) { // :90-106
		for (int i = 0; i < ng + 1; ++i) {
			expo[i] = 0.0;
			lower[i] = (i == 0) ? S::kappa1 : S::kappa2;
		}
	};
	sim.BCs_cc.assign(sim.ncomp_cc, BCRec{}); // problem_main :180-215
	for (int n = 0; n < sim.ncomp_cc; ++n) {
		sim.BCs_cc[n].lo[0] = ext_dir;
		sim.BCs_cc[n].hi[0] = foextrap;
	}
	sim.radiationReconstructionOrder_ = 3;
	sim.radiationCflNumber_ = 0.8;
	sim.maxDt_ = 1;
	sim.maxTimesteps_ = 5000;
	sim.stopTime_ = 0.5; // the decks
	sim.customBC = [ng](int i, int j, int k, Array4<double> const &consVar, Box const &dom, double /*time*/) { // :130-178
		const double Erads[2] = {S::erad_floor, S::EradL};
		if (i < dom.lo[0]) {
			for (int g = 0; g < ng; ++g) {
				consVar(i, j, k, kNumHydroVars + 0 + kNumRadVars * g) = Erads[g];
				consVar(i, j, k, kNumHydroVars + 1 + kNumRadVars * g) = Erads[g] * S::c;
				consVar(i, j, k, kNumHydroVars + 2 + kNumRadVars * g) = 0;
				consVar(i, j, k, kNumHydroVars + 3 + kNumRadVars * g) = 0;
			}
		}
		const double Egas = S::initial_T * S::CV;
		consVar(i, j, k, energy_index) = Egas;
		consVar(i, j, k, density_index) = S::rho0;
		consVar(i, j, k, internalEnergy_index) = Egas;
		consVar(i, j, k, x1Momentum_index) = 0.;
		consVar(i, j, k, x2Momentum_index) = 0.;
		consVar(i, j, k, x3Momentum_index) = 0.;
	};
	sim.define();
	forEachValidCell(sim, [&](Array4<double> const &state_cc, int i, int j, int k) { // :108-128
		const double Egas0 = S::initial_T * S::CV;
		for (int g = 0; g < ng; ++g) {
			state_cc(i, j, k, kNumHydroVars + 0 + kNumRadVars * g) = S::erad_floor;
			state_cc(i, j, k, kNumHydroVars + 1 + kNumRadVars * g) = 0;
			state_cc(i, j, k, kNumHydroVars + 2 + kNumRadVars * g) = 0;
			state_cc(i, j, k, kNumHydroVars + 3 + kNumRadVars * g) = 0;
		}
		state_cc(i, j, k, energy_index) = Egas0;
		state_cc(i, j, k, density_index) = S::rho0;
		state_cc(i, j, k, internalEnergy_index) = Egas0;
		state_cc(i, j, k, x1Momentum_index) = 0.;
		state_cc(i, j, k, x2Momentum_index) = 0.;
		state_cc(i, j, k, x3Momentum_index) = 0.;
	});
	sim.finishInitialConditions();
}

// ---------------------------------------------------------------- Quirk's odd-even decoupling test (src/problems/HydroQuirk/test_quirk.cpp,
// deck tests/quirk.in: 128 x 16 (x 16) cells on 1 x 0.125; built for AMREX_SPACEDIM >= 2) — the 2-D pin of the hydro path
struct QuirkConstants { // :58-63
	static constexpr double dl = 3.692, ul = -0.625, pl = 26.85;
	static constexpr double dr = 1.0, ur = -5.0, pr = 0.6;
};

inline void setupQuirk(HydroSim &sim)
{
	using S = QuirkConstants;
	sim.hydro.tr.eos.tr.gamma = 5. / 3.; // :38-46
	sim.hydro.tr.eos.tr.mean_molecular_weight = C::m_u;
	sim.hydro.tr.eos.tr.boltzmann_constant = C::k_B;
	sim.hydro.tr.reconstruct_eint = false;
	sim.hydro.tr.nscalars = 0;
	sim.ncomp_cc = kNumHydroVars;
	// problem_main :248-272 (only component 0 carries ext_dir there; the functor fills every component of every ghost cell beyond x)
	sim.BCs_cc.assign(sim.ncomp_cc, BCRec{});
	for (int n = 0; n < sim.ncomp_cc; ++n) {
		sim.BCs_cc[n].lo[0] = ext_dir;
		sim.BCs_cc[n].hi[0] = ext_dir;
	}
	sim.reconstructionOrder_ = 2; // PLM
	sim.stopTime_ = 0.4;
	sim.cflNumber_ = 0.4;
	sim.maxTimesteps_ = 2000;
	const double gamma = sim.hydro.tr.eos.tr.gamma;
	// setCustomBoundaryConditions :203-246
	sim.customBC = [gamma](int i, int j, int k, Array4<double> const &consVar, Box const &dom, double /*time*/) {
		if (i < dom.lo[0]) {
			consVar(i, j, k, energy_index) = S::pl / (gamma - 1.) + 0.5 * S::dl * S::ul * S::ul;
			consVar(i, j, k, internalEnergy_index) = S::pl / (gamma - 1.);
			consVar(i, j, k, density_index) = S::dl;
			consVar(i, j, k, x1Momentum_index) = S::dl * S::ul;
			consVar(i, j, k, x2Momentum_index) = 0.;
			consVar(i, j, k, x3Momentum_index) = 0.;
		} else if (i >= dom.hi[0]) {
			consVar(i, j, k, energy_index) = S::pr / (gamma - 1.) + 0.5 * S::dr * S::ur * S::ur;
			consVar(i, j, k, internalEnergy_index) = S::pr / (gamma - 1.);
			consVar(i, j, k, density_index) = S::dr;
			consVar(i, j, k, x1Momentum_index) = S::dr * S::ur;
			consVar(i, j, k, x2Momentum_index) = 0.;
			consVar(i, j, k, x3Momentum_index) = 0.;
		}
	};
	sim.define();
	Geometry const g = sim.geom;
	EOS const eos = sim.hydro.tr.eos;
	// setInitialConditionsOnGrid :66-127
	double const xshock = 0.4;
	int ishock = 0;
	for (ishock = 0; (g.prob_lo[0] + g.dx[0] * (ishock + 0.5)) < xshock; ++ishock) {
	}
	ishock--;
	double const dd = S::dl - 0.135;
	double const ud = S::ul + 0.219;
	double const pd = S::pl - 1.31;
	forEachValidCell(sim, [&](Array4<double> const &state_cc, int i, int j, int k) {
		double vx = NAN, vy = 0., vz = 0., rho = NAN, P = NAN;
		if (i <= ishock) {
			rho = S::dl;
			vx = S::ul;
			P = S::pl;
		} else {
			rho = S::dr;
			vx = S::ur;
			P = S::pr;
		}
		if ((i == ishock) && (j % 2 == 0)) {
			rho = dd;
			vx = ud;
			P = pd;
		}
		const auto v_sq = vx * vx + vy * vy + vz * vz;
		state_cc(i, j, k, density_index) = rho;
		state_cc(i, j, k, x1Momentum_index) = rho * vx;
		state_cc(i, j, k, x2Momentum_index) = rho * vy;
		state_cc(i, j, k, x3Momentum_index) = rho * vz;
		state_cc(i, j, k, energy_index) = eos.ComputeEintFromPres(rho, P) + 0.5 * rho * v_sq;
		state_cc(i, j, k, internalEnergy_index) = eos.ComputeEintFromPres(rho, P);
	});
	sim.finishInitialConditions();
}

// ---------------------------------------------------------------- circular blast in a reflecting box (src/problems/HydroBlast2D/test_hydro2d_blast.cpp,
// deck tests/blast2d.in).  Runs as a 2-D build (x-y) or as a 3-D build whose z direction is uniform: with v_z = 0 the two builds perform the same
// arithmetic on every x-y plane (the X2 view is an index swap in one and a cyclic permutation in the other; the z sweep adds exact zeros), which
// ties the 2-D restatement to the 3-D one that the reference's known-answer tests pin.
inline void setupBlast2D(HydroSim &sim)
{
	sim.hydro.tr.eos.tr.gamma = 5. / 3.; // :28-32
	sim.hydro.tr.eos.tr.mean_molecular_weight = C::m_u;
	sim.hydro.tr.eos.tr.boltzmann_constant = C::k_B;
	sim.hydro.tr.reconstruct_eint = true;
	sim.hydro.tr.nscalars = 0;
	sim.ncomp_cc = kNumHydroVars;
	// problem_main :127-164: reflecting walls
	sim.BCs_cc.assign(sim.ncomp_cc, BCRec{});
	for (int n = 0; n < sim.ncomp_cc; ++n) {
		for (int d = 0; d < sim.geom.ndim; ++d) {
			bool const normal = (n == x1Momentum_index + d);
			sim.BCs_cc[n].lo[d] = normal ? reflect_odd : reflect_even;
			sim.BCs_cc[n].hi[d] = normal ? reflect_odd : reflect_even;
		}
	}
	sim.stopTime_ = 0.1;
	sim.cflNumber_ = 0.3;
	sim.maxTimesteps_ = 20000;
	sim.define();
	Geometry const g = sim.geom;
	double const x0 = g.prob_lo[0] + 0.5 * (g.prob_hi[0] - g.prob_lo[0]);
	double const y0 = g.prob_lo[1] + 0.5 * (g.prob_hi[1] - g.prob_lo[1]);
	double const gamma = sim.hydro.tr.eos.tr.gamma;
	forEachValidCell(sim, [&](Array4<double> const &state_cc, int i, int j, int k) { // :44-91 (the internal energy is left to the first sync)
		double const x = g.prob_lo[0] + (i + 0.5) * g.dx[0];
		double const y = g.prob_lo[1] + (j + 0.5) * g.dx[1];
		double const R = std::sqrt(std::pow(x - x0, 2) + std::pow(y - y0, 2));
		double const rho = 1.0;
		double const P = (R < 0.1) ? 10. : 0.1;
		state_cc(i, j, k, density_index) = rho;
		state_cc(i, j, k, x1Momentum_index) = 0.;
		state_cc(i, j, k, x2Momentum_index) = 0.;
		state_cc(i, j, k, x3Momentum_index) = 0.;
		state_cc(i, j, k, energy_index) = P / (gamma - 1.) + 0.5 * rho * 0.;
		state_cc(i, j, k, internalEnergy_index) = 0.;
	});
	sim.finishInitialConditions();
}

} // namespace oracle

#endif // ORACLE_PROBLEMS_MULTIGROUP_HPP_
