// qk_rad_device.hpp — per-cell / per-face device arithmetic of the two-moment radiation path.
// Counterparts (same association order, -ffp-contract=off):
//   reference src/radiation/radiation_system.hpp            RadSystem<problem_t>
//   reference src/radiation/source_terms_single_group.hpp   AddSourceTermsSingleGroup
// The face flux and the state repair act on one photon group at a time (the transport of the groups is independent); the single-group source
// term is here, the multigroup one in qk_rad_mg_device.hpp.
#ifndef QK_RAD_DEVICE_HPP_
#define QK_RAD_DEVICE_HPP_

#include "qk_device.hpp"

namespace qk
{

constexpr int NRAD = 4;	  // Physics_NumVars::numRadVars
constexpr int RAD0 = 6;	  // Physics_Indices::radFirstIndex (no passive scalars)
constexpr double IMEX_a32 = 0.5; // radiation_system.hpp:52

struct Rad {
	double c, chat, arad, Erad_floor;
	double kappaP0, kappaE0, kappaF0;
	double kT_ref, kT_exp, kT_floor;
	int beta_order, pow_mode, opacity_model, eddington_model;
	int ngroups; // Physics_Traits::nGroups
	double dust_coeff; // QuokkaSimulation::dustGasInteractionCoeff_ (DUST instantiation of the source kernel only)
	double dust_threshold; // ISM_Traits::gas_dust_coupling_threshold (multigroup dust model)
	double cool0, cr_heat, pe_rate; // ISM hooks, closed set: net cooling rate of group 0 = cool0 * T; cosmic-ray heating rate; photoelectric E1 derivative
	int pe_on; // ISM_Traits::enable_photoelectric_heating
	double mean_molecular_mass = 0.; // EOS_Traits::mean_molecular_weight as given (ComputeNumberDensityH; set by the source-term launcher)
	int thermal_model; // 0: a T^4; 1: a T (RadDust's hooks; DUST instantiation only)
	// quotients of run-time constants, formed once by the constructor's IEEE divisions instead of once per cell: 1 / (c c_hat) (the momentum exchange),
	// c / c_hat and its reciprocal, 1 / c^2 (the work term)
	double r_cchat, cscale, inv_cscale, r_cc;
	__host__ __device__ explicit Rad(qk_rad_traits const &t)
	    : c(t.c_light), chat(t.c_hat), arad(t.radiation_constant),
	      Erad_floor(t.Erad_floor / ((t.ngroups > 1) ? t.ngroups : 1)), // Erad_floor_ = RadSystem_Traits::Erad_floor / nGroups_ (radiation_system.hpp:211)
	      kappaP0(t.kappaP), kappaE0(t.kappaE), kappaF0(t.kappaF), kT_ref(t.opacity_T_ref), kT_exp(t.opacity_T_exponent), kT_floor(t.opacity_pow_floor),
	      beta_order(t.beta_order), pow_mode(t.pow_mode), opacity_model(t.opacity_model), eddington_model(t.eddington_model),
	      ngroups((t.ngroups > 1) ? t.ngroups : 1), dust_coeff(t.dust_gas_interaction_coeff), dust_threshold(t.gas_dust_coupling_threshold),
	      cool0(t.cooling_linear_coeff[0]), cr_heat(t.cr_heating_rate), pe_rate(t.pe_heating_E1_derivative), pe_on(t.enable_photoelectric_heating),
	      thermal_model(t.thermal_model), r_cchat(1.0 / (t.c_light * t.c_hat)), cscale(t.c_light / t.c_hat), inv_cscale(1 / (t.c_light / t.c_hat)),
	      r_cc(1.0 / (t.c_light * t.c_light))
	{
	}
	// problem hooks ComputePlanckOpacity / ComputeEnergyMeanOpacity / ComputeFluxMeanOpacity (radiation_system.hpp:1141-1154):
	// closed set.  Model 0: constants [cm^2 g^-1]; model 1: constant absorption coefficient rho * kappa, i.e. kappa = k0 / rho
	// (src/problems/RadhydroShockCGS/test_radhydro_shock_cgs.cpp:78-86)
	// Model 2 (TDEP instantiation of the source kernel only, so that the constant-opacity kernel keeps its registers): temperature
	// power law  k0 * max((T / T_ref)^p, floor) / rho  (src/problems/RadMarshakAsymptotic/test_radiation_marshak_asymptotic.cpp:55-59)
	QK_DEV auto opacityPow(double T) const -> double
	{
		const double x = T / kT_ref;
		double pw;
		if (pow_mode == 1 && kT_exp == 3.0) {
			pw = (x * x) * x;
		} else if (pow_mode == 1 && kT_exp == -3.0) {
			pw = 1.0 / ((x * x) * x);
		} else if (pow_mode == 1 && kT_exp == -3.5) {
			pw = 1.0 / (((x * x) * x) * sqrt(x));
		} else {
			pw = pow(x, kT_exp);
		}
		return (pw > kT_floor) ? pw : kT_floor;
	}
	template <bool TDEP> QK_DEV auto kappaX(double k0, double rho, double T) const -> double
	{
		if constexpr (TDEP) {
			return (k0 * opacityPow(T)) / rho;
		} else {
			return (opacity_model == 0) ? k0 : k0 / rho;
		}
	}
	template <bool TDEP = false> QK_DEV auto kappaP(double rho, double T) const -> double { return kappaX<TDEP>(kappaP0, rho, T); }
	template <bool TDEP = false> QK_DEV auto kappaE(double rho, double T) const -> double { return kappaX<TDEP>(kappaE0, rho, T); }
	template <bool TDEP = false> QK_DEV auto kappaF(double rho, double T) const -> double { return kappaX<TDEP>(kappaF0, rho, T); }
	// the ComputeEddingtonFactor hook: 0 = Levermore closure (radiation_system.hpp:773-790, the default), 1 = Eddington approximation
	QK_DEV auto eddingtonFactor(double f_in) const -> double
	{
		if (eddington_model == 1) {
			return (1. / 3.);
		}
		const double f = clampd(f_in, 0., 1.);
		const double f_fac = sqrtN(4.0 - 3.0 * (f * f));
		return divN(3.0 + 4.0 * (f * f), 5.0 + 2.0 * f_fac);
	}
	// pow_mode 0 stands for the reference's std::pow(T, 4) / std::pow(T, 3), which glibc rounds correctly in all but a vanishing fraction of
	// cases.  Evaluated here as compensated products (T^2 = hi + lo exactly by one fma; the product of the pair carried with its rounding
	// error): faithfully rounded, i.e. within 0.5 ulp + 2^-100 of the exact power — the same bits as a correctly rounded pow except in
	// near-tie cases — at a dozen instructions instead of the ~200 of the device libm's general pow (which is only accurate to ~1 ulp).
	// pow_mode 1: plain repeated multiplication (what the bit-level tests share with the oracle).
	QK_DEV static auto pow4Faithful(double T) -> double
	{
		const double hi = T * T;
		const double lo = __builtin_fma(T, T, -hi);
		const double r = hi * hi;
		const double e = __builtin_fma(hi, hi, -r);
		return r + (e + 2.0 * (hi * lo));
	}
	QK_DEV static auto pow3Faithful(double T) -> double
	{
		const double hi = T * T;
		const double lo = __builtin_fma(T, T, -hi);
		const double r = hi * T;
		const double e = __builtin_fma(hi, T, -r);
		return r + (e + lo * T);
	}
	QK_DEV auto pow4(double T) const -> double { return (pow_mode == 0) ? pow4Faithful(T) : (T * T) * (T * T); }
	QK_DEV auto pow3(double T) const -> double { return (pow_mode == 0) ? pow3Faithful(T) : (T * T) * T; }
	// radiation_system.hpp:471-479, :499-503
	QK_DEV auto thermalRadiation(double T) const -> double
	{
		double power = arad * pow4(T);
		if (power < Erad_floor) {
			power = Erad_floor;
		}
		return power;
	}
	QK_DEV auto thermalRadiationTempDerivative(double T) const -> double { return 4. * arad * pow3(T); }
	// the same two hooks as RadDust specialises them (test_rad_dust.cpp:86-97) when thermal_model == 1
	// the ISM hooks DefineCosmicRayHeatingRate / DefineNetCoolingRate (radiation_system.hpp:344-353), closed set: constant rate, cooling linear in T
	QK_DEV auto cosmicRayHeatingRate(double /*num_density*/) const -> double { return cr_heat; }
	QK_DEV auto netCoolingRate(double T, double /*num_density*/) const -> double { return cool0 * T; }
	QK_DEV auto thermalRadiationHook(double T) const -> double { return (thermal_model == 1) ? arad * T : thermalRadiation(T); }
	QK_DEV auto thermalRadiationTempDerivativeHook(double T) const -> double { return (thermal_model == 1) ? arad : thermalRadiationTempDerivative(T); }
};

// radiation_system.hpp:873-916: row `row` of the Eddington tensor, plus T[row][row] via Tn[row]
QK_DEV void eddingtonTensor(Rad const &r, double fx, double fy, double fz, double T[3][3])
{
	const double f = sqrtN(fx * fx + fy * fy + fz * fz);
	const double fv[3] = {fx, fy, fz};
	const Recip Rf = recipOf(f); // (f == 0: the NaN quotients are discarded by the select below)
	double n[3];
#pragma unroll
	for (int ii = 0; ii < 3; ++ii) {
		n[ii] = (f > 0.) ? divBy(fv[ii], Rf) : 0.;
	}
	const double chi = r.eddingtonFactor(f);
	const double Tdiag = (1.0 - chi) / 2.0;
	const double Tf = (3.0 * chi - 1.0) / 2.0;
#pragma unroll
	for (int ii = 0; ii < 3; ++ii) {
#pragma unroll
		for (int jj = 0; jj < 3; ++jj) {
			const double delta_ij = (ii == jj) ? 1 : 0;
			T[ii][jj] = Tdiag * delta_ij + Tf * (n[ii] * n[jj]);
		}
	}
}

// radiation_system.hpp:918-983 + :1054-1131: HLL flux at one face from the L/R primitive states (E, fx, fy, fz);
// `consL/consR` are the cell-centred conserved radiation states either side (first-order fallback).
// `eps0`: the factor of the optional wavespeed correction on the dissipative part of the ENERGY flux (:1098-1117; epsilon = {S_corr, 1, 1, 1}), 1 without it
template <int DIR>
QK_DEV void radFaceFlux(Rad const &r, const double pL[NRAD], const double pR[NRAD], const double consL[NRAD], const double consR[NRAD], double F[NRAD], double eps0 = 1.0)
{
	double erad_L = pL[0], erad_R = pR[0];
	double fx_L = pL[1], fx_R = pR[1];
	double fy_L = pL[2], fy_R = pR[2];
	double fz_L = pL[3], fz_R = pR[3];
	double f_L = sqrtN(fx_L * fx_L + fy_L * fy_L + fz_L * fz_L);
	double f_R = sqrtN(fx_R * fx_R + fy_R * fy_R + fz_R * fz_R);
	double Fx_L = fx_L * (r.c * erad_L);
	double Fx_R = fx_R * (r.c * erad_R);
	double Fy_L = fy_L * (r.c * erad_L);
	double Fy_R = fy_R * (r.c * erad_R);
	double Fz_L = fz_L * (r.c * erad_L);
	double Fz_R = fz_R * (r.c * erad_R);
	if ((erad_L <= 0.) || (erad_R <= 0.) || (f_L >= 1.) || (f_R >= 1.)) {
		erad_L = consL[0];
		erad_R = consR[0];
		Fx_L = consL[1];
		Fx_R = consR[1];
		Fy_L = consL[2];
		Fy_R = consR[2];
		Fz_L = consL[3];
		Fz_R = consR[3];
		fx_L = Fx_L / (r.c * erad_L);
		fx_R = Fx_R / (r.c * erad_R);
		fy_L = Fy_L / (r.c * erad_L);
		fy_R = Fy_R / (r.c * erad_R);
		fz_L = Fz_L / (r.c * erad_L);
		fz_R = Fz_R / (r.c * erad_R);
		f_L = sqrtN(fx_L * fx_L + fy_L * fy_L + fz_L * fz_L);
		f_R = sqrtN(fx_R * fx_R + fy_R * fy_R + fz_R * fz_R);
	}
	double TL[3][3], TR[3][3];
	eddingtonTensor(r, fx_L, fy_L, fz_L, TL);
	eddingtonTensor(r, fx_R, fy_R, fz_R, TR);
	const double FnL = (DIR == 0) ? Fx_L : (DIR == 1) ? Fy_L : Fz_L;
	const double FnR = (DIR == 0) ? Fx_R : (DIR == 1) ? Fy_R : Fz_R;
	double FL[NRAD] = {FnL, TL[DIR][0] * erad_L, TL[DIR][1] * erad_L, TL[DIR][2] * erad_L};
	double FR[NRAD] = {FnR, TR[DIR][0] * erad_R, TR[DIR][1] * erad_R, TR[DIR][2] * erad_R};
	double S_L = smax(0.1, sqrtN(TL[DIR][DIR]));
	double S_R = smax(0.1, sqrtN(TR[DIR][DIR]));
	S_L *= -1.;
	FL[0] *= r.chat / r.c;
	FR[0] *= r.chat / r.c;
#pragma unroll
	for (int n = 1; n < NRAD; ++n) {
		FL[n] *= r.chat * r.c;
		FR[n] *= r.chat * r.c;
	}
	S_L *= r.chat;
	S_R *= r.chat;
	const double UL[NRAD] = {erad_L, Fx_L, Fy_L, Fz_L};
	const double UR[NRAD] = {erad_R, Fx_R, Fy_R, Fz_R};
	const Recip RS = recipOf(S_R - S_L);
	const double sR_over = divBy(S_R, RS), sL_over = divBy(S_L, RS), sRL_over = divBy(S_R * S_L, RS);
#pragma unroll
	for (int n = 0; n < NRAD; ++n) {
		// :1116-1117
		const double epsilon = (n == 0) ? eps0 : 1.0;
		F[n] = sR_over * FL[n] - sL_over * FR[n] + epsilon * sRL_over * (UR[n] - UL[n]);
	}
}

// epsilon of face (i, j, k) and photon group g from the arrays qk_rad_ComputeWavespeedCorrection filled (NULL: no correction)
QK_DEV auto faceEpsilon(const qk_array4 *eps_t, int b, int i, int j, int k, int g) -> double
{
	if (eps_t == nullptr) {
		return 1.0;
	}
	return RA4(eps_t[b])(i, j, k, g);
}

// radiation_system.hpp:626-665
QK_DEV auto radStateValid(Rad const &r, const double U[NRAD]) -> bool
{
	const double Fnorm = sqrtN(U[1] * U[1] + U[2] * U[2] + U[3] * U[3]);
	const double f = divN(Fnorm, r.c * U[0]);
	return (U[0] > 0.) && (f <= 1.);
}
QK_DEV void amendRadState(Rad const &r, double U[NRAD])
{
	double E_r = U[0];
	if (E_r < r.Erad_floor) {
		E_r = r.Erad_floor;
		U[0] = r.Erad_floor;
	}
	const double Fx = U[1], Fy = U[2], Fz = U[3];
	if (Fx * Fx + Fy * Fy + Fz * Fz > r.c * r.c * E_r * E_r) {
		const double Fnorm = sqrt(Fx * Fx + Fy * Fy + Fz * Fz);
		U[1] = Fx / Fnorm * r.c * E_r;
		U[2] = Fy / Fnorm * r.c * E_r;
		U[3] = Fz / Fnorm * r.c * E_r;
	}
}

// EOS.hpp:202-244 with the direct gamma-law forms
QK_DEV auto eintTempDerivative(Eos const &eos, double rho, double T) -> double
{
	if (eos.tmodel == 1) { // the heat capacity of the Su-Olson material: alpha T^3
		return eos.alpha * ((T * T) * T);
	}
	const double p = rho * T * Eos::k_B / (eos.mu * Eos::m_u);
	const double e = p / (eos.gm1 * rho);
	const double dedT = e / T;
	return dedT * rho * eos.kB_user / Eos::k_B;
}

// quokka::EOS for ONE cell inside the Newton iteration: Eos::tgasFromEint and eintTempDerivative above with the quotients whose denominators do
// not change from one iteration to the next — rho, (gamma - 1) rho, k_B, mu m_u, the user's k_B — taken through their refined reciprocals
// (divBy: the same bits as `/` for normal-range operands, qk_device.hpp).  Ten of the ~14 divisions of a Newton iteration are of this kind
// or share a denominator; an FP64 division costs ~14 issue slots, divBy three.
struct EosCell {
	Eos const &eos;
	double rho;
	Recip Rrho, Rg;
	Recip const &RkB, &Rmu, &RkBu; // run-time constants: their reciprocals come with the Eos (formed on the host, scalar registers)
	QK_DEV EosCell(Eos const &e, double rho_) : eos(e), rho(rho_), Rrho(recipOf(rho_)), Rg(recipOf(e.gm1 * rho_)), RkB(e.RkB), Rmu(e.Rmu), RkBu(e.RkBu) {}
	QK_DEV auto tgasFromEint(double Eint) const -> double
	{
		if (eos.tmodel == 1) {
			return eos.tgasFromEint(rho, Eint);
		}
		const double e = divBy(Eint, Rrho);
		const double T = divBy(e * eos.mu * Eos::m_u * eos.gm1, RkB);
		return divBy(T * Eos::k_B, RkBu);
	}
	QK_DEV auto eintFromTgas(double T) const -> double { return eos.eintFromTgas(rho, T); }
	QK_DEV auto eintTempDerivative(double T) const -> double
	{
		if (eos.tmodel == 1) {
			return eos.alpha * ((T * T) * T);
		}
		const double p = divBy(rho * T * Eos::k_B, Rmu);
		const double e = divBy(p, Rg);
		const double dedT = e / T;
		return divBy(dedT * rho * eos.kB_user, RkB);
	}
};

QK_DEV auto eintFromEgas(double rho, double px, double py, double pz, double Etot) -> double
{
	const double p_sq = px * px + py * py + pz * pz;
	const double Ekin = p_sq / (2.0 * rho);
	return Etot - Ekin;
}
QK_DEV auto egasFromEint(double rho, double px, double py, double pz, double Eint) -> double
{
	const double p_sq = px * px + py * py + pz * pz;
	const double Ekin = p_sq / (2.0 * rho);
	return Eint + Ekin;
}
// the same two with the refined reciprocal of rho: a / (2 rho) == 0.5 (a / rho), scaling by two is exact
QK_DEV auto eintFromEgas(Recip const &Rrho, double px, double py, double pz, double Etot) -> double
{
	return Etot - 0.5 * divBy(px * px + py * py + pz * pz, Rrho);
}
QK_DEV auto egasFromEint(Recip const &Rrho, double px, double py, double pz, double Eint) -> double
{
	return Eint + 0.5 * divBy(px * px + py * py + pz * pz, Rrho);
}

// source_terms_single_group.hpp:29-563 for one cell.  U[10] in place; counters as in the reference:
// it_counter[0] += 1, [1] += n+1, [2] = max(n+1); fail[0] Newton failure, fail[2] outer-iteration failure.
// radiation_system.hpp:1420-1483 (nGroups_ == 1) with BackwardEulerOneVariable (:1387-1418): the dust temperature between gas and radiation
template <bool TDEP, class RadT> QK_DEV auto dustTemperatureBateKeto(RadT const &r, double T_gas, double T_d_init, double rho, double Erad0, double N_d, double dt, double R_sum, int n_step) -> double
{
	if (n_step > 0) {
		return T_gas - R_sum / (N_d * sqrt(T_gas));
	}
	const double Lambda_compare = N_d * sqrt(T_gas) * T_gas;
	double x = T_d_init;
	const double rel_tol = 1.0e-8;
	const double rel_change_tol = 1.0e-6;
	const int max_iter_td = 100;
	int iter_Td = 0;
	for (; iter_Td < max_iter_td; ++iter_Td) {
		const double fourPiBoverC = r.thermalRadiationHook(x);
		const double kE = r.template kappaE<TDEP>(rho, x);
		const double kP = r.template kappaP<TDEP>(rho, x);
		const double the_rhs = r.chat * dt * rho * (kE * Erad0 - kP * fourPiBoverC) + N_d * sqrt(T_gas) * (T_gas - x);
		if (fabs(the_rhs) < rel_tol * Lambda_compare) {
			break;
		}
		const double jac = -r.chat * dt * rho * (r.template kappaP<TDEP>(rho, x) * r.thermalRadiationTempDerivativeHook(x)) - N_d * sqrt(T_gas);
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

// DUST: ISM_Traits::enable_dust_gas_thermal_coupling_model (its own instantiation: the gas-radiation kernel keeps its registers)
// RadT / EosT: the objects that answer the problem hooks.  The library's own instantiation uses Rad (closed opacity / emission sets) and EosCell
// (gamma law or the T^4 material); a problem's translation unit instantiates the same function with objects whose members call the problem's
// compiled ComputePlanckOpacity / ComputeEnergyMeanOpacity / ComputeFluxMeanOpacity / ComputeThermalRadiation* / quokka::EOS hooks
// (quokka_amd/host/qk_problem_kernels.hpp) — radiation_system.hpp:1141-1154, EOS.hpp:74-244.
// BETA: RadSystem_Traits::beta_order as a compile-time constant (0 or 1; -1: read from the traits at run time) — what the reference's kernel sees,
// whose beta_order is a constexpr of the problem.  The Lorentz factors are then the literal 1.0 (x * 1.0 is x: same bits) and the std::pow / 3 x 3
// solve of the beta_order >= 2 branches leave the kernel.
template <bool TDEP = false, bool DUST = false, class RadT = Rad, class EosT = EosCell, int BETA = -1>
QK_DEV void radSourceCell(RadT const &r, Eos const &eos, double U[10], double srcval, double dt_radiation, int stage, int &n_newton_total, int &n_newton_max,
			  int &n_solves, int &fail_newton, int &fail_outer, int *fail_dust = nullptr)
{
	double dt = dt_radiation;
	if (stage == 2) {
		dt = (1.0 - IMEX_a32) * dt_radiation;
	}
	const double c = r.c;
	const double chat = r.chat;
	const double rho = U[RHO];
	const double x1GasMom0 = U[MX], x2GasMom0 = U[MY], x3GasMom0 = U[MZ];
	const double gasMtm0[3] = {x1GasMom0, x2GasMom0, x3GasMom0};
	const double Egastot0 = U[ENE];
	const double Erad0 = U[RAD0];
	const double Src = srcval * dt * chat;
	const double Frad_t0[3] = {U[RAD0 + 1], U[RAD0 + 2], U[RAD0 + 3]};
	const bool gamma_ne_1 = !eos.isothermal;
	const int beta_order = (BETA >= 0) ? BETA : r.beta_order;

	double Egas0 = __builtin_nan(""), Ekin0 = __builtin_nan(""), Etot0 = __builtin_nan(""), Egas_guess = __builtin_nan("");
	double T_gas = __builtin_nan(""), T_d = __builtin_nan("");
	double lorentz_factor = __builtin_nan(""), lorentz_factor_v = __builtin_nan(""), lorentz_factor_v_v = __builtin_nan("");
	double fourPiBoverC = __builtin_nan(""), Erad_guess = __builtin_nan(""), kappaP = __builtin_nan(""), kappaE = __builtin_nan("");
	double kappaF = __builtin_nan(""), kappaPoverE = __builtin_nan("");
	double work = 0.0, work_prev = 0.0;
	double dMomentum[3] = {0., 0., 0.};
	double Frad_t1[3] = {0., 0., 0.};
	const double cscale = r.cscale;       // c / chat
	const Recip Rcc{c * chat, r.r_cchat}; // (host-made reciprocals of run-time constants)
	const Recip Rc2{c * c, r.r_cc};
	const Recip Rrho = recipOf(rho); // (the same value EosCell refines: merged by the compiler)
	const EosT ec(eos, rho);

	if (gamma_ne_1) {
		Egas0 = eintFromEgas(Rrho, x1GasMom0, x2GasMom0, x3GasMom0, Egastot0);
		Etot0 = Egas0 + cscale * (Erad0 + Src);
	}
	double gas_update_factor = 1.0;
	if (stage == 1) {
		gas_update_factor = IMEX_a32;
	}
	// source_terms_single_group.hpp:89-98
	double coeff_n = __builtin_nan("");
	const double H_num_den = rho / r.mean_molecular_mass; // ComputeNumberDensityH (radiation_system.hpp:463-467); read by the ISM hooks and the dust model
	if constexpr (DUST) {
		coeff_n = dt * r.dust_coeff * H_num_den * H_num_den / cscale;
	}

	const int max_ite = 5;
	int ite = 0;
	for (; ite < max_ite; ++ite) {
		double R = __builtin_nan("");
		Erad_guess = Erad0;
		if (gamma_ne_1) {
			double tau0 = __builtin_nan("");
			double tau = __builtin_nan("");
			Recip Rtau{1.0, 1.0};
			Egas_guess = Egas0;
			Ekin0 = Egastot0 - Egas0;
			const double betaSqr = (x1GasMom0 * x1GasMom0 + x2GasMom0 * x2GasMom0 + x3GasMom0 * x3GasMom0) / (rho * rho * c * c);
			if ((beta_order == 0) || (beta_order == 1)) {
				lorentz_factor = 1.0;
				lorentz_factor_v = 1.0;
			} else if (beta_order == 2) {
				lorentz_factor = 1.0 + 0.5 * betaSqr;
				lorentz_factor_v = 1.0;
				lorentz_factor_v_v = 1.0;
			} else if (beta_order == 3) {
				lorentz_factor = 1.0 + 0.5 * betaSqr;
				lorentz_factor_v = 1.0 + 0.5 * betaSqr;
				lorentz_factor_v_v = 1.0;
			} else {
				lorentz_factor = 1.0 / sqrt(1.0 - betaSqr);
				lorentz_factor_v = lorentz_factor;
				lorentz_factor_v_v = lorentz_factor;
			}

			double F_G, deltaEgas, deltaR, F_D;
			const double resid_tol = 1.0e-11;
			const int maxIter = 100;
			int n = 0;
			for (; n < maxIter; ++n) {
				T_gas = ec.tgasFromEint(Egas_guess);
				if constexpr (!DUST) {
					T_d = T_gas;
					fourPiBoverC = r.thermalRadiation(T_d);
				} else { // :165-177
					T_d = dustTemperatureBateKeto<TDEP>(r, T_gas, T_gas, rho, Erad_guess, coeff_n, dt, R, n);
					if (T_d < 0.0 && fail_dust != nullptr) {
						*fail_dust += 1;
					}
					fourPiBoverC = r.thermalRadiationHook(T_d);
				}
				kappaP = r.template kappaP<TDEP>(rho, T_d);
				kappaE = r.template kappaE<TDEP>(rho, T_d);
				if (kappaE > 0.0) {
					// (x / x is exactly 1 for finite x: equal Planck and energy means — the common case — skip the division)
					kappaPoverE = (kappaP == kappaE) ? 1.0 : kappaP / kappaE;
				} else {
					kappaPoverE = 1.0;
				}
				if (n == 0) {
					kappaF = r.template kappaF<TDEP>(rho, T_d);
					if (beta_order != 0) { // include_work_term_in_source = true
						if (ite == 0) {
							work = divBy((x1GasMom0 * Frad_t0[0] + x2GasMom0 * Frad_t0[1] + x3GasMom0 * Frad_t0[2]) * (2.0 * kappaE - kappaF) * chat, Rc2) *
							       lorentz_factor_v * dt;
						}
					}
					tau0 = dt * rho * kappaP * chat * lorentz_factor;
					tau = tau0;
					if (tau > 0.0) {
						Rtau = recipOf(tau);
					}
					R = (fourPiBoverC - ((kappaPoverE == 1.0) ? Erad_guess : Erad_guess / kappaPoverE)) * tau0 + work; // (y / 1 is y exactly)
					tau0 = smax(tau0, 1.0);
				} else {
					tau = dt * rho * kappaP * chat * lorentz_factor;
					if (tau > 0.0) {
						Rtau = recipOf(tau); // (also divides kappaPoverE in the Jacobian below)
						Erad_guess = kappaPoverE * (fourPiBoverC - divBy(R - work, Rtau));
					}
				}
				// the ISM hooks (closed set, carried with the dust model): net line cooling linear in T into the group, constant cosmic-ray heating
				double cooling = 0.0;
				const double cooling_derivative = 0.0; // (:284-287: read by the gas-only Jacobian, where the cooling hook is not evaluated)
				const double CR_heating = r.cosmicRayHeatingRate(H_num_den) * dt; // :233 (the closed set carries a non-zero rate with the dust model only)
				if constexpr (DUST) { // :234-237
					cooling = r.netCoolingRate(T_gas, H_num_den);
				}
				F_G = Egas_guess - Egas0 + cscale * R + cooling * dt - CR_heating;
				F_D = Erad_guess - Erad0 - (R + Src);
				double F_D_abs;
				if (tau > 0.0) {
					F_D_abs = fabs(F_D);
				} else {
					F_D_abs = fabs(F_D + R);
				}
				if ((fabs(F_G) < resid_tol * Etot0) && (cscale * F_D_abs < resid_tol * Etot0)) {
					break;
				}
				const double c_v = ec.eintTempDerivative(T_gas);
				const Recip Rcv = recipOf(c_v);
				const double d_fourpiboverc_d_t = DUST ? r.thermalRadiationTempDerivativeHook(T_d) : r.thermalRadiationTempDerivative(T_d);
				double dEg_dT = kappaPoverE * d_fourpiboverc_d_t;
				double J00, J01, J10, J11;
				if constexpr (!DUST) {
					J00 = 1.0 + divBy(cooling_derivative * dt, Rcv);
					J01 = cscale;
					J10 = divBy(1.0, Rcv) * dEg_dT - r.inv_cscale * cooling_derivative * dt;
					if (tau <= 0.0) {
						J11 = -__builtin_inf();
					} else {
						J11 = divBy(-1.0 * kappaPoverE, Rtau) - 1.0;
					}
				} else { // :293-305
					const double d_Td_d_T = 3. / 2. - T_d / (2. * T_gas);
					dEg_dT *= d_Td_d_T;
					const double dTd_dRg = -1.0 / (coeff_n * sqrt(T_gas));
					J00 = 1.0;
					J01 = cscale;
					J10 = divBy(1.0, Rcv) * dEg_dT;
					if (tau <= 0.0) {
						J11 = -1.0e100; // LARGE (:7)
					} else {
						J11 = kappaPoverE * d_fourpiboverc_d_t * dTd_dRg - divBy(kappaPoverE, Rtau) - 1.0;
					}
				}
				const double y0 = -F_G;
				const double y1 = -1. * F_D;
				const double det = J00 * J11 - J01 * J10;
				if (tau > 0.0) { // det is finite: both quotients through its reciprocal
					const Recip Rdet = recipOf(det);
					deltaEgas = divBy(J11 * y0 - J01 * y1, Rdet);
					deltaR = divBy(J00 * y1 - J10 * y0, Rdet);
				} else { // plain divisions: det = -inf (or -LARGE), where IEEE inf arithmetic is part of the algorithm
					deltaEgas = (J11 * y0 - J01 * y1) / det;
					deltaR = (J00 * y1 - J10 * y0) / det;
				}
				// enable_dE_constrain = true (radiation_system.hpp:44): the step is cut when deltaEgas / c_v > max(T_gas, T_rad).  The radiation
				// temperature (a division and two square roots) can only matter when the quotient already exceeds T_gas: evaluated then, not
				// in every iteration — the same decision and the same values.
				const double dT_step = divBy(deltaEgas, Rcv);
				bool cut = false;
				if (dT_step > T_gas) {
					const double T_rad = sqrt(sqrt(Erad_guess / r.arad));
					if (dT_step > smax(T_gas, T_rad)) {
						cut = true;
						Egas_guess = ec.eintFromTgas(T_rad);
					}
				}
				if (!cut) {
					Egas_guess += deltaEgas;
					R += deltaR;
				}
			}
			if (n >= maxIter) {
				fail_newton += 1;
			}
			n_solves += 1;
			n_newton_total += n + 1;
			n_newton_max = max(n_newton_max, n + 1);
			// :351-356: the energy the line cooled away goes to the radiation
			if constexpr (DUST) {
				const double cooling_tend = r.netCoolingRate(T_gas, H_num_den) * dt;
				Erad_guess += r.inv_cscale * cooling_tend;
			} else {
				Erad_guess += r.inv_cscale * (0.0 * dt);
			}
			if (n > 0) {
				kappaF = r.template kappaF<TDEP>(rho, T_d);
			}
		} else {
			T_d = T_gas;
			kappaF = r.template kappaF<TDEP>(rho, T_d);
		}

		// 2. radiation flux update
		dMomentum[0] = dMomentum[1] = dMomentum[2] = 0.;
		if (gamma_ne_1 && (beta_order != 0)) {
			const double erad = Erad_guess;
			double v_terms[3];
			const Recip RcE = recipOf(r.c * erad);
			const double fx = divBy(Frad_t0[0], RcE);
			const double fy = divBy(Frad_t0[1], RcE);
			const double fz = divBy(Frad_t0[2], RcE);
			const double F_coeff = chat * rho * kappaF * dt * lorentz_factor;
			double Tedd[3][3];
			eddingtonTensor(r, fx, fy, fz, Tedd);
#pragma unroll
			for (int n = 0; n < 3; ++n) {
				double Planck_term = kappaP * fourPiBoverC * lorentz_factor_v;
				if (kappaF != kappaE) {
					Planck_term += (kappaF - kappaE) * erad * pow(lorentz_factor_v, 3.0);
				}
				Planck_term *= chat * dt * gasMtm0[n];
				double pressure_term = 0.0;
#pragma unroll
				for (int z = 0; z < 3; ++z) {
					pressure_term += gasMtm0[z] * Tedd[n][z] * erad;
				}
				pressure_term *= chat * dt * kappaF * lorentz_factor_v;
				v_terms[n] = Planck_term + pressure_term;
			}
			if (beta_order == 1 || kappaF == kappaE) {
				const Recip R1F = recipOf(1.0 + F_coeff);
#pragma unroll
				for (int n = 0; n < 3; ++n) {
					Frad_t1[n] = divBy(Frad_t0[n] + v_terms[n], R1F);
					dMomentum[n] += divBy(-(Frad_t1[n] - Frad_t0[n]), Rcc);
				}
			} else {
				// Solve3x3matrix (radiation_system.hpp:560-579) with gasVel = 0 as in the reference (:437, never assigned)
				const double K0 = 2.0 * rho * chat * dt * (kappaF - kappaE) / c / c * pow(lorentz_factor_v_v, 3.0);
				const double C00 = 1.0 + F_coeff + K0 * 0. * 0., C11 = C00, C22 = C00;
				const double C01 = K0 * 0. * 0., C02 = C01, C10 = C01, C12 = C01, C20 = C01, C21 = C01;
				const double Y0 = v_terms[0] + Frad_t0[0], Y1 = v_terms[1] + Frad_t0[1], Y2 = v_terms[2] + Frad_t0[2];
				const double E11 = C11 - C01 * C10 / C00;
				const double E12 = C12 - C02 * C10 / C00;
				const double E21 = C21 - C01 * C20 / C00;
				const double E22 = C22 - C02 * C20 / C00;
				const double Z1 = Y1 - Y0 * C10 / C00;
				const double Z2 = Y2 - Y0 * C20 / C00;
				const double X2 = (Z2 - Z1 * E21 / E11) / (E22 - E12 * E21 / E11);
				const double X1 = (Z1 - E12 * X2) / E11;
				const double X0 = (Y0 - C01 * X1 - C02 * X2) / C00;
				Frad_t1[0] = X0;
				Frad_t1[1] = X1;
				Frad_t1[2] = X2;
#pragma unroll
				for (int n = 0; n < 3; ++n) {
					dMomentum[n] += -(Frad_t1[n] - Frad_t0[n]) / (c * chat);
				}
			}
		} else {
#pragma unroll
			for (int n = 0; n < 3; ++n) {
				Frad_t1[n] = Frad_t0[n] / (1.0 + rho * kappaF * chat * dt);
				dMomentum[n] += -(Frad_t1[n] - Frad_t0[n]) / (c * chat);
			}
		}
		const double x1GasMom1 = U[MX] + dMomentum[0];
		const double x2GasMom1 = U[MY] + dMomentum[1];
		const double x3GasMom1 = U[MZ] + dMomentum[2];

		// 3. work term
		if (gamma_ne_1 && (beta_order != 0)) {
			const double Egastot1 = egasFromEint(Rrho, x1GasMom1, x2GasMom1, x3GasMom1, Egas_guess);
			const double Ekin1 = Egastot1 - Egas_guess;
			const double dEkin_work = Ekin1 - Ekin0;
			Egas_guess -= dEkin_work;
		}
		if ((beta_order == 0) || !gamma_ne_1) {
			break;
		}
		work_prev = work;
		work = divBy((x1GasMom1 * Frad_t1[0] + x2GasMom1 * Frad_t1[1] + x3GasMom1 * Frad_t1[2]) * chat, Rc2) * lorentz_factor_v * (2.0 * kappaE - kappaF) * dt;
		const double lag_tol = 1.0e-13;
		if ((fabs(work) == 0.0) || (cscale * fabs(work - work_prev) < lag_tol * Etot0) || (fabs(work - work_prev) <= lag_tol * R) ||
		    (fabs(work - work_prev) <= 1.0e-8 * fabs(work))) {
			break;
		}
	}
	if (ite >= max_ite) {
		fail_outer += 1;
	}

	// 4b. store
	const double x1GasMom1 = U[MX] + dMomentum[0] * gas_update_factor;
	const double x2GasMom1 = U[MY] + dMomentum[1] * gas_update_factor;
	const double x3GasMom1 = U[MZ] + dMomentum[2] * gas_update_factor;
	U[MX] = x1GasMom1;
	U[MY] = x2GasMom1;
	U[MZ] = x3GasMom1;
	if (gamma_ne_1) {
		Egas_guess = Egas0 + (Egas_guess - Egas0) * gas_update_factor;
		U[EINT] = Egas_guess;
		U[ENE] = egasFromEint(Rrho, x1GasMom1, x2GasMom1, x3GasMom1, Egas_guess);
		U[RAD0] = Erad_guess;
	}
	U[RAD0 + 1] = Frad_t1[0];
	U[RAD0 + 2] = Frad_t1[1];
	U[RAD0 + 3] = Frad_t1[2];
}

} // namespace qk

#endif // QK_RAD_DEVICE_HPP_
