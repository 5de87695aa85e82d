// ORACLE — TEST INFRASTRUCTURE ONLY (see grid.hpp header).
//
// eos.hpp: restatement of quokka::EOS<problem_t> (reference src/hydro/EOS.hpp:40-383) for the
// gamma-law branch (no CHEMISTRY).  The arithmetic behind every `eos(eos_input_*, estate)` call
// lives in the un-vendored Microphysics submodule (reference .gitmodules:10-13,
// psharda/Microphysics branch `development`, no SHA recorded; network `gamma_law`,
// reference src/CMakeLists.txt:118-122).  PARITY UNPINNED at the ulp level for this file:
// the published gamma_law EOS algorithm is restated below in two association variants and the
// one that is used (kEosVariant) is the one that passes the reference's own zero-tolerance
// known-answer test HydroContact (src/problems/HydroContact/test_hydro_contact.cpp:213-216);
// see tests/test_oracle_known_answers.py and DESIGN.md §Oracle.
#ifndef ORACLE_EOS_HPP_
#define ORACLE_EOS_HPP_

#include <cmath>
#include <limits>

namespace oracle
{

// Microphysics fundamental_constants.H (CODATA 2018, cgs) — used through C::k_B, C::m_u,
// C::c_light, C::a_rad at reference EOS.hpp:36,107 and radiation_system.hpp:58-59,78.
namespace C
{
constexpr double k_B = 1.380649e-16;
constexpr double m_u = 1.6605390666e-24;
constexpr double m_p = 1.67262192369e-24; // CODATA 2018, as Microphysics' fundamental_constants.H (not vendored)
constexpr double m_e = 9.1093837015e-28;
constexpr double c_light = 2.99792458e10;
constexpr double sigma_SB = 5.670374419e-5;
constexpr double a_rad = 4.0 * sigma_SB / c_light;
constexpr double hplanck = 6.62607015e-27;
constexpr double ev2erg = 1.602176634e-12;
} // namespace C

// EOS variant:
//  0 = direct gamma-law forms  p = (gamma-1) rho e,  e = p / ((gamma-1) rho)
//  1 = temperature round trip as in the upstream AMReX-Astro gamma_law actual_eos.H
//      (T from the input pair, then p = rho T k_B/(mu m_u), e = p/(gamma-1)/rho)
#ifndef ORACLE_EOS_VARIANT
#define ORACLE_EOS_VARIANT 0
#endif
constexpr int kEosVariant = ORACLE_EOS_VARIANT;

// runtime stand-in for quokka::EOS_Traits<problem_t> (reference EOS.hpp:32-37)
struct EOSTraits {
	// temperature hooks a problem may specialise (EOS.hpp:74-244): 0 gamma law; 1 E_int = (alpha / 4) T^4, the Su-Olson material of
	// RadMatterCoupling / RadSuOlson / RadMarshak (e.g. src/problems/RadMatterCoupling/test_radiation_matter_coupling.cpp:72-103).
	// The fourth root is taken as two square roots (the reference calls std::pow(x, 1. / 4.)): identical in the GPU build.
	int temperature_model = 0;
	double alpha = 0.0;
	double gamma = 5. / 3.;
	double cs_isothermal = std::numeric_limits<double>::quiet_NaN();
	double mean_molecular_weight = std::numeric_limits<double>::quiet_NaN();
	double boltzmann_constant = C::k_B;
};

// what Microphysics' chem_eos_t carries for the gamma_law EOS
struct eos_state {
	double rho = NAN, T = NAN, p = NAN, e = NAN, mu = NAN;
	double dpdT = NAN, dpdr = NAN, dedT = NAN, dedr = NAN, dpde = NAN, cs = NAN, G = NAN;
};

enum eos_input { eos_input_rt, eos_input_re, eos_input_rp };

// the gamma_law `actual_eos` (Microphysics EOS/gamma_law/actual_eos.H, published algorithm)
inline void eos(eos_input input, eos_state &s, double gamma)
{
	const double m_nucleon = C::m_u;
	if constexpr (kEosVariant == 1) {
		switch (input) {
		case eos_input_rt:
			break;
		case eos_input_rp:
			s.T = s.p * s.mu * m_nucleon / (C::k_B * s.rho);
			break;
		case eos_input_re:
			s.T = s.e * s.mu * m_nucleon * (gamma - 1.0) / C::k_B;
			break;
		}
		const double Tinv = 1.0 / s.T;
		const double rhoinv = 1.0 / s.rho;
		const double pressure = s.rho * s.T * C::k_B / (s.mu * m_nucleon);
		const double energy = pressure / (gamma - 1.0) * rhoinv;
		s.p = pressure;
		s.e = energy;
		s.dpdT = s.p * Tinv;
		s.dpdr = s.p * rhoinv;
		s.dedT = s.e * Tinv;
		s.dedr = 0.0;
		s.dpde = s.dpdT / s.dedT;
		s.cs = std::sqrt(gamma * s.p * rhoinv);
		s.G = 0.5 * (1.0 + gamma);
	} else {
		// direct forms: never round-trip the (p, e) pair through T
		switch (input) {
		case eos_input_rt:
			s.p = s.rho * s.T * C::k_B / (s.mu * m_nucleon);
			s.e = s.p / ((gamma - 1.0) * s.rho);
			break;
		case eos_input_rp:
			s.e = s.p / ((gamma - 1.0) * s.rho);
			s.T = s.p * s.mu * m_nucleon / (C::k_B * s.rho);
			break;
		case eos_input_re:
			s.p = (gamma - 1.0) * s.rho * s.e;
			s.T = s.e * s.mu * m_nucleon * (gamma - 1.0) / C::k_B;
			break;
		}
		s.dpdT = s.p / s.T;
		s.dpdr = s.p / s.rho;
		s.dedT = s.e / s.T;
		s.dedr = 0.0;
		s.dpde = (gamma - 1.0) * s.rho;
		s.cs = std::sqrt(gamma * s.p / s.rho);
		s.G = 0.5 * (1.0 + gamma);
	}
}

// quokka::EOS<problem_t> (reference EOS.hpp:74-383), non-CHEMISTRY branch, gamma != 1 guard kept
struct EOS {
	EOSTraits tr;

	// EOS.hpp:74-114
	[[nodiscard]] auto ComputeTgasFromEint(double rho, double Eint) const -> double
	{
		if (tr.temperature_model == 1) {
			return std::sqrt(std::sqrt(4.0 * Eint / tr.alpha));
		}
		double Tgas = NAN;
		if (tr.gamma != 1.0) {
			eos_state estate;
			estate.rho = rho;
			estate.e = Eint / rho;
			estate.mu = tr.mean_molecular_weight / C::m_u;
			eos(eos_input_re, estate, tr.gamma);
			Tgas = estate.T * C::k_B / tr.boltzmann_constant;
		}
		return Tgas;
	}

	// EOS.hpp:116-159
	[[nodiscard]] auto ComputeEintFromTgas(double rho, double Tgas) const -> double
	{
		if (tr.temperature_model == 1) {
			return (tr.alpha / 4.0) * ((Tgas * Tgas) * (Tgas * Tgas));
		}
		double Eint = NAN;
		if (tr.gamma != 1.0) {
			eos_state estate;
			estate.rho = rho;
			estate.T = Tgas;
			estate.mu = tr.mean_molecular_weight / C::m_u;
			eos(eos_input_rt, estate, tr.gamma);
			Eint = estate.e * rho * tr.boltzmann_constant / C::k_B;
		}
		return Eint;
	}

	// EOS.hpp:161-200
	[[nodiscard]] auto ComputeEintFromPres(double rho, double Pressure) const -> double
	{
		double Eint = NAN;
		if (tr.gamma != 1.0) {
			eos_state estate;
			estate.rho = rho;
			estate.p = Pressure;
			estate.mu = tr.mean_molecular_weight / C::m_u;
			eos(eos_input_rp, estate, tr.gamma);
			Eint = estate.e * rho;
		}
		return Eint;
	}

	// EOS.hpp:202-244
	[[nodiscard]] auto ComputeEintTempDerivative(double rho, double Tgas) const -> double
	{
		if (tr.temperature_model == 1) {
			return tr.alpha * ((Tgas * Tgas) * Tgas);
		}
		double dEint_dT = NAN;
		if (tr.gamma != 1.0) {
			eos_state estate;
			estate.rho = rho;
			estate.T = Tgas;
			estate.mu = tr.mean_molecular_weight / C::m_u;
			eos(eos_input_rt, estate, tr.gamma);
			dEint_dT = estate.dedT * rho * tr.boltzmann_constant / C::k_B;
		}
		return dEint_dT;
	}

	struct Derivs {
		double deint_dRho, deint_dP, dRho_dP, dP_dRho_s, G;
	};

	// EOS.hpp:246-302
	[[nodiscard]] auto ComputeOtherDerivatives(double rho, double P) const -> Derivs
	{
		Derivs d{NAN, NAN, NAN, NAN, NAN};
		if (tr.gamma != 1.0) {
			eos_state estate;
			estate.rho = rho;
			estate.p = P;
			estate.mu = tr.mean_molecular_weight / C::m_u;
			eos(eos_input_rp, estate, tr.gamma);
			d.deint_dRho = estate.dedr;
			d.deint_dP = 1.0 / estate.dpde;
			d.dRho_dP = 1.0 / (estate.dpdr * C::k_B / tr.boltzmann_constant);
			d.dP_dRho_s = estate.cs * estate.cs;
			d.G = estate.G;
		}
		return d;
	}

	// EOS.hpp:304-348
	[[nodiscard]] auto ComputePressure(double rho, double Eint) const -> double
	{
		double P = NAN;
		if (tr.gamma != 1.0) {
			eos_state estate;
			estate.rho = rho;
			if (rho == 0.0) {
				estate.e = 0;
			} else {
				estate.e = Eint / rho;
			}
			estate.mu = tr.mean_molecular_weight / C::m_u;
			eos(eos_input_re, estate, tr.gamma);
			P = estate.p;
		}
		return P;
	}

	// EOS.hpp:350-383
	[[nodiscard]] auto ComputeSoundSpeed(double rho, double Pressure) const -> double
	{
		double cs = NAN;
		if (tr.gamma != 1.0) {
			eos_state estate;
			estate.rho = rho;
			estate.p = Pressure;
			estate.mu = tr.mean_molecular_weight / C::m_u;
			eos(eos_input_rp, estate, tr.gamma);
			cs = estate.cs;
		}
		return cs;
	}
};

} // namespace oracle

#endif // ORACLE_EOS_HPP_
