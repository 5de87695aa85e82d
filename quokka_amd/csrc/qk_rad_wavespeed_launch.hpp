// qk_rad_wavespeed_launch.hpp — RadSystem<problem_t>::ComputeCellOpticalDepth (reference src/radiation/radiation_system.hpp:803-871) and the factor
// the optional wavespeed correction of ComputeFluxes puts on the dissipative part of the radiation-ENERGY flux (:1019-1022, :1098-1109):
//     epsilon(face) = min(1, 1 / tau_cell)  where (i + j + k) is even,  1 elsewhere
//     tau_cell      = harmonic mean of the optical depths  dl rho kappa  of the two cells of the face, kappa = the flux-mean opacity (one group) or the
//                     bin-centre opacity of each group (DefineOpacityExponentsAndLowerValues + ComputeBinCenterOpacity).
// The transport kernels (qk_rad_ops.hip) never call an opacity hook: this pass writes epsilon of every face and group into face arrays they read.
// Like the source term it is instantiated twice: by the library with the closed hook sets of qk_rad_traits, and in a problem's own translation unit
// with the problem's compiled hooks (host/qk_problem_kernels.hpp).
#ifndef QK_RAD_WAVESPEED_LAUNCH_HPP_
#define QK_RAD_WAVESPEED_LAUNCH_HPP_

#include "qk_rad_mg_launch.hpp"
#include "qk_rad_source_launch.hpp"

namespace qk
{

// (rho, T) of one cell as ComputeCellOpticalDepth forms them: piecewise-constant gas state, E_int = E - p^2 / (2 rho), T from the EOS (NaN for an
// isothermal gas, :837-847)
template <class EosT> QK_DEV void gasDensityAndTemperature(Eos const &eos, RA4 const &U, int i, int j, int k, double &rho, double &T)
{
	const int64_t c = U.idx(i, j, k);
	rho = U.p[c + U.ns * RHO];
	T = __builtin_nan("");
	if (!eos.isothermal) {
		const double Eint = eintFromEgas(rho, U.p[c + U.ns * MX], U.p[c + U.ns * MY], U.p[c + U.ns * MZ], U.p[c + U.ns * ENE]);
		const EosT ec(eos, rho);
		T = ec.tgasFromEint(Eint);
	}
}

QK_DEV auto wavespeedFactor(double tau_L, double tau_R) -> double
{
	const double tau = (tau_L * tau_R * 2.) / (tau_L + tau_R); // harmonic mean (:862)
	const double inv = 1.0 / tau;
	return (inv < 1.0) ? inv : 1.0; // std::min(1.0, 1.0 / tau)
}

// one photon group: kappa = ComputeFluxMeanOpacity(rho, T)
template <int DIR, bool TDEP, class RadT, class EosT>
void launchWavespeedSingle(qk_level *lev, qk_stream s, RadT rad, Eos eos, const qk_array4 *cons_t, qk_array4 *eps_t, double dl)
{
	launchRad(lev, s, 0, DIR, "rad_ComputeCellOpticalDepth", [=] __device__(int b, int i, int j, int k, bool valid) {
		if (!valid) {
			return;
		}
		WA4 E(eps_t[b]);
		double eps = 1.0;
		if ((i + j + k) % 2 == 0) { // no correction for odd zones (:1104)
			RA4 U(cons_t[b]);
			double rho_L, T_L, rho_R, T_R;
			gasDensityAndTemperature<EosT>(eos, U, i - unit(DIR, 0), j - unit(DIR, 1), k - unit(DIR, 2), rho_L, T_L);
			gasDensityAndTemperature<EosT>(eos, U, i, j, k, rho_R, T_R);
			const double tau_L = dl * rho_L * rad.template kappaF<TDEP>(rho_L, T_L);
			const double tau_R = dl * rho_R * rad.template kappaF<TDEP>(rho_R, T_R);
			eps = wavespeedFactor(tau_L, tau_R);
		}
		E(i, j, k, 0) = eps;
	});
}

// NG photon groups: kappa_g = lower value of the group's lower edge times (edge ratio)^(exponent / 2)
template <int DIR, int NG, class MC>
void launchWavespeedMG(qk_level *lev, qk_stream s, RadMG<NG> mg, Eos eos, const qk_array4 *cons_t, qk_array4 *eps_t, double dl)
{
	launchRad(lev, s, 0, DIR, "rad_ComputeCellOpticalDepth", [=] __device__(int b, int i, int j, int k, bool valid) {
		if (!valid) {
			return;
		}
		WA4 E(eps_t[b]);
		double eps[NG];
#pragma unroll
		for (int g = 0; g < NG; ++g) {
			eps[g] = 1.0;
		}
		if ((i + j + k) % 2 == 0) {
			RA4 U(cons_t[b]);
			double rho_L, T_L, rho_R, T_R;
			gasDensityAndTemperature<EosCell>(eos, U, i - unit(DIR, 0), j - unit(DIR, 1), k - unit(DIR, 2), rho_L, T_L);
			gasDensityAndTemperature<EosCell>(eos, U, i, j, k, rho_R, T_R);
			MC cell_L(mg), cell_R(mg);
			cell_L.at(rho_L, T_L);
			cell_R.at(rho_R, T_R);
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				const double ratio = mg.bnd[g + 1] / mg.bnd[g];
				const double kappa_L = cell_L.lower(g, rho_L, T_L) * pow(ratio, 0.5 * cell_L.kexp[g]);
				const double kappa_R = cell_R.lower(g, rho_R, T_R) * pow(ratio, 0.5 * cell_R.kexp[g]);
				eps[g] = wavespeedFactor(dl * rho_L * kappa_L, dl * rho_R * kappa_R);
			}
		}
#pragma unroll
		for (int g = 0; g < NG; ++g) {
			E(i, j, k, g) = eps[g];
		}
	});
}

// all directions of a level; eps_t[d]: face arrays (nodal in d, no ghost cells, ngroups components)
template <bool TDEP, class RadT = Rad, class EosT = EosCell>
auto radWavespeedImpl(qk_level *lev, qk_stream s, const qk_rad_traits *rt, const qk_hydro_traits *t, int ndim, const qk_array4 *cons_t, const double dx[3],
		      qk_array4 *const eps_t[3]) -> int
{
	RadT rad(*rt);
	rad.mean_molecular_mass = t->mean_molecular_weight;
	const Eos eos(*t);
	launchWavespeedSingle<0, TDEP, RadT, EosT>(lev, s, rad, eos, cons_t, eps_t[0], dx[0]);
	if (ndim >= 2) {
		launchWavespeedSingle<1, TDEP, RadT, EosT>(lev, s, rad, eos, cons_t, eps_t[1], dx[1]);
	}
	if (ndim == 3) {
		launchWavespeedSingle<2, TDEP, RadT, EosT>(lev, s, rad, eos, cons_t, eps_t[2], dx[2]);
	}
	return radStatus(lev, "ComputeCellOpticalDepth");
}
template <int NG, class MC = RadMG<NG>>
auto radWavespeedMGImpl(qk_level *lev, qk_stream s, const qk_rad_traits *rt, const qk_hydro_traits *t, int ndim, const qk_array4 *cons_t, const double dx[3],
			qk_array4 *const eps_t[3]) -> int
{
	const RadMG<NG> mg(*rt, t->boltzmann_constant);
	const Eos eos(*t);
	launchWavespeedMG<0, NG, MC>(lev, s, mg, eos, cons_t, eps_t[0], dx[0]);
	if (ndim >= 2) {
		launchWavespeedMG<1, NG, MC>(lev, s, mg, eos, cons_t, eps_t[1], dx[1]);
	}
	if (ndim == 3) {
		launchWavespeedMG<2, NG, MC>(lev, s, mg, eos, cons_t, eps_t[2], dx[2]);
	}
	return radStatus(lev, "ComputeCellOpticalDepth");
}

} // namespace qk

#endif // QK_RAD_WAVESPEED_LAUNCH_HPP_
