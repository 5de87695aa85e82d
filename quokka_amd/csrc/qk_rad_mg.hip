// qk_rad_mg.hip — the multigroup matter-radiation exchange (RadSystem<problem_t>::AddSourceTermsMultiGroup) behind the C-ABI.
// One thread per cell, the NG per-group vectors in registers (one kernel instantiation per supported NG).  Arithmetic in qk_rad_mg_device.hpp.
#include "qk_internal.hpp"
#include "qk_rad_mg_device.hpp"
#include "qk_rad_mg_launch.hpp"
#include "qk_rad_wavespeed_launch.hpp"

using namespace qk;

namespace
{

template <int NG> __global__ void k_mg_planck_fractions(Rad rad, RadMG<NG> mg, int n, const double *T, double *frac_out, double *E_out)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) {
		return;
	}
	double f[NG], E[NG];
	planckEnergyFractions<NG>(mg, T[i], f);
	thermalRadiationMG<NG>(rad, f, T[i], E);
	for (int g = 0; g < NG; ++g) {
		frac_out[i * NG + g] = f[g];
		E_out[i * NG + g] = E[g];
	}
}

template <int NG> void launchPlanckFractions(const qk_rad_traits *rt, double kB, int n, const double *dT, double *dF, double *dE)
{
	const Rad rad(*rt);
	const RadMG<NG> mg(*rt, kB);
	hipLaunchKernelGGL((k_mg_planck_fractions<NG>), dim3((n + 63) / 64), dim3(64), 0, nullptr, rad, mg, n, dT, dF, dE);
}

auto checkMG(qk_ctx *ctx, const qk_rad_traits *rt) -> int
{
	if (rt == nullptr) {
		return setError(ctx, QK_ERR_INVALID, "rad traits is NULL");
	}
	const int ng = rt->ngroups;
	if (!(ng == 2 || ng == 3 || ng == 4 || ng == 5 || ng == 6 || ng == 8)) {
		return setError(ctx, QK_ERR_UNSUPPORTED, "multigroup source term: ngroups must be one of 2, 3, 4, 5, 6, 8");
	}
	if (rt->mg_opacity_model < MG_PIECEWISE_CONSTANT || rt->mg_opacity_model > MG_PPL_FULL_SPECTRUM) {
		return setError(ctx, QK_ERR_UNSUPPORTED,
				"mg_opacity_model must be 1 (piecewise_constant_opacity), 2 (PPL_opacity_fixed_slope_spectrum) or 3 (PPL_opacity_full_spectrum)");
	}
	if (rt->opacity_model == QK_HOOK_COMPILED) {
		return setError(ctx, QK_ERR_UNSUPPORTED,
				"multigroup: a compiled DefineOpacityExponentsAndLowerValues hook exists only in the problem's own translation unit (qk_problem_kernels.hpp)");
	}
	if (rt->beta_order != 0 && rt->beta_order != 1) {
		return setError(ctx, QK_ERR_UNSUPPORTED, "multigroup source term: beta_order must be 0 or 1 (source_terms_multi_group.hpp:526)");
	}
	if (rt->thermal_model != 0 && !(rt->thermal_model == 1 && rt->enable_dust_gas_thermal_coupling_model != 0)) {
		return setError(ctx, QK_ERR_UNSUPPORTED, "multigroup source term: thermal_model must be 0, or 1 together with the dust model");
	}
	if (rt->enable_dust_gas_thermal_coupling_model != 0 && !(rt->dust_gas_interaction_coeff > 0.0 && rt->gas_dust_coupling_threshold >= 0.0)) {
		return setError(ctx, QK_ERR_INVALID, "multigroup dust model: dust_gas_interaction_coeff must be positive, gas_dust_coupling_threshold not negative");
	}
	if (rt->enable_dust_gas_thermal_coupling_model == 0) {
		bool hooks = (rt->cr_heating_rate != 0.0) || (rt->enable_photoelectric_heating != 0);
		for (int g = 0; g < ng; ++g) {
			hooks = hooks || (rt->cooling_linear_coeff[g] != 0.0);
		}
		if (hooks) {
			return setError(ctx, QK_ERR_UNSUPPORTED,
					"multigroup source term: the line-cooling / cosmic-ray / photoelectric heating hooks are carried together with the dust model only");
		}
	}
	if (!(rt->energy_unit > 0.0)) {
		return setError(ctx, QK_ERR_INVALID, "multigroup: energy_unit must be positive");
	}
	for (int g = 0; g < ng; ++g) {
		if (!(rt->rad_boundaries[g] > 0.0 && rt->rad_boundaries[g + 1] > rt->rad_boundaries[g])) {
			return setError(ctx, QK_ERR_INVALID, "multigroup: rad_boundaries must be positive and increasing");
		}
	}
	if (rt->mg_kappa_T_exponent != 0.0 && !(rt->mg_kappa_T_ref > 0.0)) {
		return setError(ctx, QK_ERR_INVALID, "multigroup: mg_kappa_T_ref must be positive when mg_kappa_T_exponent is not 0");
	}
	return QK_OK;
}

} // namespace

#define QK_MG_DISPATCH(NGV, CALL)                                                                                                                    \
	switch (NGV) {                                                                                                                               \
	case 2: {                                                                                                                                    \
		constexpr int NG = 2;                                                                                                                \
		CALL;                                                                                                                                \
	} break;                                                                                                                                     \
	case 3: {                                                                                                                                    \
		constexpr int NG = 3;                                                                                                                \
		CALL;                                                                                                                                \
	} break;                                                                                                                                     \
	case 4: {                                                                                                                                    \
		constexpr int NG = 4;                                                                                                                \
		CALL;                                                                                                                                \
	} break;                                                                                                                                     \
	case 5: {                                                                                                                                    \
		constexpr int NG = 5;                                                                                                                \
		CALL;                                                                                                                                \
	} break;                                                                                                                                     \
	case 6: {                                                                                                                                    \
		constexpr int NG = 6;                                                                                                                \
		CALL;                                                                                                                                \
	} break;                                                                                                                                     \
	default: {                                                                                                                                   \
		constexpr int NG = 8;                                                                                                                \
		CALL;                                                                                                                                \
	} break;                                                                                                                                     \
	}

extern "C" {

int qk_rad_AddSourceTermsMultiGroup(qk_level *lev, qk_stream s, const qk_rad_traits *rt, const qk_hydro_traits *t, qk_array4 *cons_t, const qk_array4 *src_t,
				    double dt, int stage, int *d_iteration_counter, int *d_failure_counter)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkMG(lev->ctx, rt); rc != QK_OK) {
		return rc;
	}
	if (int rc = checkTraits(lev->ctx, t); rc != QK_OK) {
		return rc;
	}
	if (int rc = needsLibraryEos(lev->ctx, t, "AddSourceTermsMultiGroup"); rc != QK_OK) {
		return rc;
	}
	QK_REQUIRE(lev->ctx, cons_t && src_t && d_iteration_counter && d_failure_counter, "AddSourceTermsMultiGroup: NULL");
	QK_REQUIRE(lev->ctx, stage == 1 || stage == 2, "AddSourceTermsMultiGroup: stage must be 1 or 2");
	QK_REQUIRE(lev->ctx, t->nscalars == 0 && t->nmscalars == 0, "AddSourceTermsMultiGroup: radFirstIndex is 6 (no passive scalars beside radiation)");
	int rc_launch = QK_OK;
	QK_MG_DISPATCH(rt->ngroups, (rc_launch = radSourceMGImpl<NG>(lev, s, rt, t, cons_t, src_t, dt, stage, d_iteration_counter, d_failure_counter)))
	return rc_launch;
}

// RadSystem::ComputeCellOpticalDepth + the S_corr of ComputeFluxes (radiation_system.hpp:803-871, :1098-1109) for every face and photon group
int qk_rad_ComputeWavespeedCorrection(qk_level *lev, qk_stream s, const qk_rad_traits *rt, const qk_hydro_traits *t, int ndim, const qk_array4 *cons_t,
				      const double dx[3], qk_array4 *const eps[3])
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(lev->ctx, rt != nullptr && cons_t != nullptr && dx != nullptr && eps != nullptr, "ComputeWavespeedCorrection: NULL argument");
	QK_REQUIRE(lev->ctx, ndim == lev->ndim && eps[0] != nullptr && (ndim < 2 || eps[1] != nullptr) && (ndim < 3 || eps[2] != nullptr),
		   "ComputeWavespeedCorrection: one face array per direction of the level");
	if (int rc = checkTraits(lev->ctx, t); rc != QK_OK) {
		return rc;
	}
	if (int rc = needsLibraryEos(lev->ctx, t, "ComputeWavespeedCorrection"); rc != QK_OK) {
		return rc;
	}
	if (rt->opacity_model == QK_HOOK_COMPILED) {
		return setError(lev->ctx, QK_ERR_UNSUPPORTED, "ComputeWavespeedCorrection",
				"the opacity hooks of this problem are compiled device code: instantiate the kernel in the problem's translation unit "
				"(quokka_amd/host/qk_problem_kernels.hpp)");
	}
	if (rt->ngroups <= 1) {
		QK_REQUIRE(lev->ctx, rt->opacity_model >= 0 && rt->opacity_model <= 2, "ComputeWavespeedCorrection: opacity_model must be 0, 1 or 2");
		return (rt->opacity_model == 2) ? radWavespeedImpl<true>(lev, s, rt, t, ndim, cons_t, dx, eps) : radWavespeedImpl<false>(lev, s, rt, t, ndim, cons_t, dx, eps);
	}
	if (int rc = checkMG(lev->ctx, rt); rc != QK_OK) {
		return rc;
	}
	int rc_launch = QK_OK;
	QK_MG_DISPATCH(rt->ngroups, (rc_launch = radWavespeedMGImpl<NG>(lev, s, rt, t, ndim, cons_t, dx, eps)))
	return rc_launch;
}

int qk_rad_mg_planck_fractions(qk_ctx *ctx, const qk_rad_traits *rt, double kB, int n, const double *T, double *fractions, double *Erad_g)
{
	if (ctx == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkMG(ctx, rt); rc != QK_OK) {
		return rc;
	}
	QK_REQUIRE(ctx, n > 0 && T && fractions && Erad_g, "qk_rad_mg_planck_fractions: NULL or empty");
	const int ng = rt->ngroups;
	double *dT = nullptr, *dF = nullptr, *dE = nullptr;
	const size_t nb = sizeof(double) * static_cast<size_t>(n), nbg = nb * static_cast<size_t>(ng);
	hipError_t e = hipMalloc(reinterpret_cast<void **>(&dT), nb);
	e = (e == hipSuccess) ? hipMalloc(reinterpret_cast<void **>(&dF), nbg) : e;
	e = (e == hipSuccess) ? hipMalloc(reinterpret_cast<void **>(&dE), nbg) : e;
	e = (e == hipSuccess) ? hipMemcpy(dT, T, nb, hipMemcpyHostToDevice) : e;
	if (e == hipSuccess) {
		QK_MG_DISPATCH(ng, (launchPlanckFractions<NG>(rt, kB, n, dT, dF, dE)))
		e = hipGetLastError();
	}
	e = (e == hipSuccess) ? hipMemcpy(fractions, dF, nbg, hipMemcpyDeviceToHost) : e;
	e = (e == hipSuccess) ? hipMemcpy(Erad_g, dE, nbg, hipMemcpyDeviceToHost) : e;
	(void)hipFree(dT);
	(void)hipFree(dF);
	(void)hipFree(dE);
	if (e != hipSuccess) {
		return setError(ctx, QK_ERR_HIP, "qk_rad_mg_planck_fractions", hipGetErrorString(e));
	}
	return QK_OK;
}

} // extern "C"
