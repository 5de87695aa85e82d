// qk_rad_mg.hip — the multigroup matter-radiation exchange (RadSystem<problem_t>::AddSourceTermsMultiGroup) behind the C-ABI.
// One thread per cell, the NG per-group vectors in registers (one kernel instantiation per supported NG).  Arithmetic in qk_rad_mg_device.hpp.
#include "qk_internal.hpp"
#include "qk_rad_mg_device.hpp"

using namespace qk;

namespace
{

// the spread counter slots of qk_rad_ops.hip (same layout, same finishing kernel semantics)
constexpr int NSLOT = 1024, SLOT_STRIDE = 32;

__global__ void __launch_bounds__(NSLOT) k_mg_counters_finish(int *slots, int *it, int *fail)
{
	__shared__ int red[NSLOT / 64][5];
	int *slot = slots + static_cast<size_t>(threadIdx.x) * SLOT_STRIDE;
	int v[5];
#pragma unroll
	for (int n = 0; n < 5; ++n) {
		v[n] = slot[n];
		slot[n] = 0;
	}
	for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
		for (int n = 0; n < 5; ++n) {
			const int o = __shfl_xor(v[n], off);
			v[n] = (n == 2) ? max(v[n], o) : v[n] + o;
		}
	}
	if ((threadIdx.x & 63) == 0) {
#pragma unroll
		for (int n = 0; n < 5; ++n) {
			red[threadIdx.x / 64][n] = v[n];
		}
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		int t[5] = {0, 0, 0, 0, 0};
		for (int w = 0; w < NSLOT / 64; ++w) {
#pragma unroll
			for (int n = 0; n < 5; ++n) {
				t[n] = (n == 2) ? max(t[n], red[w][n]) : t[n] + red[w][n];
			}
		}
		it[0] += t[0];
		it[1] += t[1];
		it[2] = max(it[2], t[2]);
		fail[0] += t[3];
		fail[2] += t[4];
	}
}

auto mgCounterSlots(qk_ctx *ctx) -> int *
{
	std::lock_guard<std::mutex> lock(ctx->mtx);
	if (ctx->counter_slots == nullptr) {
		void *p = nullptr;
		const size_t bytes = sizeof(int) * NSLOT * SLOT_STRIDE;
		if (hipMalloc(&p, bytes) != hipSuccess || hipMemsetAsync(p, 0, bytes, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
			return nullptr;
		}
		ctx->owned.push_back(p);
		ctx->counter_slots = static_cast<int *>(p);
	}
	return ctx->counter_slots;
}

template <int NG, bool DUST>
__global__ void __launch_bounds__(256, 1) k_rad_source_mg(const qk_box *boxes, Rad rad, RadMG<NG> mg, Eos eos, qk_array4 *cons_t, const qk_array4 *src_t, double dt, int stage,
							   int *slots, int *d_iteration_counter, int *d_failure_counter)
{
	const int b = blockIdx.y;
	const qk_box bx = boxes[b];
	const int len0 = bx.hi[0] - bx.lo[0] + 1, len1 = bx.hi[1] - bx.lo[1] + 1, len2 = bx.hi[2] - bx.lo[2] + 1;
	const int64_t t_raw = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
	const int64_t n01 = static_cast<int64_t>(len0) * len1;
	const bool valid = t_raw < n01 * len2; // lanes past the end stay alive for the wave reduction of the counters
	const int64_t t = valid ? t_raw : 0;
	const int k = static_cast<int>(t / n01);
	const int rr = static_cast<int>(t - k * n01);
	const int j = rr / len0;
	const int i = rr - j * len0;
	int ntot = 0, nmax = 0, nsolve = 0, fnewton = 0, fouter = 0, ndecoupled = 0, fdust = 0;
	if (valid) {
		WA4 S(cons_t[b]);
		RA4 Q(src_t[b]);
		const int64_t c = S.idx(bx.lo[0] + i, bx.lo[1] + j, bx.lo[2] + k);
		constexpr int NC = RAD0 + NRAD * NG;
		double U[NC], srcval[NG];
#pragma unroll
		for (int n = 0; n < NC; ++n) {
			U[n] = S.p[c + S.ns * n];
		}
#pragma unroll
		for (int g = 0; g < NG; ++g) {
			srcval[g] = Q(bx.lo[0] + i, bx.lo[1] + j, bx.lo[2] + k, g);
		}
		radSourceCellMG<NG, DUST>(rad, mg, eos, U, srcval, dt, stage, ntot, nmax, nsolve, fnewton, fouter, &ndecoupled, &fdust);
#pragma unroll
		for (int n = 1; n < NC; ++n) { // rho (comp 0) is never modified
			S.p[c + S.ns * n] = U[n];
		}
	}
	int wsolve = nsolve, wtot = ntot, wmax = nmax, wfn = fnewton, wfo = fouter;
	for (int off = 32; off > 0; off >>= 1) {
		wsolve += __shfl_xor(wsolve, off);
		wtot += __shfl_xor(wtot, off);
		wmax = max(wmax, __shfl_xor(wmax, off));
		wfn += __shfl_xor(wfn, off);
		wfo += __shfl_xor(wfo, off);
	}
	if ((threadIdx.x & 63) == 0) {
		const unsigned wave = (blockIdx.x + gridDim.x * blockIdx.y) * (blockDim.x / 64) + threadIdx.x / 64;
		int *slot = slots + static_cast<size_t>(wave % NSLOT) * SLOT_STRIDE;
		atomicAdd(&slot[0], wsolve);
		atomicAdd(&slot[1], wtot);
		atomicMax(&slot[2], wmax);
		if (wfn != 0) {
			atomicAdd(&slot[3], wfn);
		}
		if (wfo != 0) {
			atomicAdd(&slot[4], wfo);
		}
	}
	if constexpr (DUST) { // p_iteration_counter[3] (decoupled solves) and p_iteration_failure_counter[1] (negative dust temperature), one atomic per wave
		int wdec = ndecoupled, wfd = fdust;
		for (int off = 32; off > 0; off >>= 1) {
			wdec += __shfl_xor(wdec, off);
			wfd += __shfl_xor(wfd, off);
		}
		if ((threadIdx.x & 63) == 0) {
			if (wdec != 0) {
				atomicAdd(&d_iteration_counter[3], wdec);
			}
			if (wfd != 0) {
				atomicAdd(&d_failure_counter[1], wfd);
			}
		}
	}
}

template <int NG>
auto launchSourceMG(qk_level *lev, qk_stream s, const qk_rad_traits *rt, const qk_hydro_traits *t, qk_array4 *cons_t, const qk_array4 *src_t, double dt, int stage,
		    int *slots, int *d_it, int *d_fail) -> void
{
	Rad rad(*rt);
	rad.mean_molecular_mass = t->mean_molecular_weight;
	const RadMG<NG> mg(*rt, t->boltzmann_constant);
	const Eos eos(*t);
	const CellLaunch L = cellLaunch(lev, 0, -1);
	ProfScope ps(lev->ctx, static_cast<hipStream_t>(s), "rad_AddSourceTermsMultiGroup");
	if (rt->enable_dust_gas_thermal_coupling_model != 0) {
		hipLaunchKernelGGL((k_rad_source_mg<NG, true>), L.grid, L.block, 0, static_cast<hipStream_t>(s), lev->d_boxes, rad, mg, eos, cons_t, src_t, dt, stage, slots,
				   d_it, d_fail);
	} else {
		hipLaunchKernelGGL((k_rad_source_mg<NG, false>), L.grid, L.block, 0, static_cast<hipStream_t>(s), lev->d_boxes, rad, mg, eos, cons_t, src_t, dt, stage, slots,
				   d_it, d_fail);
	}
}

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
	int *slots = mgCounterSlots(lev->ctx);
	QK_REQUIRE(lev->ctx, slots != nullptr, "AddSourceTermsMultiGroup: cannot allocate the counter slots");
	if (lev->nboxes > 0) {
		QK_MG_DISPATCH(rt->ngroups, (launchSourceMG<NG>(lev, s, rt, t, cons_t, src_t, dt, stage, slots, d_iteration_counter, d_failure_counter)))
	}
	hipLaunchKernelGGL(k_mg_counters_finish, dim3(1), dim3(NSLOT), 0, static_cast<hipStream_t>(s), slots, d_iteration_counter, d_failure_counter);
	const hipError_t e = hipGetLastError();
	if (e != hipSuccess) {
		return setError(lev->ctx, QK_ERR_HIP, "AddSourceTermsMultiGroup", hipGetErrorString(e));
	}
	return QK_OK;
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
