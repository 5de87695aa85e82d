// qk_rad_mg_launch.hpp — launch side of AddSourceTermsMultiGroup (reference src/radiation/source_terms_multi_group.hpp:522-813), shared by
//   * qk_rad_mg.hip: the library's instantiations with the CLOSED set of DefineOpacityExponentsAndLowerValues (exponents independent of the
//     state, lower values k_g rho^a T^b) that qk_rad_traits carries, and
//   * a problem's own translation unit (quokka_amd/host/qk_problem_kernels.hpp): the same kernel with a cell type whose at(rho, T) CALLS the
//     problem's compiled DefineOpacityExponentsAndLowerValues — any dependence on rho and T, exponents included (RadhydroPulseMGint).
#ifndef QK_RAD_MG_LAUNCH_HPP_
#define QK_RAD_MG_LAUNCH_HPP_

#include <type_traits>

#include "qk_rad_mg_device.hpp"
#include "qk_rad_source_launch.hpp"

namespace qk
{

// MC: what radSourceCellMG sees as the multigroup description of ONE cell.  RadMG<NG> itself (the kernel argument, uniform) inside the library;
// a type derived from it and constructible from it (thread-private, refreshed by at()) for a problem's compiled hook.
template <int NG, bool DUST, class MC>
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
		if constexpr (std::is_same_v<MC, RadMG<NG>>) {
			radSourceCellMG<NG, DUST>(rad, mg, eos, U, srcval, dt, stage, ntot, nmax, nsolve, fnewton, fouter, &ndecoupled, &fdust);
		} else {
			MC cell(mg);
			radSourceCellMG<NG, DUST>(rad, cell, eos, U, srcval, dt, stage, ntot, nmax, nsolve, fnewton, fouter, &ndecoupled, &fdust);
		}
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

// the launch + the folding of the counter slots; the caller has validated the traits
template <int NG, class MC = RadMG<NG>>
static auto radSourceMGImpl(qk_level *lev, qk_stream s, const qk_rad_traits *rt, const qk_hydro_traits *t, qk_array4 *cons_t, const qk_array4 *src_t, double dt, int stage,
			    int *d_iteration_counter, int *d_failure_counter) -> int
{
	int *slots = counterSlots(lev->ctx);
	QK_REQUIRE(lev->ctx, slots != nullptr, "AddSourceTermsMultiGroup: cannot allocate the counter slots");
	if (lev->nboxes > 0) {
		Rad rad(*rt);
		rad.mean_molecular_mass = t->mean_molecular_weight;
		const RadMG<NG> mg(*rt, t->boltzmann_constant);
		const Eos eos(*t);
		const CellLaunch L = cellLaunch(lev, 0, -1);
		ProfScope ps(lev->ctx, static_cast<hipStream_t>(s), "rad_AddSourceTermsMultiGroup");
		if (rt->enable_dust_gas_thermal_coupling_model != 0) {
			hipLaunchKernelGGL((k_rad_source_mg<NG, true, MC>), L.grid, L.block, 0, static_cast<hipStream_t>(s), lev->d_boxes, rad, mg, eos, cons_t, src_t, dt, stage,
					   slots, d_iteration_counter, d_failure_counter);
		} else {
			hipLaunchKernelGGL((k_rad_source_mg<NG, false, MC>), L.grid, L.block, 0, static_cast<hipStream_t>(s), lev->d_boxes, rad, mg, eos, cons_t, src_t, dt, stage,
					   slots, d_iteration_counter, d_failure_counter);
		}
	}
	hipLaunchKernelGGL(k_counters_finish, dim3(1), dim3(NSLOT), 0, static_cast<hipStream_t>(s), slots, d_iteration_counter, d_failure_counter);
	return radStatus(lev, "AddSourceTermsMultiGroup");
}

} // namespace qk

#endif // QK_RAD_MG_LAUNCH_HPP_
