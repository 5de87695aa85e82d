// qk_rad_source_launch.hpp — launch side of AddSourceTermsSingleGroup (reference src/radiation/source_terms_single_group.hpp), shared by
//   * qk_rad_ops.hip: the library's instantiations with the CLOSED hook sets of qk_rad_traits (what the Python host and the C-ABI tests use), and
//   * a problem's own translation unit (quokka_amd/host/qk_problem_kernels.hpp): the same kernel instantiated with the problem's compiled
//     ComputePlanckOpacity / ComputeEnergyMeanOpacity / ComputeFluxMeanOpacity / ComputeThermalRadiation* / quokka::EOS / ISM hooks —
//     SURVEY §8(b) option (ii): arbitrary hook code, nothing sampled.
#ifndef QK_RAD_SOURCE_LAUNCH_HPP_
#define QK_RAD_SOURCE_LAUNCH_HPP_

#include "qk_internal.hpp"
#include "qk_rad_device.hpp"

namespace qk
{

// MINW: waves per SIMD the register allocation must leave room for (__launch_bounds__'s second argument on AMD GPUs)
template <int MINW, class F> __global__ void __launch_bounds__(256, MINW) k_rad_cells(const qk_box *boxes, int ndim, int ng, int facedir, F f)
{
	const int b = blockIdx.y;
	const qk_box bx = boxes[b];
	int lo[3], len[3];
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		const int g = (d < ndim) ? ng : 0;
		lo[d] = bx.lo[d] - g;
		len[d] = bx.hi[d] - bx.lo[d] + 1 + 2 * g + ((d == facedir) ? 1 : 0);
	}
	const int64_t t_raw = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
	const int64_t n01 = static_cast<int64_t>(len[0]) * len[1];
	// lanes past the end stay alive with a clamped index and valid = false (wave reductions need every lane)
	const bool valid = t_raw < n01 * len[2];
	const int64_t t = valid ? t_raw : 0;
	int k, r;
	if (n01 * len[2] < (static_cast<int64_t>(1) << 31)) { // uniform: 32-bit division (a 64-bit one is ~100 instructions per thread: 6 % of the Newton-Raphson kernel)
		const unsigned ut = static_cast<unsigned>(t), un01 = static_cast<unsigned>(n01);
		k = static_cast<int>(ut / un01);
		r = static_cast<int>(ut - static_cast<unsigned>(k) * un01);
	} else {
		k = static_cast<int>(t / n01);
		r = static_cast<int>(t - k * n01);
	}
	const int j = r / len[0];
	const int i = r - j * len[0];
	f(b, lo[0] + i, lo[1] + j, lo[2] + k, valid);
}

// Newton-iteration / failure counters: NSLOT slots of one 128-byte line each, folded into the caller's words by one block
constexpr int NSLOT = 1024, SLOT_STRIDE = 32;

#ifndef QK_RAD_SRC_WAVES
#define QK_RAD_SRC_WAVES 2
#endif

static __global__ void __launch_bounds__(NSLOT) k_counters_finish(int *slots, int *it, int *fail)
{
	__shared__ int red[NSLOT / 64][5];
	int *slot = slots + static_cast<size_t>(threadIdx.x) * SLOT_STRIDE;
	int v[5];
#pragma unroll
	for (int n = 0; n < 5; ++n) {
		v[n] = slot[n];
		slot[n] = 0; // ready for the next launch (stream ordered)
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

inline auto counterSlots(qk_ctx *ctx) -> int *
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

// nz: third grid dimension (the photon group of the per-group kernels, read as blockIdx.z inside `f`)
template <int MINW = 1, class F> void launchRad(qk_level *lev, qk_stream s, int ng, int facedir, const char *name, F f, int nz = 1)
{
	if (lev->nboxes == 0) {
		return; // a rank without boxes on this level
	}
	CellLaunch L = cellLaunch(lev, ng, facedir);
	L.grid.z = static_cast<unsigned>(nz);
	ProfScope ps(lev->ctx, static_cast<hipStream_t>(s), name);
	hipLaunchKernelGGL((k_rad_cells<MINW, F>), L.grid, L.block, 0, static_cast<hipStream_t>(s), lev->d_boxes, lev->ndim, ng, facedir, f);
}

inline auto radStatus(qk_level *lev, const char *name) -> int
{
	const hipError_t e = hipGetLastError();
	if (e != hipSuccess) {
		return setError(lev->ctx, QK_ERR_HIP, name, hipGetErrorString(e));
	}
	return QK_OK;
}

// Non-temporal hints on the state the exchange kernel reads and writes once per launch (0: none, 1: stores, 2: stores and loads — the default: the kernel
// moves 3.4 GB per launch beside its arithmetic; same box 1.24 -> 1.21 ms, RadhydroShell 256^3 299.8 -> 305.1 M, profiles/round4/ab12_*)
#ifndef QK_RAD_SRC_NT
#define QK_RAD_SRC_NT 2
#endif
template <class P> QK_DEV void srcStreamStore(P *p, double v)
{
#if QK_RAD_SRC_NT >= 1
	__builtin_nontemporal_store(v, p);
#else
	*p = v;
#endif
}
template <class P> QK_DEV auto srcStreamLoad(P *p) -> double
{
#if QK_RAD_SRC_NT >= 2
	return __builtin_nontemporal_load(p);
#else
	return *p;
#endif
}

// RadT / EosT as in radSourceCell: Rad + EosCell inside the library (closed hook sets), the problem's compiled hooks in a problem's translation unit
template <bool TDEP, bool DUST = false, class RadT = Rad, class EosT = EosCell, int BETA = -1>
static auto radSourceImpl(qk_level *lev, qk_stream s, const qk_rad_traits *rt, const qk_hydro_traits *t, qk_array4 *cons_t, const qk_array4 *src_t, double dt,
			  int stage, int *d_iteration_counter, int *d_failure_counter, qk_array4 *mirror_t = nullptr) -> int
{
	if constexpr (BETA < 0 && !DUST) { // beta_order 1 (and 0): an instantiation of its own, as the reference compiles its kernel per problem
		if (rt->beta_order == 1) {
			return radSourceImpl<TDEP, DUST, RadT, EosT, 1>(lev, s, rt, t, cons_t, src_t, dt, stage, d_iteration_counter, d_failure_counter, mirror_t);
		}
		if (rt->beta_order == 0) {
			return radSourceImpl<TDEP, DUST, RadT, EosT, 0>(lev, s, rt, t, cons_t, src_t, dt, stage, d_iteration_counter, d_failure_counter, mirror_t);
		}
	}
	RadT rad(*rt);
	rad.mean_molecular_mass = t->mean_molecular_weight;
	const Eos eos(*t);
	int *slots = counterSlots(lev->ctx);
	QK_REQUIRE(lev->ctx, slots != nullptr, "AddSourceTermsSingleGroup: cannot allocate the counter slots");
	launchRad<QK_RAD_SRC_WAVES>(lev, s, 0, -1, "rad_AddSourceTerms", [=] __device__(int b, int i, int j, int k, bool valid) {
		int ntot = 0, nmax = 0, nsolve = 0, fnewton = 0, fouter = 0, fdust = 0;
		if (valid) {
			WA4 S(cons_t[b]);
			RA4 Q(src_t[b]);
			const int64_t c = S.idx(i, j, k);
			double U[10];
#pragma unroll
			for (int n = 0; n < 10; ++n) {
				U[n] = srcStreamLoad(&S.p[c + S.ns * n]);
			}
			radSourceCell<TDEP, DUST, RadT, EosT, BETA>(rad, eos, U, Q(i, j, k), dt, stage, ntot, nmax, nsolve, fnewton, fouter, DUST ? &fdust : nullptr);
			if (DUST && fdust != 0) {
				atomicAdd(&d_failure_counter[1], fdust); // (rare: a negative dust temperature; the reference counts it the same way, :172-174)
			}
			// rho (comp 0) is never modified
#pragma unroll
			for (int n = 1; n < 10; ++n) {
				srcStreamStore(&S.p[c + S.ns * n], U[n]);
			}
			if (mirror_t != nullptr) { // the next substep's swapRadiationState, valid cells (QuokkaSimulation.hpp:1783-1788), from registers
				WA4 M(mirror_t[b]);
				const int64_t cm = M.idx(i, j, k);
#pragma unroll
				for (int n = RAD0; n < RAD0 + NRAD; ++n) {
					srcStreamStore(&M.p[cm + M.ns * n], U[n]);
				}
			}
		}
		// counters: wave-level reduction, then one atomic set per wave into one of NSLOT cache-line-sized slots.  (The
		// reference issues 3 atomics per cell on 3 addresses, :344-346.  Even one set per wave on the same three words
		// serialises in L2: measured 8.9 ms per launch against 1.4 ms for the arithmetic of the whole kernel.)
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
	});
	hipLaunchKernelGGL(k_counters_finish, dim3(1), dim3(NSLOT), 0, static_cast<hipStream_t>(s), slots, d_iteration_counter, d_failure_counter);
	return radStatus(lev, "AddSourceTermsSingleGroup");
}

} // namespace qk

#endif // QK_RAD_SOURCE_LAUNCH_HPP_
