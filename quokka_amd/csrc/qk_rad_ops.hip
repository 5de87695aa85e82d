// qk_rad_ops.hip — two-moment radiation operators (RadSystem<problem_t>) behind the C-ABI: transport for 1 .. QK_MAX_GROUPS photon groups
// (the groups are independent there: the group index is a grid dimension), the single-group source term.  The multigroup source term is in
// qk_rad_mg.hip.  One launch covers all local boxes.  Arithmetic in qk_rad_device.hpp.
#include "qk_internal.hpp"
#include "qk_rad_device.hpp"
#include "qk_rad_source_launch.hpp"

using namespace qk;

namespace
{

inline auto checkRad(qk_ctx *ctx, const qk_rad_traits *rt) -> int
{
	if (rt == nullptr) {
		return setError(ctx, QK_ERR_INVALID, "rad traits is NULL");
	}
	if (rt->opacity_model == 2 && !(rt->opacity_T_ref > 0.0 && rt->opacity_pow_floor >= 0.0 && rt->opacity_T_exponent == rt->opacity_T_exponent)) {
		return setError(ctx, QK_ERR_INVALID, "opacity_model 2 needs opacity_T_ref > 0, a finite opacity_T_exponent and opacity_pow_floor >= 0");
	}
	if (rt->opacity_model == QK_HOOK_COMPILED) {
		// the problem's opacities are compiled device code: the transport operators never evaluate them; the source term refuses below
	} else if (rt->opacity_model < 0 || rt->opacity_model > 2) {
		return setError(ctx, QK_ERR_UNSUPPORTED, "opacity_model must be 0 (constant kappa), 1 (constant rho * kappa), 2 (temperature power law) or QK_HOOK_COMPILED");
	}
	if (rt->eddington_model < 0 || rt->eddington_model > 1) {
		return setError(ctx, QK_ERR_UNSUPPORTED,
				"opacity_model must be 0 (constant kappa), 1 (constant rho * kappa) or 2 (temperature power law), eddington_model 0 (Levermore) or 1 (1/3)");
	}
	if (rt->beta_order < 0 || rt->beta_order > 3) {
		return setError(ctx, QK_ERR_INVALID, "beta_order must be 0..3");
	}
	if (rt->ngroups < 0 || rt->ngroups > QK_MAX_GROUPS) {
		return setError(ctx, QK_ERR_UNSUPPORTED, "ngroups must be 1 .. QK_MAX_GROUPS (8)");
	}
	return QK_OK;
}

template <int DIR>
void launchRadComputeFluxes(qk_level *lev, qk_stream s, Rad rad, qk_array4 *flux_t, const qk_array4 *left_t, const qk_array4 *right_t, const qk_array4 *cons_t,
			    const qk_array4 *eps_t)
{
	launchRad(lev, s, 0, DIR, "rad_ComputeFluxes", [=] __device__(int b, int i, int j, int k, bool valid) {
		if (!valid) {
			return;
		}
		RA4 L(left_t[b]);
		RA4 R(right_t[b]);
		RA4 U(cons_t[b]);
		WA4 F(flux_t[b]);
		const int im = i - unit(DIR, 0), jm = j - unit(DIR, 1), km = k - unit(DIR, 2);
		const int pg = NRAD * static_cast<int>(blockIdx.z); // component offset of this block's photon group
		double pL[NRAD], pR[NRAD], cL[NRAD], cR[NRAD], Fo[NRAD];
#pragma unroll
		for (int n = 0; n < NRAD; ++n) {
			pL[n] = L(i, j, k, pg + n);
			pR[n] = R(i, j, k, pg + n);
			cL[n] = U(im, jm, km, RAD0 + pg + n);
			cR[n] = U(i, j, k, RAD0 + pg + n);
		}
		radFaceFlux<DIR>(rad, pL, pR, cL, cR, Fo, faceEpsilon(eps_t, b, i, j, k, static_cast<int>(blockIdx.z)));
#pragma unroll
		for (int n = 0; n < NRAD; ++n) {
			F(i, j, k, pg + n) = Fo[n];
		}
	}, rad.ngroups);
}

// cons -> prim of one cell (radiation_system.hpp:603-610)
QK_DEV void radPrim(Rad const &r, const double c[NRAD], double p[NRAD])
{
	p[0] = c[0];
	const Recip RcE = recipOf(r.c * c[0]); // three quotients, one refined reciprocal (qk_device.hpp: same bits as three `/`)
	p[1] = divBy(c[1], RcE);
	p[2] = divBy(c[2], RcE);
	p[3] = divBy(c[3], RcE);
}

template <int DIR, int ORDER> void launchRadFusedFlux(qk_level *lev, qk_stream s, Rad rad, const qk_array4 *cons_t, qk_array4 *flux_t, const qk_array4 *eps_t)
{
	launchRad(lev, s, 0, DIR, "rad_fluxFunction", [=] __device__(int b, int i, int j, int k, bool valid) {
		if (!valid) {
			return;
		}
		RA4 U(cons_t[b]);
		WA4 F(flux_t[b]);
		const int dx = unit(DIR, 0), dy = unit(DIR, 1), dz = unit(DIR, 2);
		const int pg = NRAD * static_cast<int>(blockIdx.z); // component offset of this block's photon group
		// cells i-3 .. i+2 along DIR
		double c[6][NRAD], p[6][NRAD];
		constexpr int M0 = (ORDER == 3) ? 0 : (ORDER == 2) ? 1 : 2;
		constexpr int M1 = (ORDER == 3) ? 5 : (ORDER == 2) ? 4 : 3;
#pragma unroll
		for (int m = M0; m <= M1; ++m) {
			const int64_t o = U.idx(i + (m - 3) * dx, j + (m - 3) * dy, k + (m - 3) * dz);
#pragma unroll
			for (int n = 0; n < NRAD; ++n) {
				c[m][n] = U.p[o + U.ns * (RAD0 + pg + n)];
			}
			radPrim(rad, c[m], p[m]);
		}
		double pL[NRAD], pR[NRAD], Fo[NRAD];
#pragma unroll
		for (int n = 0; n < NRAD; ++n) {
			if (ORDER == 3) {
				double am, ap;
				ppmEdges(p[0][n], p[1][n], p[2][n], p[3][n], p[4][n], am, ap); // cell i-1: right edge -> leftState(i)
				pL[n] = ap;
				ppmEdges(p[1][n], p[2][n], p[3][n], p[4][n], p[5][n], am, ap); // cell i: left edge -> rightState(i)
				pR[n] = am;
			} else if (ORDER == 2) {
				// hyperbolic_system.hpp:243-246 with the MC limiter (QuokkaSimulation.hpp:1948)
				const double lslope = MC(p[3][n] - p[2][n], p[2][n] - p[1][n]);
				const double rslope = MC(p[4][n] - p[3][n], p[3][n] - p[2][n]);
				pL[n] = p[2][n] + 0.25 * lslope;
				pR[n] = p[3][n] - 0.25 * rslope;
			} else {
				pL[n] = p[2][n];
				pR[n] = p[3][n];
			}
		}
		radFaceFlux<DIR>(rad, pL, pR, c[2], c[3], Fo, faceEpsilon(eps_t, b, i, j, k, static_cast<int>(blockIdx.z)));
		const int64_t o = F.idx(i, j, k);
#pragma unroll
		for (int n = 0; n < NRAD; ++n) {
			F.p[o + F.ns * (pg + n)] = Fo[n];
		}
	}, rad.ngroups);
}

// Y / Z sweeps: one thread marches a strip of STRIP faces along DIR with a rolling window of cells, so that every cell's primitives and
// edge states are evaluated once per strip instead of once per face that touches it (the per-face kernel above is FP64-issue bound:
// 4-6 cons->prim conversions and two reconstructions per face).  Same functions, same operands: identical fluxes.
template <int DIR, int ORDER, int STRIP>
__global__ void __launch_bounds__(256) k_rad_flux_march(const qk_box *boxes, Rad rad, const qk_array4 *cons_t, qk_array4 *flux_t, int notb, const qk_array4 *eps_t)
{
	static_assert(DIR == 1 || DIR == 2, "marching flux kernel: strided directions only");
	constexpr int OT = 3 - DIR;
	const int b = static_cast<int>(blockIdx.z) / rad.ngroups;
	const int pg = NRAD * (static_cast<int>(blockIdx.z) % rad.ngroups); // component offset of this block's photon group
	const qk_box bx = boxes[b];
	const int i = bx.lo[0] + static_cast<int>(blockIdx.x) * 64 + static_cast<int>(threadIdx.x);
	const int oblk = static_cast<int>(blockIdx.y) % notb, strip = static_cast<int>(blockIdx.y) / notb;
	const int ot = bx.lo[OT] + oblk * 4 + static_cast<int>(threadIdx.y);
	const int f0 = bx.lo[DIR] + strip * STRIP;
	if (i > bx.hi[0] || ot > bx.hi[OT] || f0 > bx.hi[DIR] + 1) {
		return;
	}
	RA4 U(cons_t[b]);
	WA4 F(flux_t[b]);
	constexpr int M0 = (ORDER == 3) ? 0 : (ORDER == 2) ? 1 : 2;
	constexpr int M1 = (ORDER == 3) ? 5 : (ORDER == 2) ? 4 : 3;
	double c[6][NRAD], p[6][NRAD]; // cells face-3 .. face+2 along DIR (only M0..M1 are live)
	double carry[NRAD];	       // PPM: right edge of cell face-1; PLM: its limited slope
	int pos[3];
	pos[0] = i;
	pos[OT] = ot;
	auto load = [&](int m, int face) {
		pos[DIR] = face + (m - 3);
		const int64_t o = U.idx(pos[0], pos[1], pos[2]);
#pragma unroll
		for (int n = 0; n < NRAD; ++n) {
			c[m][n] = U.p[o + U.ns * (RAD0 + pg + n)];
		}
		radPrim(rad, c[m], p[m]);
	};
	const int nface = min(STRIP, bx.hi[DIR] + 1 - f0 + 1);
	for (int sI = 0; sI < nface; ++sI) {
		const int face = f0 + sI;
		if (sI == 0) {
#pragma unroll
			for (int m = M0; m <= M1; ++m) {
				load(m, face);
			}
#pragma unroll
			for (int n = 0; n < NRAD; ++n) {
				if (ORDER == 3) {
					double am, ap;
					ppmEdges(p[0][n], p[1][n], p[2][n], p[3][n], p[4][n], am, ap);
					carry[n] = ap;
				} else if (ORDER == 2) {
					carry[n] = MC(p[3][n] - p[2][n], p[2][n] - p[1][n]);
				}
			}
		} else {
#pragma unroll
			for (int m = M0; m < M1; ++m) {
#pragma unroll
				for (int n = 0; n < NRAD; ++n) {
					c[m][n] = c[m + 1][n];
					p[m][n] = p[m + 1][n];
				}
			}
			load(M1, face);
		}
		double pL[NRAD], pR[NRAD], Fo[NRAD];
#pragma unroll
		for (int n = 0; n < NRAD; ++n) {
			if (ORDER == 3) {
				double am, ap;
				ppmEdges(p[1][n], p[2][n], p[3][n], p[4][n], p[5][n], am, ap); // cell `face`
				pL[n] = carry[n];
				pR[n] = am;
				carry[n] = ap;
			} else if (ORDER == 2) {
				const double rslope = MC(p[4][n] - p[3][n], p[3][n] - p[2][n]); // slope of cell `face`
				pL[n] = p[2][n] + 0.25 * carry[n];
				pR[n] = p[3][n] - 0.25 * rslope;
				carry[n] = rslope;
			} else {
				pL[n] = p[2][n];
				pR[n] = p[3][n];
			}
		}
		pos[DIR] = face;
		radFaceFlux<DIR>(rad, pL, pR, c[2], c[3], Fo, faceEpsilon(eps_t, b, pos[0], pos[1], pos[2], pg / NRAD));
		const int64_t o = F.idx(pos[0], pos[1], pos[2]);
#pragma unroll
		for (int n = 0; n < NRAD; ++n) {
			F.p[o + F.ns * (pg + n)] = Fo[n];
		}
	}
}

// X sweep: one thread per cell of a flat slab (rows j = lo.y .. hi.y of plane k are contiguous, ghost cells included), primitives and
// edge states exchanged through LDS, each thread evaluates the flux at the left face of its cell: one cons->prim and one reconstruction per
// cell instead of 4-6 and 2 per face.  250 faces per 256-thread workgroup (3 halo cells on each side).  Same functions, same operands.
constexpr int RXB = 256, RXOUT = 250;

template <int ORDER> __global__ void __launch_bounds__(RXB) k_rad_flux_x(const qk_box *boxes, Rad rad, const qk_array4 *cons_t, qk_array4 *flux_t, const qk_array4 *eps_t)
{
	__shared__ double s_p[NRAD][RXB]; // primitives
	__shared__ double s_e[NRAD][RXB]; // PPM: right edge a_plus of the cell; PLM: its limited slope
	__shared__ double s_c[NRAD][RXB]; // conserved radiation state (the first-order fallback of the HLL flux needs it)
	const int b = static_cast<int>(blockIdx.z) / rad.ngroups;
	const int pg = NRAD * (static_cast<int>(blockIdx.z) % rad.ngroups); // component offset of this block's photon group
	const qk_box bx = boxes[b];
	const int k = bx.lo[2] + static_cast<int>(blockIdx.y);
	if (k > bx.hi[2]) {
		return; // uniform for the workgroup
	}
	RA4 U(cons_t[b]);
	const int t = threadIdx.x;
	const int64_t rowlen = cons_t[b].end[0] - cons_t[b].begin[0]; // x extent of the array, ghost cells included (the row PITCH, U.js, may be larger)
	const int64_t slablen = rowlen * (bx.hi[1] - bx.lo[1] + 1);
	const int64_t f = static_cast<int64_t>(blockIdx.x) * RXOUT + t - 3; // flat position inside the slab of plane k
	const bool inside = (f >= 0) && (f < slablen);
	const int64_t fc = inside ? f : 0;
	const int jj = static_cast<int>(fc / rowlen);
	const int i = U.bx + static_cast<int>(fc - jj * rowlen);
	const int j = bx.lo[1] + jj;
	const int64_t o = U.idx(i, j, k);
	double c0[NRAD], p0[NRAD];
#pragma unroll
	for (int n = 0; n < NRAD; ++n) {
		c0[n] = U.p[o + U.ns * (RAD0 + pg + n)];
	}
	radPrim(rad, c0, p0);
#pragma unroll
	for (int n = 0; n < NRAD; ++n) {
		s_c[n][t] = c0[n];
		s_p[n][t] = p0[n];
	}
	__syncthreads();
	const int tm2 = max(t - 2, 0), tm1 = max(t - 1, 0), tp1 = min(t + 1, RXB - 1), tp2 = min(t + 2, RXB - 1);
	double edgeL[NRAD]; // left edge state of my cell = rightState of my left face
#pragma unroll
	for (int n = 0; n < NRAD; ++n) {
		if (ORDER == 3) {
			double am, ap;
			ppmEdges(s_p[n][tm2], s_p[n][tm1], p0[n], s_p[n][tp1], s_p[n][tp2], am, ap);
			edgeL[n] = am;
			s_e[n][t] = ap;
		} else if (ORDER == 2) {
			const double slope = MC(s_p[n][tp1] - p0[n], p0[n] - s_p[n][tm1]); // hyperbolic_system.hpp:243-246, MC limiter
			edgeL[n] = p0[n] - 0.25 * slope;
			s_e[n][t] = slope;
		} else {
			edgeL[n] = p0[n];
		}
	}
	__syncthreads();
	const bool isFace = inside && (i >= bx.lo[0]) && (i <= bx.hi[0] + 1) && (t >= 3) && (t <= RXB - 3);
	if (!isFace) {
		return;
	}
	double pL[NRAD], cL[NRAD], Fo[NRAD];
#pragma unroll
	for (int n = 0; n < NRAD; ++n) {
		cL[n] = s_c[n][tm1];
		if (ORDER == 3) {
			pL[n] = s_e[n][tm1];
		} else if (ORDER == 2) {
			pL[n] = s_p[n][tm1] + 0.25 * s_e[n][tm1];
		} else {
			pL[n] = s_p[n][tm1];
		}
	}
	radFaceFlux<0>(rad, pL, edgeL, cL, c0, Fo, faceEpsilon(eps_t, b, i, j, k, pg / NRAD));
	WA4 F(flux_t[b]);
	const int64_t of = F.idx(i, j, k);
#pragma unroll
	for (int n = 0; n < NRAD; ++n) {
		F.p[of + F.ns * (pg + n)] = Fo[n];
	}
}

template <int ORDER> void launchRadXFlux(qk_level *lev, qk_stream s, Rad rad, const qk_array4 *cons_t, qk_array4 *flux_t, int nghost, const qk_array4 *eps_t)
{
	if (lev->nboxes == 0) {
		return;
	}
	const int64_t slab = static_cast<int64_t>(lev->maxlen[0] + 2 * nghost) * lev->maxlen[1];
	const dim3 grid(static_cast<unsigned>((slab + RXOUT - 1) / RXOUT), static_cast<unsigned>(lev->maxlen[2]),
			static_cast<unsigned>(lev->nboxes * rad.ngroups));
	ProfScope ps(lev->ctx, static_cast<hipStream_t>(s), "rad_fluxFunction");
	hipLaunchKernelGGL((k_rad_flux_x<ORDER>), grid, dim3(RXB), 0, static_cast<hipStream_t>(s), lev->d_boxes, rad, cons_t, flux_t, eps_t);
}

template <int DIR, int ORDER> void launchRadMarchFlux(qk_level *lev, qk_stream s, Rad rad, const qk_array4 *cons_t, qk_array4 *flux_t, const qk_array4 *eps_t)
{
	if (lev->nboxes == 0) {
		return;
	}
	constexpr int STRIP = 8;
	constexpr int OT = 3 - DIR;
	const int notb = (lev->maxlen[OT] + 3) / 4;
	const int nstrips = (lev->maxlen[DIR] + 1 + STRIP - 1) / STRIP;
	const dim3 grid((lev->maxlen[0] + 63) / 64, notb * nstrips, lev->nboxes * rad.ngroups);
	ProfScope ps(lev->ctx, static_cast<hipStream_t>(s), "rad_fluxFunction");
	hipLaunchKernelGGL((k_rad_flux_march<DIR, ORDER, STRIP>), grid, dim3(64, 4), 0, static_cast<hipStream_t>(s), lev->d_boxes, rad, cons_t, flux_t, notb, eps_t);
}


// ------------------------------------------------------------------------------------------------------------------------------------------
// One transport stage without the face-flux round trip (qk_rad_stage_fused): the flux kernels above with the flux divergence taken where the
// fluxes are produced, as the hydro sweeps do.  X: acc = (dt/dx)(F_i - F_i+1); Y: acc += (dt/dy)(...); Z: d = acc + (dt/dz)(...), then the
// update of PredictStep (stage 1) or AddFluxesRK2 (stage 2) and the validity repair — the same operands in the same association as the
// separate kernels, so the state is the same in every bit.  The face fluxes are stored only on request (flux registers).
struct RadSweep {
	const qk_array4 *U_in; // state whose fluxes are taken (ghost cells filled)
	const qk_array4 *U0;   // state at the start of the substep (stage 1: the same arrays as U_in)
	qk_array4 *U_new;      // may alias U_in: the Z sweep marches every column in one thread and touches no other column
	qk_array4 *acc;	       // NRAD components per cell, no ghost cells needed
	qk_array4 *flux[3];    // each NULL or the face-centred array that receives the fluxes
	double dtdx[3];
	const qk_array4 *eps[3]; // each NULL or the wavespeed-correction factors of the faces of that direction (qk_rad_ComputeWavespeedCorrection)
	int pg; // component offset of the photon group these launches advance (NRAD * group): the groups are transported independently of each other
};

// Non-temporal hints on the flux-divergence accumulator of the fused transport stage where they pay: the X sweep's stores and the Z sweep's loads
// (same box: Z -3.5 %, X -1.5 %).  The Y sweep, which reads AND rewrites the accumulator, is 5 % slower with them and keeps plain accesses.
// Level 2 (default) adds the Z sweep's stores of the new radiation state and its loads of the substep's starting state (Z another -2 %).  QK_RAD_NT=0: none.
#ifndef QK_RAD_NT
#define QK_RAD_NT 2
#endif
template <class P> QK_DEV void radStreamStore(P *p, double v)
{
#if QK_RAD_NT
	__builtin_nontemporal_store(v, p);
#else
	*p = v;
#endif
}
template <class P> QK_DEV auto radStreamLoad(P *p) -> double
{
#if QK_RAD_NT
	return __builtin_nontemporal_load(p);
#else
	return *p;
#endif
}

template <class P> QK_DEV void radStreamStore2(P *p, double v)
{
#if QK_RAD_NT >= 2
	__builtin_nontemporal_store(v, p);
#else
	*p = v;
#endif
}
template <class P> QK_DEV auto radStreamLoad2(P *p) -> double
{
#if QK_RAD_NT >= 2
	return __builtin_nontemporal_load(p);
#else
	return *p;
#endif
}

constexpr int RXCELLS = RXB - 7; // cells updated per workgroup of the X sweep: threads 3 .. RXB-4 (their right neighbour holds the other face)

template <int ORDER, bool STORE> __global__ void __launch_bounds__(RXB) k_rad_sweep_x(const qk_box *boxes, Rad rad, RadSweep a)
{
	__shared__ double s_p[NRAD][RXB];
	__shared__ double s_e[NRAD][RXB];
	__shared__ double s_c[NRAD][RXB];
	__shared__ double s_f[NRAD][RXB]; // flux at the left face of the thread's cell
	const int b = static_cast<int>(blockIdx.z); // (no XCD-contiguous remap: measured 5 % slower here, qk_device.hpp)
	const qk_box bx = boxes[b];
	const int k = bx.lo[2] + static_cast<int>(blockIdx.y);
	if (k > bx.hi[2]) {
		return; // uniform for the workgroup
	}
	RA4 U(a.U_in[b]);
	const int t = threadIdx.x;
	const int64_t rowlen = a.U_in[b].end[0] - a.U_in[b].begin[0]; // cells of a row, ghost cells included (not the row pitch U.js)
	const int64_t slablen = rowlen * (bx.hi[1] - bx.lo[1] + 1);
	const int64_t f = static_cast<int64_t>(blockIdx.x) * RXCELLS + t - 3;
	const bool inside = (f >= 0) && (f < slablen);
	const int64_t fc = inside ? f : 0;
	const int jj = static_cast<int>(fc / rowlen);
	const int i = U.bx + static_cast<int>(fc - jj * rowlen);
	const int j = bx.lo[1] + jj;
	const int64_t o = U.idx(i, j, k);
	double c0[NRAD], p0[NRAD];
#pragma unroll
	for (int n = 0; n < NRAD; ++n) {
		c0[n] = U.p[o + U.ns * (RAD0 + a.pg + n)];
	}
	radPrim(rad, c0, p0);
#pragma unroll
	for (int n = 0; n < NRAD; ++n) {
		s_c[n][t] = c0[n];
		s_p[n][t] = p0[n];
	}
	__syncthreads();
	const int tm2 = max(t - 2, 0), tm1 = max(t - 1, 0), tp1 = min(t + 1, RXB - 1), tp2 = min(t + 2, RXB - 1);
	double edgeL[NRAD];
#pragma unroll
	for (int n = 0; n < NRAD; ++n) {
		if (ORDER == 3) {
			double am, ap;
			ppmEdges(s_p[n][tm2], s_p[n][tm1], p0[n], s_p[n][tp1], s_p[n][tp2], am, ap);
			edgeL[n] = am;
			s_e[n][t] = ap;
		} else if (ORDER == 2) {
			const double slope = MC(s_p[n][tp1] - p0[n], p0[n] - s_p[n][tm1]);
			edgeL[n] = p0[n] - 0.25 * slope;
			s_e[n][t] = slope;
		} else {
			edgeL[n] = p0[n];
		}
	}
	__syncthreads();
	const bool isFace = inside && (i >= bx.lo[0]) && (i <= bx.hi[0] + 1) && (t >= 3) && (t <= RXB - 3);
	double Fo[NRAD] = {0.0, 0.0, 0.0, 0.0};
	if (isFace) {
		double pL[NRAD], cL[NRAD];
#pragma unroll
		for (int n = 0; n < NRAD; ++n) {
			cL[n] = s_c[n][tm1];
			if (ORDER == 3) {
				pL[n] = s_e[n][tm1];
			} else if (ORDER == 2) {
				pL[n] = s_p[n][tm1] + 0.25 * s_e[n][tm1];
			} else {
				pL[n] = s_p[n][tm1];
			}
		}
		radFaceFlux<0>(rad, pL, edgeL, cL, c0, Fo, faceEpsilon(a.eps[0], b, i, j, k, a.pg / NRAD));
#pragma unroll
		for (int n = 0; n < NRAD; ++n) {
			s_f[n][t] = Fo[n];
		}
	}
	__syncthreads();
	if (!isFace || t > RXB - 4) {
		return;
	}
	if (STORE) {
		WA4 F(a.flux[0][b]);
		const int64_t of = F.idx(i, j, k);
#pragma unroll
		for (int n = 0; n < NRAD; ++n) {
			F.p[of + F.ns * (a.pg + n)] = Fo[n];
		}
	}
	if (i <= bx.hi[0]) { // a cell of the box: its right face is the next thread's
		WA4 A(a.acc[b]);
		const int64_t oa = A.idx(i, j, k);
#pragma unroll
		for (int n = 0; n < NRAD; ++n) {
			radStreamStore(&A.p[oa + A.ns * (a.pg + n)], a.dtdx[0] * (Fo[n] - s_f[n][t + 1]));
		}
	}
}

// Y and Z sweeps: one thread marches `strip` cells of a pencil along DIR (strip + 1 faces) with the rolling window of k_rad_flux_march.
// EPI 0: acc += term (Y); 1: PredictStep update; 2: AddFluxesRK2 update (Z; the strip is the whole pencil then, see RadSweep::U_new).
template <int DIR, int ORDER, int EPI, bool STORE>
__global__ void __launch_bounds__(256) k_rad_sweep_march(const qk_box *boxes, Rad rad, RadSweep a, int notb, int strip)
{
	static_assert(DIR == 1 || DIR == 2, "marching sweep: strided directions only");
	static_assert((0.5 - IMEX_a32) == 0.0, "the fused stage drops the old-state fluxes of AddFluxesRK2: PD-ARS only");
	constexpr int OT = 3 - DIR;
	const BlockId blk = xcdContiguousBlock(); // (the two chunks of a row, and consecutive rows, share cache lines: qk_device.hpp)
	const int b = blk.z;
	const qk_box bx = boxes[b];
	const int i = bx.lo[0] + blk.x * 64 + static_cast<int>(threadIdx.x);
	const int oblk = blk.y % notb, sno = blk.y / notb;
	const int ot = bx.lo[OT] + oblk * 4 + static_cast<int>(threadIdx.y);
	const int c0 = bx.lo[DIR] + sno * strip;
	if (i > bx.hi[0] || ot > bx.hi[OT] || c0 > bx.hi[DIR]) {
		return;
	}
	int c1 = min(c0 + strip - 1, bx.hi[DIR]); // last cell of this strip
	if (EPI != 0) {
		// The final sweep writes U_new.  Written IN PLACE (the fab of U_new is the fab of U_in — seen here from the fab pointers themselves: two
		// descriptor tables may describe one storage, which the host cannot tell from the table pointers) a pencil must be one thread's: the
		// cells behind its march are then the only ones it has overwritten.  The first strip takes the whole pencil, the others leave.
		if (a.U_new[b].p == a.U_in[b].p && strip <= bx.hi[DIR] - bx.lo[DIR]) {
			if (sno != 0) {
				return;
			}
			c1 = bx.hi[DIR];
		}
	}
	RA4 U(a.U_in[b]);
	constexpr int M0 = (ORDER == 3) ? 0 : (ORDER == 2) ? 1 : 2;
	constexpr int M1 = (ORDER == 3) ? 5 : (ORDER == 2) ? 4 : 3;
	double c[6][NRAD], p[6][NRAD];
	double carry[NRAD];
	double Fprev[NRAD];
	double nextc[NRAD]; // the cell that enters the window at the next face: loaded one face ahead, so that its latency hides behind a flux
	double accv[NRAD];  // accumulator of the cell that is completed at this face: loaded before the flux, used after it
	double u0v[NRAD];   // (Z) and its state at the start of the substep
	int pos[3];
	pos[0] = i;
	pos[OT] = ot;
	WA4 A(a.acc[b]);
	auto loadCons = [&](int cell, double out[NRAD]) {
		pos[DIR] = cell;
		const int64_t o = U.idx(pos[0], pos[1], pos[2]);
#pragma unroll
		for (int n = 0; n < NRAD; ++n) {
			out[n] = U.p[o + U.ns * (RAD0 + a.pg + n)];
		}
	};
#pragma unroll
	for (int m = M0; m <= M1; ++m) {
		loadCons(c0 + (m - 3), c[m]);
		radPrim(rad, c[m], p[m]);
	}
#pragma unroll
	for (int n = 0; n < NRAD; ++n) {
		if (ORDER == 3) {
			double am, ap;
			ppmEdges(p[0][n], p[1][n], p[2][n], p[3][n], p[4][n], am, ap);
			carry[n] = ap;
		} else if (ORDER == 2) {
			carry[n] = MC(p[3][n] - p[2][n], p[2][n] - p[1][n]);
		}
	}
	for (int face = c0; face <= c1 + 1; ++face) {
		if (face > c0) {
#pragma unroll
			for (int m = M0; m < M1; ++m) {
#pragma unroll
				for (int n = 0; n < NRAD; ++n) {
					c[m][n] = c[m + 1][n];
					p[m][n] = p[m + 1][n];
				}
			}
#pragma unroll
			for (int n = 0; n < NRAD; ++n) {
				c[M1][n] = nextc[n];
			}
			radPrim(rad, c[M1], p[M1]);
			pos[DIR] = face - 1;
			const int64_t oa = A.idx(pos[0], pos[1], pos[2]);
#pragma unroll
			for (int n = 0; n < NRAD; ++n) {
				accv[n] = (EPI == 0) ? A.p[oa + A.ns * (a.pg + n)] : radStreamLoad(&A.p[oa + A.ns * (a.pg + n)]);
			}
			if (EPI != 0) {
				RA4 Uo(a.U0[b]);
				const int64_t oo = Uo.idx(pos[0], pos[1], pos[2]);
#pragma unroll
				for (int n = 0; n < NRAD; ++n) {
					u0v[n] = radStreamLoad2(&Uo.p[oo + Uo.ns * (RAD0 + a.pg + n)]);
				}
			}
		}
		if (face <= c1) {
			loadCons(face + 1 + (M1 - 3), nextc); // (ahead of every cell this thread writes)
		}
		double pL[NRAD], pR[NRAD], Fo[NRAD];
#pragma unroll
		for (int n = 0; n < NRAD; ++n) {
			if (ORDER == 3) {
				double am, ap;
				ppmEdges(p[1][n], p[2][n], p[3][n], p[4][n], p[5][n], am, ap);
				pL[n] = carry[n];
				pR[n] = am;
				carry[n] = ap;
			} else if (ORDER == 2) {
				const double rslope = MC(p[4][n] - p[3][n], p[3][n] - p[2][n]);
				pL[n] = p[2][n] + 0.25 * carry[n];
				pR[n] = p[3][n] - 0.25 * rslope;
				carry[n] = rslope;
			} else {
				pL[n] = p[2][n];
				pR[n] = p[3][n];
			}
		}
		pos[DIR] = face;
		radFaceFlux<DIR>(rad, pL, pR, c[2], c[3], Fo, faceEpsilon(a.eps[DIR], b, pos[0], pos[1], pos[2], a.pg / NRAD));
		if (STORE && (face <= c1 || c1 == bx.hi[DIR])) { // (the face after the strip belongs to the next strip, except at the end of the pencil)
			WA4 F(a.flux[DIR][b]);
			pos[DIR] = face;
			const int64_t of = F.idx(pos[0], pos[1], pos[2]);
#pragma unroll
			for (int n = 0; n < NRAD; ++n) {
				F.p[of + F.ns * (a.pg + n)] = Fo[n];
			}
		}
		if (face > c0) { // cell face-1 has both its faces now
			pos[DIR] = face - 1;
			double cons[NRAD];
#pragma unroll
			for (int n = 0; n < NRAD; ++n) {
				cons[n] = accv[n] + a.dtdx[DIR] * (Fprev[n] - Fo[n]);
			}
			if (EPI == 0) {
				const int64_t oa = A.idx(pos[0], pos[1], pos[2]);
#pragma unroll
				for (int n = 0; n < NRAD; ++n) {
					A.p[oa + A.ns * (a.pg + n)] = cons[n];
				}
			} else {
#pragma unroll
				for (int n = 0; n < NRAD; ++n) {
					if (EPI == 1) {
						cons[n] = u0v[n] + cons[n]; // radiation_system.hpp:681-690
					} else {
						const double U_0 = u0v[n];
						const double U_1 = c[2][n]; // the cell left of this face: still in the window
						cons[n] = (1.0 - IMEX_a32) * U_0 + IMEX_a32 * U_1 + (0.5 * (cons[n])); // :758-759 with the zero-weight term dropped
					}
				}
				if (!radStateValid(rad, cons)) {
					amendRadState(rad, cons);
				}
				WA4 Un(a.U_new[b]);
				const int64_t on = Un.idx(pos[0], pos[1], pos[2]);
#pragma unroll
				for (int n = 0; n < NRAD; ++n) {
					radStreamStore2(&Un.p[on + Un.ns * (RAD0 + a.pg + n)], cons[n]);
				}
			}
		}
#pragma unroll
		for (int n = 0; n < NRAD; ++n) {
			Fprev[n] = Fo[n];
		}
	}
}

template <int ORDER, int STAGE, bool STORE> void launchRadSweeps(qk_level *lev, qk_stream s, Rad rad, RadSweep const &a, int nghost)
{
	if (lev->nboxes == 0) {
		return;
	}
	auto const st = static_cast<hipStream_t>(s);
	{
		const int64_t slab = static_cast<int64_t>(lev->maxlen[0] + 2 * nghost) * lev->maxlen[1];
		const dim3 grid(static_cast<unsigned>((slab + RXCELLS - 1) / RXCELLS), static_cast<unsigned>(lev->maxlen[2]), static_cast<unsigned>(lev->nboxes));
		ProfScope ps(lev->ctx, st, "rad_sweep_x");
		hipLaunchKernelGGL((k_rad_sweep_x<ORDER, STORE>), grid, dim3(RXB), 0, st, lev->d_boxes, rad, a);
	}
	{
		constexpr int STRIP = 16;
		const int notb = (lev->maxlen[2] + 3) / 4;
		const int nstrips = (lev->maxlen[1] + STRIP - 1) / STRIP;
		const dim3 grid((lev->maxlen[0] + 63) / 64, notb * nstrips, lev->nboxes);
		ProfScope ps(lev->ctx, st, "rad_sweep_y");
		hipLaunchKernelGGL((k_rad_sweep_march<1, ORDER, 0, STORE>), grid, dim3(64, 4), 0, st, lev->d_boxes, rad, a, notb, STRIP);
	}
	{
		// written in place (stage 2 of the drivers: U_new is U_in), a pencil is one thread's: the cells behind its march are the only ones it
		// has overwritten.  Otherwise strips, for more waves in flight.
		// (Aliasing cannot be seen from the table POINTERS alone — two tables may describe the same storage, e.g. a sub-level's gathered
		// descriptors.  Stage 2, the stage the drivers run in place, always marches whole pencils; stage 1 does when the tables are the same —
		// and when two tables over one storage reach the kernel after all, it sees the equal fab pointers and falls back to whole pencils itself.)
		const int strip = (STAGE == 2 || a.U_new == a.U_in) ? lev->maxlen[2] : 32;
		const int notb = (lev->maxlen[1] + 3) / 4;
		const int nstrips = (lev->maxlen[2] + strip - 1) / strip;
		const dim3 grid((lev->maxlen[0] + 63) / 64, notb * nstrips, lev->nboxes);
		ProfScope ps(lev->ctx, st, "rad_sweep_z");
		hipLaunchKernelGGL((k_rad_sweep_march<2, ORDER, STAGE, STORE>), grid, dim3(64, 4), 0, st, lev->d_boxes, rad, a, notb, strip);
	}
}

} // namespace

extern "C" {

int qk_rad_ConservedToPrimitive(qk_level *lev, qk_stream s, const qk_rad_traits *rt, const qk_array4 *cons_t, qk_array4 *prim_t, int nghost)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkRad(lev->ctx, rt); rc != QK_OK) {
		return rc;
	}
	QK_REQUIRE(lev->ctx, cons_t && prim_t, "rad ConservedToPrimitive: NULL array");
	const Rad rad(*rt);
	launchRad(lev, s, nghost, -1, "rad_ConservedToPrimitive", [=] __device__(int b, int i, int j, int k, bool valid) {
		if (!valid) {
			return;
		}
		RA4 U(cons_t[b]);
		WA4 P(prim_t[b]);
		const int pg = NRAD * static_cast<int>(blockIdx.z); // component offset of this block's photon group
		double c[NRAD], p[NRAD];
#pragma unroll
		for (int n = 0; n < NRAD; ++n) {
			c[n] = U(i, j, k, RAD0 + pg + n);
		}
		radPrim(rad, c, p);
#pragma unroll
		for (int n = 0; n < NRAD; ++n) {
			P(i, j, k, pg + n) = p[n];
		}
	}, rad.ngroups);
	return radStatus(lev, "rad ConservedToPrimitive");
}

int qk_rad_ComputeFluxes(qk_level *lev, qk_stream s, const qk_rad_traits *rt, int dir, qk_array4 *flux_t, const qk_array4 *left_t,
			 const qk_array4 *right_t, const qk_array4 *cons_t, const qk_array4 *wavespeed_eps)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkRad(lev->ctx, rt); rc != QK_OK) {
		return rc;
	}
	QK_REQUIRE(lev->ctx, flux_t && left_t && right_t && cons_t, "rad ComputeFluxes: NULL array");
	const Rad rad(*rt);
	switch (dir) {
	case 0:
		launchRadComputeFluxes<0>(lev, s, rad, flux_t, left_t, right_t, cons_t, wavespeed_eps);
		break;
	case 1:
		launchRadComputeFluxes<1>(lev, s, rad, flux_t, left_t, right_t, cons_t, wavespeed_eps);
		break;
	case 2:
		launchRadComputeFluxes<2>(lev, s, rad, flux_t, left_t, right_t, cons_t, wavespeed_eps);
		break;
	default:
		return setError(lev->ctx, QK_ERR_INVALID, "rad ComputeFluxes: bad direction");
	}
	return radStatus(lev, "rad ComputeFluxes");
}

int qk_rad_computeRadiationFluxes(qk_level *lev, qk_stream s, const qk_rad_traits *rt, int ndim, int order, const qk_array4 *cons_t,
				  qk_array4 *const flux[3], const qk_array4 *const wavespeed_eps[3])
{
	const qk_array4 *const no_eps[3] = {nullptr, nullptr, nullptr};
	const qk_array4 *const *eps = (wavespeed_eps != nullptr) ? wavespeed_eps : no_eps;
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkRad(lev->ctx, rt); rc != QK_OK) {
		return rc;
	}
	QK_REQUIRE(lev->ctx, cons_t && flux && flux[0] && (ndim < 2 || flux[1]) && (ndim < 3 || flux[2]), "computeRadiationFluxes: NULL array");
	QK_REQUIRE(lev->ctx, order >= 1 && order <= 3, "computeRadiationFluxes: reconstruction order must be 1..3");
	QK_REQUIRE(lev->ctx, ndim == lev->ndim, "computeRadiationFluxes: ndim mismatch");
	const Rad rad(*rt);
#define QK_RAD_DIR(D)                                                                                                                                \
	if (order == 3) {                                                                                                                            \
		launchRadFusedFlux<D, 3>(lev, s, rad, cons_t, flux[D], eps[D]);                                                                              \
	} else if (order == 2) {                                                                                                                     \
		launchRadFusedFlux<D, 2>(lev, s, rad, cons_t, flux[D], eps[D]);                                                                              \
	} else {                                                                                                                                     \
		launchRadFusedFlux<D, 1>(lev, s, rad, cons_t, flux[D], eps[D]);                                                                              \
	}
	// 2-D builds: the radiation fluxes do not permute components with the direction (radiation_system.hpp:1026-1040), so the X2 view of
	// ArrayView_2d.hpp and the cyclic one of the 3-D build address the same cells: the 3-D kernels on a single plane.
	if (ndim >= 2) { // (state arrays carry nghost_cc = 4 ghost cells throughout the library; the slab below is sized for that)
		if (order == 3) {
			launchRadXFlux<3>(lev, s, rad, cons_t, flux[0], 4, eps[0]);
		} else if (order == 2) {
			launchRadXFlux<2>(lev, s, rad, cons_t, flux[0], 4, eps[0]);
		} else {
			launchRadXFlux<1>(lev, s, rad, cons_t, flux[0], 4, eps[0]);
		}
	} else {
		QK_RAD_DIR(0)
	}
#undef QK_RAD_DIR
#define QK_RAD_MARCH(D)                                                                                                                              \
	if (order == 3) {                                                                                                                            \
		launchRadMarchFlux<D, 3>(lev, s, rad, cons_t, flux[D], eps[D]);                                                                              \
	} else if (order == 2) {                                                                                                                     \
		launchRadMarchFlux<D, 2>(lev, s, rad, cons_t, flux[D], eps[D]);                                                                              \
	} else {                                                                                                                                     \
		launchRadMarchFlux<D, 1>(lev, s, rad, cons_t, flux[D], eps[D]);                                                                              \
	}
	if (ndim >= 2) {
		QK_RAD_MARCH(1)
	}
	if (ndim == 3) {
		QK_RAD_MARCH(2)
	}
#undef QK_RAD_MARCH
	return radStatus(lev, "computeRadiationFluxes");
}

int qk_rad_PredictStep(qk_level *lev, qk_stream s, const qk_rad_traits *rt, int ndim, const qk_array4 *old_t, qk_array4 *new_t,
		       const qk_array4 *const fluxArray[3], double dt, const double dx_in[3])
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkRad(lev->ctx, rt); rc != QK_OK) {
		return rc;
	}
	QK_REQUIRE(lev->ctx, old_t && new_t && fluxArray && dx_in && fluxArray[0] && (ndim < 2 || fluxArray[1]) && (ndim < 3 || fluxArray[2]), "rad PredictStep: NULL");
	const Rad rad(*rt);
	const qk_array4 *f0 = fluxArray[0], *f1 = (ndim >= 2) ? fluxArray[1] : nullptr, *f2 = (ndim == 3) ? fluxArray[2] : nullptr;
	const double dx0 = dx_in[0], dx1 = dx_in[1], dx2 = dx_in[2];
	// (the build dimension is a compile-time constant of the kernel: with run-time branches the y and z flux loads of a cell wait for one another)
	auto run = [&](auto ND) {
	constexpr int NDIM = decltype(ND)::value;
	launchRad(lev, s, 0, -1, "rad_PredictStep", [=] __device__(int b, int i, int j, int k, bool valid) {
		if (!valid) {
			return;
		}
		RA4 Uo(old_t[b]);
		WA4 Un(new_t[b]);
		RA4 x1(f0[b]);
		RA4 x2((NDIM >= 2) ? f1[b] : f0[b]); // (descriptors read once per cell, outside the component loops)
		RA4 x3((NDIM == 3) ? f2[b] : f0[b]);
		// one photon group: the state with the flux divergence added
		auto update = [&](int pg, double cons[NRAD]) {
#pragma unroll
			for (int n = 0; n < NRAD; ++n) {
				double d = (dt / dx0) * (x1(i, j, k, pg + n) - x1(i + 1, j, k, pg + n));
				if constexpr (NDIM >= 2) { // radiation_system.hpp:681-690
					d = d + (dt / dx1) * (x2(i, j, k, pg + n) - x2(i, j + 1, k, pg + n));
				}
				if constexpr (NDIM == 3) {
					d = d + (dt / dx2) * (x3(i, j, k, pg + n) - x3(i, j, k + 1, pg + n));
				}
				cons[n] = Uo(i, j, k, RAD0 + pg + n) + d;
			}
		};
		// isStateValid is a property of the whole cell (radiation_system.hpp:626-644: every group), amendRadState then repairs every group
		// (:646-665: a valid group below its floor is lifted as well).  Several groups: one pass for the verdict, one to store.
		bool cellValid = true;
		double cons[NRAD];
		if (rad.ngroups > 1) {
			for (int g = 0; g < rad.ngroups; ++g) {
				update(NRAD * g, cons);
				cellValid = cellValid && radStateValid(rad, cons);
			}
		}
		for (int g = 0; g < rad.ngroups; ++g) {
			const int pg = NRAD * g;
			update(pg, cons);
			if (rad.ngroups == 1) {
				cellValid = radStateValid(rad, cons);
			}
			if (!cellValid) {
				amendRadState(rad, cons);
			}
#pragma unroll
			for (int n = 0; n < NRAD; ++n) {
				Un(i, j, k, RAD0 + pg + n) = cons[n];
			}
		}
	});
	};
	if (ndim == 3) {
		run(std::integral_constant<int, 3>{});
	} else if (ndim == 2) {
		run(std::integral_constant<int, 2>{});
	} else {
		run(std::integral_constant<int, 1>{});
	}
	return radStatus(lev, "rad PredictStep");
}

int qk_rad_AddFluxesRK2(qk_level *lev, qk_stream s, const qk_rad_traits *rt, int ndim, qk_array4 *new_t, const qk_array4 *U0_t, const qk_array4 *U1_t,
			const qk_array4 *const fluxArrayOld[3], const qk_array4 *const fluxArray[3], double dt, const double dx_in[3])
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkRad(lev->ctx, rt); rc != QK_OK) {
		return rc;
	}
	QK_REQUIRE(lev->ctx, new_t && U0_t && U1_t && fluxArrayOld && fluxArray && dx_in && fluxArrayOld[0] && fluxArray[0], "rad AddFluxesRK2: NULL");
	QK_REQUIRE(lev->ctx, (ndim < 2 || (fluxArrayOld[1] && fluxArray[1])) && (ndim < 3 || (fluxArrayOld[2] && fluxArray[2])), "rad AddFluxesRK2: NULL flux");
	const Rad rad(*rt);
	const qk_array4 *o0 = fluxArrayOld[0], *o1 = (ndim >= 2) ? fluxArrayOld[1] : nullptr, *o2 = (ndim == 3) ? fluxArrayOld[2] : nullptr;
	const qk_array4 *f0 = fluxArray[0], *f1 = (ndim >= 2) ? fluxArray[1] : nullptr, *f2 = (ndim == 3) ? fluxArray[2] : nullptr;
	const double dx0 = dx_in[0], dx1 = dx_in[1], dx2 = dx_in[2];
	auto run = [&](auto ND) {
	constexpr int NDIM = decltype(ND)::value;
	launchRad(lev, s, 0, -1, "rad_AddFluxesRK2", [=] __device__(int b, int i, int j, int k, bool valid) {
		if (!valid) {
			return;
		}
		WA4 Un(new_t[b]);
		RA4 U0(U0_t[b]);
		RA4 U1(U1_t[b]);
		RA4 xn(f0[b]);
		RA4 yn((NDIM >= 2) ? f1[b] : f0[b]); // (descriptors read once per cell, outside the component loops)
		RA4 zn((NDIM == 3) ? f2[b] : f0[b]);
		// The old-state fluxes enter with the weight (0.5 - IMEX_a32), which is exactly 0 for the PD-ARS scheme (a32 = 0.5): the term is
		// +-0 for finite fluxes and adding it can only change the sign of a zero result — they are not read (12 of the 36 words this kernel
		// would otherwise stream).  Any other a32 takes the general form.
		constexpr bool useOld = (0.5 - IMEX_a32) != 0.0;
		auto update = [&](int pg, double cons[NRAD]) {
#pragma unroll
			for (int n = 0; n < NRAD; ++n) {
				const double U_0 = U0(i, j, k, RAD0 + pg + n);
				const double U_1 = U1(i, j, k, RAD0 + pg + n);
				double s0 = 0.0;
				double s1 = (dt / dx0) * (xn(i, j, k, pg + n) - xn(i + 1, j, k, pg + n));
				if (useOld) {
					RA4 xo(o0[b]);
					s0 = (dt / dx0) * (xo(i, j, k, pg + n) - xo(i + 1, j, k, pg + n));
				}
				if constexpr (NDIM >= 2) { // radiation_system.hpp:728-757
					s1 = s1 + (dt / dx1) * (yn(i, j, k, pg + n) - yn(i, j + 1, k, pg + n));
					if (useOld) {
						RA4 yo(o1[b]);
						s0 = s0 + (dt / dx1) * (yo(i, j, k, pg + n) - yo(i, j + 1, k, pg + n));
					}
				}
				if constexpr (NDIM == 3) {
					s1 = s1 + (dt / dx2) * (zn(i, j, k, pg + n) - zn(i, j, k + 1, pg + n));
					if (useOld) {
						RA4 zo(o2[b]);
						s0 = s0 + (dt / dx2) * (zo(i, j, k, pg + n) - zo(i, j, k + 1, pg + n));
					}
				}
				// radiation_system.hpp:758-759
				cons[n] = useOld ? (1.0 - IMEX_a32) * U_0 + IMEX_a32 * U_1 + ((0.5 - IMEX_a32) * (s0)) + (0.5 * (s1))
						 : (1.0 - IMEX_a32) * U_0 + IMEX_a32 * U_1 + (0.5 * (s1));
			}
		};
		// whole-cell validity, then every group repaired (see PredictStep); U_new may alias U1: nothing is stored before the verdict
		bool cellValid = true;
		double cons[NRAD];
		if (rad.ngroups > 1) {
			for (int g = 0; g < rad.ngroups; ++g) {
				update(NRAD * g, cons);
				cellValid = cellValid && radStateValid(rad, cons);
			}
		}
		for (int g = 0; g < rad.ngroups; ++g) {
			const int pg = NRAD * g;
			update(pg, cons);
			if (rad.ngroups == 1) {
				cellValid = radStateValid(rad, cons);
			}
			if (!cellValid) {
				amendRadState(rad, cons);
			}
#pragma unroll
			for (int n = 0; n < NRAD; ++n) {
				Un(i, j, k, RAD0 + pg + n) = cons[n];
			}
		}
	});
	};
	if (ndim == 3) {
		run(std::integral_constant<int, 3>{});
	} else if (ndim == 2) {
		run(std::integral_constant<int, 2>{});
	} else {
		run(std::integral_constant<int, 1>{});
	}
	return radStatus(lev, "rad AddFluxesRK2");
}


int qk_rad_stage_fused(qk_level *lev, qk_stream s, const qk_rad_traits *rt, int order, int stage, const qk_array4 *U_in, const qk_array4 *U0, qk_array4 *U_new,
		       qk_array4 *acc, qk_array4 *const flux_out[3], double dt, const double dx_in[3], const qk_array4 *const wavespeed_eps[3])
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkRad(lev->ctx, rt); rc != QK_OK) {
		return rc;
	}
	QK_REQUIRE(lev->ctx, U_in && U0 && U_new && acc && dx_in, "rad stage_fused: NULL array");
	QK_REQUIRE(lev->ctx, order >= 1 && order <= 3, "rad stage_fused: reconstruction order must be 1..3");
	QK_REQUIRE(lev->ctx, stage == 1 || stage == 2, "rad stage_fused: stage must be 1 or 2");
	if (lev->ndim != 3) {
		return setError(lev->ctx, QK_ERR_UNSUPPORTED, "rad stage_fused: 3-D levels (otherwise computeRadiationFluxes + PredictStep / AddFluxesRK2)");
	}
	const bool store = flux_out != nullptr && (flux_out[0] != nullptr || flux_out[1] != nullptr || flux_out[2] != nullptr);
	QK_REQUIRE(lev->ctx, !store || (flux_out[0] && flux_out[1] && flux_out[2]), "rad stage_fused: the face fluxes are stored in all directions or in none");
	const Rad rad(*rt);
	RadSweep a{};
	a.U_in = U_in;
	a.U0 = U0;
	a.U_new = U_new;
	a.acc = acc;
	for (int d = 0; d < 3; ++d) {
		a.flux[d] = store ? flux_out[d] : nullptr;
		a.dtdx[d] = dt / dx_in[d];
		a.eps[d] = (wavespeed_eps != nullptr) ? wavespeed_eps[d] : nullptr;
	}
	// the photon groups are transported independently (radiation_system.hpp:667-771 loops over them inside one kernel): one set of sweeps per group
#define QK_RAD_SWEEPS(O)                                                                                                                             \
	for (int g = 0; g < rad.ngroups; ++g) {                                                                                                      \
	a.pg = NRAD * g;                                                                                                                             \
	if (stage == 1) {                                                                                                                            \
		if (store) {                                                                                                                         \
			launchRadSweeps<O, 1, true>(lev, s, rad, a, 4);                                                                              \
		} else {                                                                                                                             \
			launchRadSweeps<O, 1, false>(lev, s, rad, a, 4);                                                                             \
		}                                                                                                                                    \
	} else {                                                                                                                                     \
		if (store) {                                                                                                                         \
			launchRadSweeps<O, 2, true>(lev, s, rad, a, 4);                                                                              \
		} else {                                                                                                                             \
			launchRadSweeps<O, 2, false>(lev, s, rad, a, 4);                                                                             \
		}                                                                                                                                    \
	}                                                                                                                                            \
	}
	if (order == 3) {
		QK_RAD_SWEEPS(3)
	} else if (order == 2) {
		QK_RAD_SWEEPS(2)
	} else {
		QK_RAD_SWEEPS(1)
	}
#undef QK_RAD_SWEEPS
	return radStatus(lev, "rad stage_fused");
}

} // extern "C"

namespace
{
auto addSourceTermsSingleGroup(qk_level *lev, qk_stream s, const qk_rad_traits *rt, const qk_hydro_traits *t, qk_array4 *cons_t, const qk_array4 *src_t, double dt, int stage,
			       int *d_iteration_counter, int *d_failure_counter, qk_array4 *mirror_t) -> int
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkRad(lev->ctx, rt); rc != QK_OK) {
		return rc;
	}
	if (int rc = checkTraits(lev->ctx, t); rc != QK_OK) {
		return rc;
	}
	if (int rc = needsLibraryEos(lev->ctx, t, "AddSourceTermsSingleGroup"); rc != QK_OK) {
		return rc;
	}
	if (rt->opacity_model == QK_HOOK_COMPILED || rt->thermal_model == QK_HOOK_COMPILED) {
		return setError(lev->ctx, QK_ERR_UNSUPPORTED, "AddSourceTermsSingleGroup",
				"the opacity / emission hooks of this problem are compiled device code: instantiate the kernel in the problem's translation unit "
				"(quokka_amd/host/qk_problem_kernels.hpp)");
	}
	QK_REQUIRE(lev->ctx, cons_t && src_t && d_iteration_counter && d_failure_counter, "AddSourceTermsSingleGroup: NULL");
	QK_REQUIRE(lev->ctx, stage == 1 || stage == 2, "AddSourceTermsSingleGroup: stage must be 1 or 2");
	QK_REQUIRE(lev->ctx, rt->thermal_model == 0 || (rt->thermal_model == 1 && rt->enable_dust_gas_thermal_coupling_model != 0),
		   "AddSourceTermsSingleGroup: thermal_model must be 0, or 1 together with the dust model");
	QK_REQUIRE(lev->ctx, rt->enable_dust_gas_thermal_coupling_model != 0 || (rt->cooling_linear_coeff[0] == 0.0 && rt->cr_heating_rate == 0.0),
		   "AddSourceTermsSingleGroup: the line-cooling / cosmic-ray heating hooks are carried together with the dust model only");
	QK_REQUIRE(lev->ctx, rt->enable_photoelectric_heating == 0, "AddSourceTermsSingleGroup: photoelectric heating is a multigroup model (radiation_dust_system.hpp)");
	if (rt->enable_dust_gas_thermal_coupling_model != 0) {
		QK_REQUIRE(lev->ctx, rt->dust_gas_interaction_coeff > 0.0 && t->mean_molecular_weight > 0.0, "dust model: needs dust_gas_interaction_coeff > 0");
		return (rt->opacity_model == 2) ? radSourceImpl<true, true>(lev, s, rt, t, cons_t, src_t, dt, stage, d_iteration_counter, d_failure_counter, mirror_t)
						: radSourceImpl<false, true>(lev, s, rt, t, cons_t, src_t, dt, stage, d_iteration_counter, d_failure_counter, mirror_t);
	}
	// the temperature-dependent opacities get their own instantiation: the constant-opacity kernel keeps its register budget
	return (rt->opacity_model == 2) ? radSourceImpl<true>(lev, s, rt, t, cons_t, src_t, dt, stage, d_iteration_counter, d_failure_counter, mirror_t)
					: radSourceImpl<false>(lev, s, rt, t, cons_t, src_t, dt, stage, d_iteration_counter, d_failure_counter, mirror_t);
}
} // namespace

extern "C" {

int qk_rad_AddSourceTermsSingleGroup(qk_level *lev, qk_stream s, const qk_rad_traits *rt, const qk_hydro_traits *t, qk_array4 *cons_t,
				     const qk_array4 *src_t, double dt, int stage, int *d_iteration_counter, int *d_failure_counter)
{
	return addSourceTermsSingleGroup(lev, s, rt, t, cons_t, src_t, dt, stage, d_iteration_counter, d_failure_counter, nullptr);
}

int qk_rad_AddSourceTermsSingleGroupMirror(qk_level *lev, qk_stream s, const qk_rad_traits *rt, const qk_hydro_traits *t, qk_array4 *cons_t,
					   const qk_array4 *src_t, double dt, int stage, int *d_iteration_counter, int *d_failure_counter, qk_array4 *mirror_t)
{
	if (lev != nullptr && mirror_t == nullptr) {
		return setError(lev->ctx, QK_ERR_INVALID, "AddSourceTermsSingleGroupMirror", "mirror is NULL");
	}
	return addSourceTermsSingleGroup(lev, s, rt, t, cons_t, src_t, dt, stage, d_iteration_counter, d_failure_counter, mirror_t);
}

} // extern "C"
