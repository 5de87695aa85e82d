// qk_hydro_ops.hip — reference-shaped (one kernel per reference operator) HIP implementation of
// HyperbolicSystem<problem_t> / HydroSystem<problem_t> static methods and the QuokkaSimulation helpers
// on the hydro path.  One launch covers all local boxes (blockIdx.y = box), like the MultiFab
// overloads of amrex::ParallelFor the reference uses.  These operators are the drop-in surface and
// the FOFC fallback; the throughput path is qk_hydro_fused.hip, which reuses the same device functions.
#include "qk_device.hpp"
#include "qk_internal.hpp"

using namespace qk;

namespace
{

// one thread per cell of (valid box b) grown by ng, optionally extended by one face in `facedir`
template <class F> __global__ void __launch_bounds__(256) k_box_cells(const qk_box *boxes, int ndim, int ng, int facedir, F f)
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
	const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
	const int64_t n01 = static_cast<int64_t>(len[0]) * len[1];
	if (t >= n01 * len[2]) {
		return;
	}
	const int k = static_cast<int>(t / n01);
	const int r = static_cast<int>(t - k * n01);
	const int j = r / len[0];
	const int i = r - j * len[0];
	f(b, lo[0] + i, lo[1] + j, lo[2] + k);
}

template <class F> void launchCells(qk_level *lev, qk_stream s, int ng, int facedir, F f)
{
	if (lev->nboxes == 0) {
		return; // a rank without boxes on this level
	}
	const CellLaunch L = cellLaunch(lev, ng, facedir);
	hipLaunchKernelGGL(k_box_cells<F>, L.grid, L.block, 0, static_cast<hipStream_t>(s), lev->d_boxes, lev->ndim, ng, facedir, f);
}

inline auto launchStatus(qk_level *lev, const char *name) -> int
{
	const hipError_t e = hipGetLastError();
	if (e != hipSuccess) {
		return setError(lev->ctx, QK_ERR_HIP, name, hipGetErrorString(e));
	}
	return QK_OK;
}

// view(i+d, j, k) for a DIR-permuted view == array(i,j,k) shifted by d along axis DIR
template <int DIR> QK_DEV auto sh(int d, int axis) -> int { return (axis == DIR) ? d : 0; }

#define QK_DISPATCH_DIR(dir, ...)                                                                                                                  \
	switch (dir) {                                                                                                                               \
	case QK_DIR_X1: {                                                                                                                            \
		constexpr int DIR = 0;                                                                                                               \
		__VA_ARGS__;                                                                                                                            \
	} break;                                                                                                                                     \
	case QK_DIR_X2: {                                                                                                                            \
		constexpr int DIR = 1;                                                                                                               \
		__VA_ARGS__;                                                                                                                            \
	} break;                                                                                                                                     \
	case QK_DIR_X3: {                                                                                                                            \
		constexpr int DIR = 2;                                                                                                               \
		__VA_ARGS__;                                                                                                                            \
	} break;                                                                                                                                     \
	default:                                                                                                                                     \
		return setError(lev->ctx, QK_ERR_INVALID, "bad direction");                                                                          \
	}

} // namespace

extern "C" {

// ------------------------------------------------------------------------------------------------
int qk_ReconstructStatesConstant(qk_level *lev, qk_stream s, int dir, const qk_array4 *q_t, qk_array4 *left_t, qk_array4 *right_t, int nghost,
				 int nvars)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(lev->ctx, q_t && left_t && right_t, "ReconstructStatesConstant: NULL array");
	QK_DISPATCH_DIR(dir, launchCells(lev, s, nghost, -1, [=] __device__(int b, int i, int j, int k) {
				RA4 q(q_t[b]);
				WA4 L(left_t[b]);
				WA4 R(right_t[b]);
				const int im = i - sh<DIR>(1, 0), jm = j - sh<DIR>(1, 1), km = k - sh<DIR>(1, 2);
				for (int n = 0; n < nvars; ++n) {
					L(i, j, k, n) = q(im, jm, km, n);
					R(i, j, k, n) = q(i, j, k, n);
				}
			}));
	return launchStatus(lev, "ReconstructStatesConstant");
}

int qk_ReconstructStatesPLM(qk_level *lev, qk_stream s, int dir, int limiter, const qk_array4 *q_t, qk_array4 *left_t, qk_array4 *right_t,
			    int nghost, int nvars)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(lev->ctx, q_t && left_t && right_t, "ReconstructStatesPLM: NULL array");
	QK_REQUIRE(lev->ctx, limiter == QK_LIMITER_MINMOD || limiter == QK_LIMITER_MC, "ReconstructStatesPLM: bad limiter");
	QK_DISPATCH_DIR(dir, launchCells(lev, s, nghost, -1, [=] __device__(int b, int i, int j, int k) {
				RA4 q(q_t[b]);
				WA4 L(left_t[b]);
				WA4 R(right_t[b]);
				const int dx = sh<DIR>(1, 0), dy = sh<DIR>(1, 1), dz = sh<DIR>(1, 2);
				for (int n = 0; n < nvars; ++n) {
					const double qm2 = q(i - 2 * dx, j - 2 * dy, k - 2 * dz, n);
					const double qm1 = q(i - dx, j - dy, k - dz, n);
					const double q0 = q(i, j, k, n);
					const double qp1 = q(i + dx, j + dy, k + dz, n);
					// hyperbolic_system.hpp:243-246
					const double lslope = (limiter == QK_LIMITER_MINMOD) ? minmod(q0 - qm1, qm1 - qm2) : MC(q0 - qm1, qm1 - qm2);
					const double rslope = (limiter == QK_LIMITER_MINMOD) ? minmod(qp1 - q0, q0 - qm1) : MC(qp1 - q0, q0 - qm1);
					L(i, j, k, n) = qm1 + 0.25 * lslope;
					R(i, j, k, n) = q0 - 0.25 * rslope;
				}
			}));
	return launchStatus(lev, "ReconstructStatesPLM");
}

int qk_ReconstructStatesPPM(qk_level *lev, qk_stream s, int dir, const qk_array4 *q_t, qk_array4 *left_t, qk_array4 *right_t, int nghost, int nvars,
			    int iReadFrom, int iWriteFrom)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(lev->ctx, q_t && left_t && right_t, "ReconstructStatesPPM: NULL array");
	QK_DISPATCH_DIR(dir, launchCells(lev, s, nghost, -1, [=] __device__(int b, int i, int j, int k) {
				RA4 q(q_t[b]);
				WA4 L(left_t[b]);
				WA4 R(right_t[b]);
				const int dx = sh<DIR>(1, 0), dy = sh<DIR>(1, 1), dz = sh<DIR>(1, 2);
				for (int n = 0; n < nvars; ++n) {
					const int nr = iReadFrom + n;
					double am, ap;
					ppmEdges(q(i - 2 * dx, j - 2 * dy, k - 2 * dz, nr), q(i - dx, j - dy, k - dz, nr), q(i, j, k, nr),
						 q(i + dx, j + dy, k + dz, nr), q(i + 2 * dx, j + 2 * dy, k + 2 * dz, nr), am, ap);
					R(i, j, k, iWriteFrom + n) = am;
					L(i + dx, j + dy, k + dz, iWriteFrom + n) = ap;
				}
			}));
	return launchStatus(lev, "ReconstructStatesPPM");
}

// ------------------------------------------------------------------------------------------------
int qk_hydro_ConservedToPrimitive(qk_level *lev, qk_stream s, const qk_hydro_traits *t, const qk_array4 *cons_t, qk_array4 *prim_t, int nghost)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkTraits(lev->ctx, t); rc != QK_OK) {
		return rc;
	}
	QK_REQUIRE(lev->ctx, cons_t && prim_t, "ConservedToPrimitive: NULL array");
	const Eos eos(*t);
	const bool re = (t->reconstruct_eint != 0);
	const int nscalars = t->nscalars;
	launchCells(lev, s, nghost, -1, [=] __device__(int b, int i, int j, int k) {
		RA4 cons(cons_t[b]);
		WA4 prim(prim_t[b]);
		const int64_t c = cons.idx(i, j, k);
		for (int n = 0; n < nscalars; ++n) { // hydro_system.hpp:340-343: passive scalars are reconstructed as they are stored
			prim(i, j, k, NVAR + n) = cons.p[c + cons.ns * (NVAR + n)];
		}
		const double rho = cons.p[c + cons.ns * RHO];
		const double px = cons.p[c + cons.ns * MX];
		const double py = cons.p[c + cons.ns * MY];
		const double pz = cons.p[c + cons.ns * MZ];
		const double E = cons.p[c + cons.ns * ENE];
		const double Eint_aux = cons.p[c + cons.ns * EINT];
		const double vx = px / rho;
		const double vy = py / rho;
		const double vz = pz / rho;
		const double kinetic_energy = 0.5 * rho * (vx * vx + vy * vy + vz * vz);
		const double Eint_cons = E - kinetic_energy;
		const int64_t o = prim.idx(i, j, k);
		prim.p[o + prim.ns * PRHO] = rho;
		prim.p[o + prim.ns * PVX] = vx;
		prim.p[o + prim.ns * PVY] = vy;
		prim.p[o + prim.ns * PVZ] = vz;
		if (re) {
			prim.p[o + prim.ns * PPRES] = Eint_cons / rho;
			prim.p[o + prim.ns * PEINT] = Eint_aux / rho;
		} else {
			const double Pgas = eos.isothermal ? rho * eos.cs_iso * eos.cs_iso : eos.pressure(rho, Eint_cons);
			prim.p[o + prim.ns * PPRES] = Pgas;
			prim.p[o + prim.ns * PEINT] = Eint_aux;
		}
	});
	return launchStatus(lev, "ConservedToPrimitive");
}

int qk_hydro_ComputeFlatteningCoefficients(qk_level *lev, qk_stream s, const qk_hydro_traits *t, int dir, const qk_array4 *prim_t, qk_array4 *chi_t,
					   int nghost)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkTraits(lev->ctx, t); rc != QK_OK) {
		return rc;
	}
	QK_REQUIRE(lev->ctx, prim_t && chi_t, "ComputeFlatteningCoefficients: NULL array");
	const Eos eos(*t);
	const bool re = (t->reconstruct_eint != 0);
	QK_DISPATCH_DIR(dir, launchCells(lev, s, nghost, -1, [=] __device__(int b, int i, int j, int k) {
				RA4 q(prim_t[b]);
				WA4 chi(chi_t[b]);
				const int dx = sh<DIR>(1, 0), dy = sh<DIR>(1, 1), dz = sh<DIR>(1, 2);
				double P[5], rho0 = 0;
_Pragma("unroll")
				for (int m = -2; m <= 2; ++m) {
					const int64_t c = q.idx(i + m * dx, j + m * dy, k + m * dz);
					const double rho = q.p[c + q.ns * PRHO];
					double Pm = q.p[c + q.ns * PPRES];
					if (re) { // hydro_system.hpp:561-577
						Pm = eos.pressure(rho, rho * Pm);
					}
					if (eos.isothermal) { // :579-586
						Pm = rho * (eos.cs_iso * eos.cs_iso);
					}
					P[m + 2] = Pm;
					if (m == 0) {
						rho0 = rho;
					}
				}
				const double vm1 = q(i - dx, j - dy, k - dz, PVX + DIR);
				const double vp1 = q(i + dx, j + dy, k + dz, PVX + DIR);
				chi(i, j, k) = flatteningChi(eos, P[0], P[1], P[2], P[3], P[4], rho0, vm1, vp1);
			}));
	return launchStatus(lev, "ComputeFlatteningCoefficients");
}

int qk_hydro_FlattenShocks(qk_level *lev, qk_stream s, const qk_hydro_traits *t, int dir, const qk_array4 *q_t, const qk_array4 *x1Chi_t,
			   const qk_array4 *x2Chi_t, const qk_array4 *x3Chi_t, qk_array4 *left_t, qk_array4 *right_t, int nghost, int nvars)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkTraits(lev->ctx, t); rc != QK_OK) {
		return rc;
	}
	const int ndim = t->ndim;
	QK_REQUIRE(lev->ctx, q_t && x1Chi_t && left_t && right_t && (ndim < 2 || x2Chi_t) && (ndim < 3 || x3Chi_t), "FlattenShocks: NULL array");
	QK_DISPATCH_DIR(dir, launchCells(lev, s, nghost, -1, [=] __device__(int b, int i, int j, int k) {
				RA4 q(q_t[b]);
				WA4 L(left_t[b]);
				WA4 R(right_t[b]);
				RA4 c1(x1Chi_t[b]);
				// hydro_system.hpp:655-669
				double chi = smin(smin(c1(i - 1, j, k), c1(i, j, k)), c1(i + 1, j, k));
				if (ndim >= 2) {
					RA4 c2(x2Chi_t[b]);
					chi = smin(smin(smin(chi, c2(i, j - 1, k)), c2(i, j, k)), c2(i, j + 1, k));
				}
				if (ndim == 3) {
					RA4 c3(x3Chi_t[b]);
					chi = smin(smin(smin(chi, c3(i, j, k - 1)), c3(i, j, k)), c3(i, j, k + 1));
				}
				const int dx = sh<DIR>(1, 0), dy = sh<DIR>(1, 1), dz = sh<DIR>(1, 2);
				for (int n = 0; n < nvars; ++n) {
					const double a_minus = R(i, j, k, n);
					const double a_plus = L(i + dx, j + dy, k + dz, n);
					const double a_mean = q(i, j, k, n);
					R(i, j, k, n) = chi * a_minus + (1. - chi) * a_mean;
					L(i + dx, j + dy, k + dz, n) = chi * a_plus + (1. - chi) * a_mean;
				}
			}));
	return launchStatus(lev, "FlattenShocks");
}

} // extern "C"

namespace
{
template <int DIR, int RIEMANN, bool TWOD = false>
void launchComputeFluxes(qk_level *lev, qk_stream s, const qk_hydro_traits *t, qk_array4 *flux_t, qk_array4 *fvel_t, const qk_array4 *left_t,
			 const qk_array4 *right_t, const qk_array4 *prim_t, double K_visc)
{
	const Eos eos(*t);
	const bool re = (t->reconstruct_eint != 0);
	const int ndim = t->ndim;
	const int nscalars = t->nscalars;
	const int nmscalars = t->nmscalars;
	launchCells(lev, s, 0, DIR, [=] __device__(int b, int i, int j, int k) {
		RA4 L(left_t[b]);
		RA4 R(right_t[b]);
		RA4 q(prim_t[b]);
		WA4 F(flux_t[b]);
		WA4 V(fvel_t[b]);
		double qL[NVAR], qR[NVAR];
		const int64_t cl = L.idx(i, j, k);
		const int64_t cr = R.idx(i, j, k);
#pragma unroll
		for (int n = 0; n < NVAR; ++n) {
			qL[n] = L.p[cl + L.ns * n];
			qR[n] = R.p[cr + R.ns * n];
		}
		// unit steps of the permuted view: e_n (normal), e_v (view-j), e_w (view-k)
		constexpr int AN = Axes<DIR, TWOD>::n, AV = Axes<DIR, TWOD>::v, AW = Axes<DIR, TWOD>::w;
		const int nx = unit(AN, 0), ny = unit(AN, 1), nz = unit(AN, 2);
		const int vx = unit(AV, 0), vy = unit(AV, 1), vz = unit(AV, 2);
		const int wx = unit(AW, 0), wy = unit(AW, 1), wz = unit(AW, 2);
		const int im = i - nx, jm = j - ny, km = k - nz; // cell to the left of the face
		// hydro_system.hpp:1019
		const double du = q(i, j, k, PVX + AN) - q(im, jm, km, PVX + AN);
		double dvl = 0., dvr = 0., dwl = 0., dwr = 0.;
		if (ndim >= 2) { // :1025-1027
			const int c = PVX + AV;
			dvl = smin(q(im + vx, jm + vy, km + vz, c) - q(im, jm, km, c), q(im, jm, km, c) - q(im - vx, jm - vy, km - vz, c));
			dvr = smin(q(i + vx, j + vy, k + vz, c) - q(i, j, k, c), q(i, j, k, c) - q(i - vx, j - vy, k - vz, c));
		}
		if (ndim == 3) { // :1030-1033
			const int c = PVX + AW;
			dwl = smin(q(im + wx, jm + wy, km + wz, c) - q(im, jm, km, c), q(im, jm, km, c) - q(im - wx, jm - wy, km - wz, c));
			dwr = smin(q(i + wx, j + wy, k + wz, c) - q(i, j, k, c), q(i, j, k, c) - q(i - wx, j - wy, k - wz, c));
		}
		double Fo[NVAR], vn;
		Wave wv;
		faceFlux<DIR, RIEMANN, TWOD>(eos, re, ndim, qL, qR, du, dvl, dvr, dwl, dwr, K_visc, Fo, vn, (nscalars > 0) ? &wv : nullptr);
		const int64_t o = F.idx(i, j, k);
#pragma unroll
		for (int n = 0; n < NVAR; ++n) {
			F.p[o + F.ns * n] = Fo[n];
		}
		for (int n = 0; n < nscalars; ++n) { // hydro_system.hpp:1062-1076, HLLC.hpp:126-136 / LLF.hpp:30-41
			F.p[o + F.ns * (NVAR + n)] = scalarFlux<RIEMANN>(wv, L.p[cl + L.ns * (NVAR + n)], R.p[cr + R.ns * (NVAR + n)]);
		}
		if (nmscalars > 0) { // :1062-1073, :1094-1104 consistent multi-fluid advection: partial-density fluxes = mass flux x upwind proportions
			double fluxSum_U_L = 0, fluxSum_U_R = 0;
			for (int n = 0; n < nmscalars; ++n) {
				fluxSum_U_L += L.p[cl + L.ns * (NVAR + n)];
				fluxSum_U_R += R.p[cr + R.ns * (NVAR + n)];
			}
			const bool fromLeft = (Fo[RHO] >= 0.);
			for (int n = 0; n < nmscalars; ++n) {
				const double U_up = fromLeft ? L.p[cl + L.ns * (NVAR + n)] : R.p[cr + R.ns * (NVAR + n)];
				F.p[o + F.ns * (NVAR + n)] = Fo[RHO] * U_up / (fromLeft ? fluxSum_U_L : fluxSum_U_R);
			}
		}
		V(i, j, k) = vn;
	});
}
} // namespace

extern "C" {

int qk_hydro_ComputeFluxes(qk_level *lev, qk_stream s, const qk_hydro_traits *t, int riemann, int dir, qk_array4 *flux_t, qk_array4 *fvel_t,
			   const qk_array4 *left_t, const qk_array4 *right_t, const qk_array4 *prim_t, double K_visc)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkTraits(lev->ctx, t); rc != QK_OK) {
		return rc;
	}
	QK_REQUIRE(lev->ctx, flux_t && fvel_t && left_t && right_t && prim_t, "ComputeFluxes: NULL array");
	QK_REQUIRE(lev->ctx, riemann == QK_RIEMANN_HLLC || riemann == QK_RIEMANN_LLF || riemann == QK_RIEMANN_HLLD, "ComputeFluxes: unknown Riemann solver");
	QK_REQUIRE(lev->ctx, dir >= 0 && dir < t->ndim, "ComputeFluxes: the direction does not exist in a build of this dimension");
	if (t->ndim == 2 && dir == QK_DIR_X2) { // the X2 view of a 2-D build: index swap instead of the cyclic permutation
		if (riemann == QK_RIEMANN_HLLC) {
			launchComputeFluxes<1, QK_RIEMANN_HLLC, true>(lev, s, t, flux_t, fvel_t, left_t, right_t, prim_t, K_visc);
		} else if (riemann == QK_RIEMANN_HLLD) {
			launchComputeFluxes<1, QK_RIEMANN_HLLD, true>(lev, s, t, flux_t, fvel_t, left_t, right_t, prim_t, K_visc);
		} else {
			launchComputeFluxes<1, QK_RIEMANN_LLF, true>(lev, s, t, flux_t, fvel_t, left_t, right_t, prim_t, K_visc);
		}
		return launchStatus(lev, "ComputeFluxes");
	}
	if (riemann == QK_RIEMANN_HLLC) {
		QK_DISPATCH_DIR(dir, (launchComputeFluxes<DIR, QK_RIEMANN_HLLC>(lev, s, t, flux_t, fvel_t, left_t, right_t, prim_t, K_visc)));
	} else if (riemann == QK_RIEMANN_HLLD) {
		QK_DISPATCH_DIR(dir, (launchComputeFluxes<DIR, QK_RIEMANN_HLLD>(lev, s, t, flux_t, fvel_t, left_t, right_t, prim_t, K_visc)));
	} else {
		QK_DISPATCH_DIR(dir, (launchComputeFluxes<DIR, QK_RIEMANN_LLF>(lev, s, t, flux_t, fvel_t, left_t, right_t, prim_t, K_visc)));
	}
	return launchStatus(lev, "ComputeFluxes");
}

int qk_hydro_ComputeRhsFromFluxes(qk_level *lev, qk_stream s, const qk_hydro_traits *t, qk_array4 *rhs_t, const qk_array4 *const fluxArray[3],
				  const double dx_in[3], int nvars)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkTraits(lev->ctx, t); rc != QK_OK) {
		return rc;
	}
	const int ndim = t->ndim;
	QK_REQUIRE(lev->ctx, rhs_t && fluxArray && dx_in && fluxArray[0] && (ndim < 2 || fluxArray[1]) && (ndim < 3 || fluxArray[2]), "ComputeRhsFromFluxes: NULL");
	const qk_array4 *f0 = fluxArray[0];
	const qk_array4 *f1 = (ndim >= 2) ? fluxArray[1] : nullptr;
	const qk_array4 *f2 = (ndim == 3) ? fluxArray[2] : nullptr;
	const double dx0 = dx_in[0], dx1 = dx_in[1], dx2 = dx_in[2];
	launchCells(lev, s, 0, -1, [=] __device__(int b, int i, int j, int k) {
		WA4 rhs(rhs_t[b]);
		RA4 x1(f0[b]);
		for (int n = 0; n < nvars; ++n) {
			// hydro_system.hpp:469-471
			double r = (1.0 / dx0) * (x1(i, j, k, n) - x1(i + 1, j, k, n));
			if (ndim >= 2) {
				RA4 x2(f1[b]);
				r = r + (1.0 / dx1) * (x2(i, j, k, n) - x2(i, j + 1, k, n));
			}
			if (ndim == 3) {
				RA4 x3(f2[b]);
				r = r + (1.0 / dx2) * (x3(i, j, k, n) - x3(i, j, k + 1, n));
			}
			rhs(i, j, k, n) = r;
		}
	});
	return launchStatus(lev, "ComputeRhsFromFluxes");
}

int qk_hydro_AddInternalEnergyPdV(qk_level *lev, qk_stream s, const qk_hydro_traits *t, qk_array4 *rhs_t, const qk_array4 *cons_t,
				  const double dx_in[3], const qk_array4 *const faceVelArray[3], const qk_iarray4 *redo_t)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkTraits(lev->ctx, t); rc != QK_OK) {
		return rc;
	}
	const int ndim = t->ndim;
	QK_REQUIRE(lev->ctx, rhs_t && cons_t && dx_in && faceVelArray && redo_t && faceVelArray[0] && (ndim < 2 || faceVelArray[1]) && (ndim < 3 || faceVelArray[2]),
		   "AddInternalEnergyPdV: NULL");
	const Eos eos(*t);
	const qk_array4 *v0 = faceVelArray[0];
	const qk_array4 *v1 = (ndim >= 2) ? faceVelArray[1] : nullptr;
	const qk_array4 *v2 = (ndim == 3) ? faceVelArray[2] : nullptr;
	const double dx0 = dx_in[0], dx1 = dx_in[1], dx2 = dx_in[2];
	launchCells(lev, s, 0, -1, [=] __device__(int b, int i, int j, int k) {
		WA4 rhs(rhs_t[b]);
		RA4 U(cons_t[b]);
		CIA4 flag(redo_t[b]);
		const double Pgas = consPressure(eos, U(i, j, k, RHO), U(i, j, k, MX), U(i, j, k, MY), U(i, j, k, MZ), U(i, j, k, ENE));
		double div_v;
		if (flag(i, j, k) == 0) { // hydro_system.hpp:802-804
			RA4 vx(v0[b]);
			div_v = (vx(i + 1, j, k) - vx(i, j, k)) / dx0;
			if (ndim >= 2) {
				RA4 vy(v1[b]);
				div_v = div_v + (vy(i, j + 1, k) - vy(i, j, k)) / dx1;
			}
			if (ndim == 3) {
				RA4 vz(v2[b]);
				div_v = div_v + (vz(i, j, k + 1) - vz(i, j, k)) / dx2;
			}
		} else { // :806-808
			double sum = (U(i + 1, j, k, MX) / U(i + 1, j, k, RHO) - U(i - 1, j, k, MX) / U(i - 1, j, k, RHO)) / dx0;
			if (ndim >= 2) {
				sum = sum + (U(i, j + 1, k, MY) / U(i, j + 1, k, RHO) - U(i, j - 1, k, MY) / U(i, j - 1, k, RHO)) / dx1;
			}
			if (ndim == 3) {
				sum = sum + (U(i, j, k + 1, MZ) / U(i, j, k + 1, RHO) - U(i, j, k - 1, MZ) / U(i, j, k - 1, RHO)) / dx2;
			}
			div_v = 0.5 * sum;
		}
		rhs(i, j, k, EINT) += -Pgas * div_v;
	});
	return launchStatus(lev, "AddInternalEnergyPdV");
}

int qk_hydro_PredictStep(qk_level *lev, qk_stream s, const qk_hydro_traits *t, const qk_array4 *old_t, qk_array4 *new_t, const qk_array4 *rhs_t,
			 double dt, int nvars, qk_iarray4 *redo_t, int64_t *d_redo_count)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkTraits(lev->ctx, t); rc != QK_OK) {
		return rc;
	}
	QK_REQUIRE(lev->ctx, old_t && new_t && rhs_t && redo_t, "PredictStep: NULL array");
	const int nmscalars = (nvars >= NVAR + t->nmscalars) ? t->nmscalars : 0;
	launchCells(lev, s, 0, -1, [=] __device__(int b, int i, int j, int k) {
		RA4 Uo(old_t[b]);
		WA4 Un(new_t[b]);
		RA4 rhs(rhs_t[b]);
		IA4 flag(redo_t[b]);
		double rho_new = 0;
		for (int n = 0; n < nvars; ++n) {
			const double v = Uo(i, j, k, n) + dt * rhs(i, j, k, n);
			Un(i, j, k, n) = v;
			if (n == RHO) {
				rho_new = v;
			}
		}
		// hydro_system.hpp:423-446 isStateValid: rho > 0 and no negative mass scalar
		int bad = (rho_new > 0.) ? 0 : 1;
		for (int idx = 0; idx < nmscalars; ++idx) {
			if (Un(i, j, k, NVAR + idx) < 0.0) {
				bad = 1;
			}
		}
		flag(i, j, k) = bad;
		if (d_redo_count != nullptr && bad != 0) {
			atomicAdd(reinterpret_cast<unsigned long long *>(d_redo_count), 1ULL);
		}
	});
	return launchStatus(lev, "PredictStep");
}

int qk_hydro_EnforceLimits(qk_level *lev, qk_stream s, const qk_hydro_traits *t, double densityFloor, double tempFloor, qk_array4 *state_t)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkTraits(lev->ctx, t); rc != QK_OK) {
		return rc;
	}
	if (int rc = needsLibraryEos(lev->ctx, t, "EnforceLimits"); rc != QK_OK) {
		return rc;
	}
	QK_REQUIRE(lev->ctx, state_t, "EnforceLimits: NULL array");
	const Eos eos(*t);
	const int nscalars = t->nscalars;
	const int nmscalars = t->nmscalars;
	launchCells(lev, s, 0, -1, [=] __device__(int b, int i, int j, int k) {
		WA4 S(state_t[b]);
		const int64_t c = S.idx(i, j, k);
		double U[NVAR];
#pragma unroll
		for (int n = 0; n < NVAR; ++n) {
			U[n] = S.p[c + S.ns * n];
		}
		if (nscalars > 0 && U[RHO] < densityFloor) { // hydro_system.hpp:713-722: the scalars keep their mass when the density is floored
			for (int n = 0; n < nscalars; ++n) {
				auto &q = S.p[c + S.ns * (NVAR + n)];
				q = (densityFloor == 0.0) ? 0.0 : q * (U[RHO] / densityFloor);
			}
		}
		if (nmscalars > 0) { // hydro_system.hpp:725-744: negative partial densities -> small_x rho (Microphysics' default 1e-30), then sum = rho
			const double rho_new = (U[RHO] < densityFloor) ? densityFloor : U[RHO];
			double sp_sum = 0.0;
			for (int idx = 0; idx < nmscalars; ++idx) {
				auto &q = S.p[c + S.ns * (NVAR + idx)];
				if (q < 0.0) {
					q = 1.0e-30 * rho_new;
				}
				sp_sum += q;
			}
			if ((sp_sum > 2.2250738585072014e-308) && (rho_new > 2.2250738585072014e-308)) {
				sp_sum /= rho_new;
				for (int idx = 0; idx < nmscalars; ++idx) {
					S.p[c + S.ns * (NVAR + idx)] /= sp_sum;
				}
			}
		}
		enforceLimits(eos, densityFloor, tempFloor, U);
		S.p[c + S.ns * RHO] = U[RHO];
		S.p[c + S.ns * ENE] = U[ENE];
		S.p[c + S.ns * EINT] = U[EINT];
	});
	return launchStatus(lev, "EnforceLimits");
}

int qk_hydro_SyncDualEnergy(qk_level *lev, qk_stream s, const qk_hydro_traits *t, qk_array4 *state_t, int *d_error_flag)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkTraits(lev->ctx, t); rc != QK_OK) {
		return rc;
	}
	QK_REQUIRE(lev->ctx, state_t, "SyncDualEnergy: NULL array");
	launchCells(lev, s, 0, -1, [=] __device__(int b, int i, int j, int k) {
		WA4 S(state_t[b]);
		const int64_t c = S.idx(i, j, k);
		double U[NVAR];
#pragma unroll
		for (int n = 0; n < NVAR; ++n) {
			U[n] = S.p[c + S.ns * n];
		}
		if (!syncDualEnergy(U)) {
			if (d_error_flag != nullptr) {
				*d_error_flag = 1;
			}
			return;
		}
		S.p[c + S.ns * ENE] = U[ENE];
		S.p[c + S.ns * EINT] = U[EINT];
	});
	return launchStatus(lev, "SyncDualEnergy");
}

} // extern "C"

namespace
{
__global__ void __launch_bounds__(256) k_maxSignal(const qk_box *boxes, const qk_array4 *cons_t, Eos eos, int which, double *result)
{
	const int b = blockIdx.y;
	const qk_box bx = boxes[b];
	const int len0 = bx.hi[0] - bx.lo[0] + 1, len1 = bx.hi[1] - bx.lo[1] + 1, len2 = bx.hi[2] - bx.lo[2] + 1;
	const int64_t ncell = static_cast<int64_t>(len0) * len1 * len2;
	RA4 U(cons_t[b]);
	double m = 0.0;
	for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < ncell; t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
		const int k = static_cast<int>(t / (static_cast<int64_t>(len0) * len1));
		const int r = static_cast<int>(t - static_cast<int64_t>(k) * len0 * len1);
		const int j = r / len0;
		const int i = r - j * len0;
		const int64_t c = U.idx(bx.lo[0] + i, bx.lo[1] + j, bx.lo[2] + k);
		const double v = signalSpeed(eos, which, U.p[c + U.ns * RHO], U.p[c + U.ns * MX], U.p[c + U.ns * MY], U.p[c + U.ns * MZ], U.p[c + U.ns * ENE]);
		m = smax(m, v);
	}
	// wave reduction (64 lanes), workgroup reduction, then one atomic per workgroup (atomics on one word serialise: one per wave made this
	// kernel three times slower than its memory traffic)
	__shared__ double red[4];
	for (int off = 32; off > 0; off >>= 1) {
		m = smax(m, __shfl_xor(m, off));
	}
	if ((threadIdx.x & 63) == 0) {
		red[threadIdx.x >> 6] = m;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		atomicMaxNonNeg(result, smax(smax(red[0], red[1]), smax(red[2], red[3])));
	}
}

__global__ void __launch_bounds__(256) k_fixupState(const qk_box *boxes, qk_array4 *state_t, Eos eos, double densityFloor, double tempFloor, int use_dual_energy,
						    int nscalars, int nmscalars, int *d_error_flag, double *result)
{
	const int b = blockIdx.y;
	const qk_box bx = boxes[b];
	const int len0 = bx.hi[0] - bx.lo[0] + 1, len1 = bx.hi[1] - bx.lo[1] + 1, len2 = bx.hi[2] - bx.lo[2] + 1;
	const int64_t ncell = static_cast<int64_t>(len0) * len1 * len2;
	WA4 S(state_t[b]);
	double m0 = 0.0, m1 = 0.0;
	for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < ncell; t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
		const int k = static_cast<int>(t / (static_cast<int64_t>(len0) * len1));
		const int r = static_cast<int>(t - static_cast<int64_t>(k) * len0 * len1);
		const int j = r / len0;
		const int i = r - j * len0;
		const int64_t c = S.idx(bx.lo[0] + i, bx.lo[1] + j, bx.lo[2] + k);
		double U[NVAR];
#pragma unroll
		for (int n = 0; n < NVAR; ++n) {
			U[n] = S.p[c + S.ns * n];
		}
		// EnforceLimits (hydro_system.hpp:700-773), as qk_hydro_EnforceLimits
		if (nscalars > 0 && U[RHO] < densityFloor) {
			for (int n = 0; n < nscalars; ++n) {
				auto &q = S.p[c + S.ns * (NVAR + n)];
				q = (densityFloor == 0.0) ? 0.0 : q * (U[RHO] / densityFloor);
			}
		}
		if (nmscalars > 0) {
			const double rho_new = (U[RHO] < densityFloor) ? densityFloor : U[RHO];
			double sp_sum = 0.0;
			for (int idx = 0; idx < nmscalars; ++idx) {
				auto &q = S.p[c + S.ns * (NVAR + idx)];
				if (q < 0.0) {
					q = 1.0e-30 * rho_new;
				}
				sp_sum += q;
			}
			if ((sp_sum > 2.2250738585072014e-308) && (rho_new > 2.2250738585072014e-308)) {
				sp_sum /= rho_new;
				for (int idx = 0; idx < nmscalars; ++idx) {
					S.p[c + S.ns * (NVAR + idx)] /= sp_sum;
				}
			}
		}
		enforceLimits(eos, densityFloor, tempFloor, U);
		// SyncDualEnergy (:825-849), as qk_hydro_SyncDualEnergy: a cell with rho <= 0 raises the flag and keeps its energies
		if (use_dual_energy == 1) {
			double V[NVAR];
#pragma unroll
			for (int n = 0; n < NVAR; ++n) {
				V[n] = U[n];
			}
			if (!syncDualEnergy(V)) {
				if (d_error_flag != nullptr) {
					*d_error_flag = 1;
				}
			} else {
				U[ENE] = V[ENE];
				U[EINT] = V[EINT];
			}
		}
		S.p[c + S.ns * RHO] = U[RHO];
		S.p[c + S.ns * ENE] = U[ENE];
		S.p[c + S.ns * EINT] = U[EINT];
		if (result != nullptr) {
			m0 = smax(m0, signalSpeed(eos, 0, U[RHO], U[MX], U[MY], U[MZ], U[ENE]));
			m1 = smax(m1, signalSpeed(eos, 1, U[RHO], U[MX], U[MY], U[MZ], U[ENE]));
		}
	}
	if (result != nullptr) { // wave, then workgroup, then one pair of atomics per workgroup (they serialise on the two result words)
		__shared__ double red[2][4];
		for (int off = 32; off > 0; off >>= 1) {
			m0 = smax(m0, __shfl_xor(m0, off));
			m1 = smax(m1, __shfl_xor(m1, off));
		}
		if ((threadIdx.x & 63) == 0) {
			red[0][threadIdx.x >> 6] = m0;
			red[1][threadIdx.x >> 6] = m1;
		}
		__syncthreads();
		if (threadIdx.x == 0) {
			atomicMaxNonNeg(result, smax(smax(red[0][0], red[0][1]), smax(red[0][2], red[0][3])));
			atomicMaxNonNeg(result + 1, smax(smax(red[1][0], red[1][1]), smax(red[1][2], red[1][3])));
		}
	}
}
} // namespace

extern "C" {

int qk_hydro_ComputeMaxSignalSpeed(qk_level *lev, qk_stream s, const qk_hydro_traits *t, const qk_array4 *cons_t, qk_array4 *maxSignal_t)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkTraits(lev->ctx, t); rc != QK_OK) {
		return rc;
	}
	QK_REQUIRE(lev->ctx, cons_t && maxSignal_t, "ComputeMaxSignalSpeed: NULL array");
	const Eos eos(*t);
	launchCells(lev, s, 0, -1, [=] __device__(int b, int i, int j, int k) {
		RA4 U(cons_t[b]);
		WA4 M(maxSignal_t[b]);
		M(i, j, k) = signalSpeed(eos, 1, U(i, j, k, RHO), U(i, j, k, MX), U(i, j, k, MY), U(i, j, k, MZ), U(i, j, k, ENE));
	});
	return launchStatus(lev, "ComputeMaxSignalSpeed");
}

int qk_hydro_maxSignalSpeedLocal(qk_level *lev, qk_stream s, const qk_hydro_traits *t, int which, const qk_array4 *cons_t, double *d_result)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkTraits(lev->ctx, t); rc != QK_OK) {
		return rc;
	}
	QK_REQUIRE(lev->ctx, cons_t && d_result, "maxSignalSpeedLocal: NULL");
	QK_REQUIRE(lev->ctx, which == 0 || which == 1, "maxSignalSpeedLocal: which must be 0 or 1");
	const Eos eos(*t);
	QK_HIP_CHECK(lev->ctx, hipMemsetAsync(d_result, 0, sizeof(double), static_cast<hipStream_t>(s)));
	const int64_t ncell = static_cast<int64_t>(lev->maxlen[0]) * lev->maxlen[1] * lev->maxlen[2];
	const unsigned gx = static_cast<unsigned>(std::min<int64_t>((ncell + 255) / 256, 512));
	if (lev->nboxes == 0) {
		return QK_OK;
	}
	hipLaunchKernelGGL(k_maxSignal, dim3(gx, lev->nboxes, 1), dim3(256, 1, 1), 0, static_cast<hipStream_t>(s), lev->d_boxes, cons_t, eos, which,
			   d_result);
	return launchStatus(lev, "maxSignalSpeedLocal");
}

// FixupState of a level (reference src/QuokkaSimulation.hpp:761-770): EnforceLimits, then SyncDualEnergy, in one pass over the state, with the
// two CFL maxima of the result (maxSignalSpeedLocal which = 0 / 1) reduced on the way out — what an AMR hierarchy runs on every level after
// reflux + average-down and then needs for the next time step.  Per cell the statements of the two operators above in their order.
int qk_hydro_FixupState(qk_level *lev, qk_stream s, const qk_hydro_traits *t, double densityFloor, double tempFloor, int use_dual_energy, qk_array4 *state_t,
			int *d_error_flag, double *d_max_signal)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkTraits(lev->ctx, t); rc != QK_OK) {
		return rc;
	}
	if (int rc = needsLibraryEos(lev->ctx, t, "FixupState"); rc != QK_OK) {
		return rc;
	}
	QK_REQUIRE(lev->ctx, state_t, "FixupState: NULL array");
	if (d_max_signal != nullptr) {
		QK_HIP_CHECK(lev->ctx, hipMemsetAsync(d_max_signal, 0, 2 * sizeof(double), static_cast<hipStream_t>(s)));
	}
	if (lev->nboxes == 0) {
		return QK_OK;
	}
	const Eos eos(*t);
	const int64_t ncell = static_cast<int64_t>(lev->maxlen[0]) * lev->maxlen[1] * lev->maxlen[2];
	const unsigned gx = static_cast<unsigned>(std::min<int64_t>((ncell + 255) / 256, 512)); // (grid-stride: a few thousand workgroups in all)
	hipLaunchKernelGGL(k_fixupState, dim3(gx, lev->nboxes, 1), dim3(256, 1, 1), 0, static_cast<hipStream_t>(s), lev->d_boxes, state_t, eos, densityFloor, tempFloor,
			   use_dual_energy, t->nscalars, t->nmscalars, d_error_flag, d_max_signal);
	return launchStatus(lev, "FixupState");
}

// ------------------------------------------------------------------------------------------------
int qk_replaceFluxes(qk_level *lev, qk_stream s, int dir, qk_array4 *flux_t, const qk_array4 *FO_t, const qk_iarray4 *redo_t, int face_ncomp)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(lev->ctx, flux_t && FO_t && redo_t, "replaceFluxes: NULL array");
	QK_REQUIRE(lev->ctx, dir >= 0 && dir < 3, "replaceFluxes: bad direction");
	const qk_box *boxes = lev->d_boxes;
	// QuokkaSimulation.hpp:1344-1366: loop over redoFlag grown by 1; faces outside the (un-ghosted) face box are skipped
	launchCells(lev, s, 1, -1, [=] __device__(int b, int i, int j, int k) {
		CIA4 flag(redo_t[b]);
		if (flag(i, j, k) != 1) {
			return;
		}
		WA4 F(flux_t[b]);
		RA4 FO(FO_t[b]);
		const qk_box bx = boxes[b];
		auto inFace = [&](int ii, int jj, int kk) {
			const int c[3] = {ii, jj, kk};
			bool in = true;
#pragma unroll
			for (int d = 0; d < 3; ++d) {
				in = in && (c[d] >= bx.lo[d]) && (c[d] <= bx.hi[d] + ((d == dir) ? 1 : 0));
			}
			return in;
		};
		const int ip = i + ((dir == 0) ? 1 : 0), jp = j + ((dir == 1) ? 1 : 0), kp = k + ((dir == 2) ? 1 : 0);
		const bool lo_in = inFace(i, j, k);
		const bool hi_in = inFace(ip, jp, kp);
		for (int n = 0; n < face_ncomp; ++n) {
			if (lo_in) {
				F(i, j, k, n) = FO(i, j, k, n);
			}
			if (hi_in) {
				F(ip, jp, kp, n) = FO(ip, jp, kp, n);
			}
		}
	});
	return launchStatus(lev, "replaceFluxes");
}

int qk_Saxpy(qk_level *lev, qk_stream s, int dir, qk_array4 *dst_t, double a, const qk_array4 *src_t, int ncomp)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(lev->ctx, dst_t && src_t, "Saxpy: NULL array");
	launchCells(lev, s, 0, dir, [=] __device__(int b, int i, int j, int k) {
		WA4 D(dst_t[b]);
		RA4 S(src_t[b]);
		for (int n = 0; n < ncomp; ++n) {
			D(i, j, k, n) += a * S(i, j, k, n);
		}
	});
	return launchStatus(lev, "Saxpy");
}

// ------------------------------------------------------------------ LinearAdvectionSystem<problem_t> (reference src/linear_advection/linear_advection.hpp)
// The scalar advection solver of the reference shares HyperbolicSystem's reconstruction (qk_ReconstructStates* above); its own three operators:

// ComputeFluxes<DIR>(x1Flux, x1LeftState, x1RightState, advectionVx, nvars)   linear_advection.hpp:165-198: the upwind side of the interface
// (the array element of a face is the same in every permuted view, and a scalar flux permutes no components: no view needed)
int qk_advect_ComputeFluxes(qk_level *lev, qk_stream s, int dir, qk_array4 *flux_t, const qk_array4 *left_t, const qk_array4 *right_t, double vx, int nvars)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(lev->ctx, flux_t && left_t && right_t, "advect ComputeFluxes: NULL array");
	QK_REQUIRE(lev->ctx, dir >= 0 && dir < lev->ndim && nvars >= 1, "advect ComputeFluxes: bad direction / nvars");
	launchCells(lev, s, 0, dir, [=] __device__(int b, int i, int j, int k) {
		WA4 F(flux_t[b]);
		RA4 L(left_t[b]);
		RA4 R(right_t[b]);
		for (int n = 0; n < nvars; ++n) {
			F(i, j, k, n) = (vx < 0.0) ? vx * R(i, j, k, n) : vx * L(i, j, k, n);
		}
	});
	return launchStatus(lev, "advect ComputeFluxes");
}

// PredictStep(consVarOld, consVarNew, fluxArray, dt, dx, nvars)                linear_advection.hpp:82-118
int qk_advect_PredictStep(qk_level *lev, qk_stream s, const qk_array4 *old_t, qk_array4 *new_t, const qk_array4 *const flux_t[3], double dt, const double dx[3],
			  int nvars)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(lev->ctx, old_t && new_t && flux_t && flux_t[0], "advect PredictStep: NULL array");
	const int ndim = lev->ndim;
	QK_REQUIRE(lev->ctx, (ndim < 2 || flux_t[1]) && (ndim < 3 || flux_t[2]), "advect PredictStep: NULL flux array");
	const qk_array4 *fx = flux_t[0], *fy = flux_t[1], *fz = flux_t[2];
	const double dtdx = dt / dx[0], dtdy = (ndim >= 2) ? dt / dx[1] : 0.0, dtdz = (ndim == 3) ? dt / dx[2] : 0.0;
	launchCells(lev, s, 0, -1, [=] __device__(int b, int i, int j, int k) {
		RA4 Uo(old_t[b]);
		WA4 Un(new_t[b]);
		RA4 Fx(fx[b]);
		for (int n = 0; n < nvars; ++n) {
			double sum = dtdx * (Fx(i, j, k, n) - Fx(i + 1, j, k, n));
			if (ndim >= 2) {
				RA4 Fy(fy[b]);
				sum = sum + dtdy * (Fy(i, j, k, n) - Fy(i, j + 1, k, n));
			}
			if (ndim == 3) {
				RA4 Fz(fz[b]);
				sum = sum + dtdz * (Fz(i, j, k, n) - Fz(i, j, k + 1, n));
			}
			Un(i, j, k, n) = Uo(i, j, k, n) + sum;
		}
	});
	return launchStatus(lev, "advect PredictStep");
}

// AddFluxesRK2(U_new, U0, U1, fluxArray, dt, dx, nvars)                        linear_advection.hpp:120-163 (U_new may be U1)
int qk_advect_AddFluxesRK2(qk_level *lev, qk_stream s, qk_array4 *new_t, const qk_array4 *U0_t, const qk_array4 *U1_t, const qk_array4 *const flux_t[3], double dt,
			   const double dx[3], int nvars)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(lev->ctx, new_t && U0_t && U1_t && flux_t && flux_t[0], "advect AddFluxesRK2: NULL array");
	const int ndim = lev->ndim;
	QK_REQUIRE(lev->ctx, (ndim < 2 || flux_t[1]) && (ndim < 3 || flux_t[2]), "advect AddFluxesRK2: NULL flux array");
	const qk_array4 *fx = flux_t[0], *fy = flux_t[1], *fz = flux_t[2];
	const double dtdx = dt / dx[0], dtdy = (ndim >= 2) ? dt / dx[1] : 0.0, dtdz = (ndim == 3) ? dt / dx[2] : 0.0;
	launchCells(lev, s, 0, -1, [=] __device__(int b, int i, int j, int k) {
		RA4 U0(U0_t[b]);
		RA4 U1(U1_t[b]);
		WA4 Un(new_t[b]);
		RA4 Fx(fx[b]);
		for (int n = 0; n < nvars; ++n) {
			const double U_0 = U0(i, j, k, n);
			const double U_1 = U1(i, j, k, n);
			double sum = 0.5 * (dtdx * (Fx(i, j, k, n) - Fx(i + 1, j, k, n)));
			if (ndim >= 2) {
				RA4 Fy(fy[b]);
				sum = sum + 0.5 * (dtdy * (Fy(i, j, k, n) - Fy(i, j + 1, k, n)));
			}
			if (ndim == 3) {
				RA4 Fz(fz[b]);
				sum = sum + 0.5 * (dtdz * (Fz(i, j, k, n) - Fz(i, j, k + 1, n)));
			}
			Un(i, j, k, n) = (0.5 * U_0 + 0.5 * U_1) + sum;
		}
	});
	return launchStatus(lev, "advect AddFluxesRK2");
}

} // extern "C"
