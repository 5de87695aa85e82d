// qk_hydro_fused.hip — the throughput path: one RK stage of the hydro update as six launches
//   k_pre_x  U -> primitive variables (valid+4)                        [HydroSystem::ConservedToPrimitive]
//            + x flattening coefficient, its 3-cell min and D_x        [ComputeFlatteningCoefficients<X1>]
//   k_pre_march<Y>, <Z>  flattening coefficient of the direction, running min over the 3x3 axis neighbours
//            (-> the combined chi of FlattenShocks) and the velocity differences D_y, D_z (valid+1)
//   k_sweep_x / k_sweep_march<Y> / k_sweep_march<Z>:
//            PPM (or PLM / donor) reconstruction + shock flattening + HLLC + flux divergence + face-velocity
//            divergence, fused per sweep direction; nothing but the per-direction half-step fluxes F1 (needed
//            bit-exactly by stage 2: flux_rk2 = 0.5 F1 + 0.5 F2) and a 7-component rhs accumulator touches HBM.
//            The Z sweep carries the epilogue: P dV term, PredictStep, validity flag, EnforceLimits, SyncDualEnergy.
//
// MI355X mapping
//   * X sweep: the pencil direction is the contiguous one, so one thread owns one cell of a flat, row-contiguous
//     slab (rows j = lo..hi of one k-plane are adjacent in memory); +-2 stencil values, right-edge states and face
//     fluxes move between neighbouring lanes through LDS (30 KB per 256-thread workgroup, 82 VGPRs -> 5 waves per
//     SIMD).  The kernel is FP64-issue bound (~1500 VALU slots per cell, 31 divisions + 6 square roots per face).
//   * Y / Z sweeps: lanes stay along x (coalesced 512-B wave loads), each thread MARCHES along the sweep direction
//     with a 5-cell primitive window, the previous right-edge state and the previous face flux in registers: every
//     face flux is evaluated exactly once, no LDS, no barriers.
//     The ~105 doubles of per-thread state hold these kernels at 2 waves per SIMD; the x-sweep body throttled to
//     2 waves per SIMD runs at the same 0.97 ms, i.e. they are bound by FP64 dependency chains, not by HBM.
//     (Measured and rejected, see DESIGN.md §6: next-step prefetch into LDS with global_load_lds_dwordx4 and a rotated
//      loop — bit-exact, +-2 %, i.e. nothing left to hide; 8-wide x tiles with LDS exchange along y/z — 64-byte row
//      segments are 2-4x slower; a 168-VGPR cap for 3 waves per SIMD — spills; fmin/fmax for min/max — slower.)
//   * all arithmetic in qk_device.hpp, shared with the reference-shaped operators -> identical bits.
#include "qk_device.hpp"
#include <algorithm>
#include <cstdlib>

#include "qk_internal.hpp"

using namespace qk;

namespace
{

constexpr int NG = 4; // nghost_cc_ (reference src/simulation.hpp:363)

// geometry of the ghost-4 scratch fab of one box
struct SGeom {
	int64_t off;   // offset of this box in cells (scratch arrays are [array][box][comp][cell])
	int glo[3];    // lower corner including ghosts
	int n[3];      // extent including ghosts
	int64_t ncell; // n0*n1*n2
};

// scratch arrays (in units of "components over total_cells")
enum { S_PRIM = 0, S_AUX = 6, S_RHS = 10, S_NCOMP = 17 };
// S_AUX + 0: chi (combined), +1..3: D_x, D_y, D_z ;  S_RHS + 0..5: flux divergence, +6: div v

struct SweepArgs {
	const qk_box *boxes;
	const SGeom *geom;
	double *scratch;
	int64_t total_cells;
	const qk_array4 *U_in;
	const qk_array4 *U_old;
	qk_array4 *U_out;
	qk_array4 *halfFlux; // of this sweep's direction
	qk_array4 *halfVel;
	qk_iarray4 *redoFlag;
	unsigned long long *redo_count;
	int *error_flag;
	double *max_signal; // optional double[2], see qk_hydro_stage_args::d_max_signal
	double inv_dx; // 1/dx of the sweep direction
	double dx;
	double dt;
	double densityFloor, tempFloor;
	double K_visc;
	int use_dual_energy;
	bool reconstruct_eint;
	bool store_rk2; // stage 2: write flux_rk2 = 0.5 F1 + 0.5 F2 to rk2Flux (never over F1: tile-boundary faces of the x sweep are evaluated twice)
	qk_array4 *rk2Flux;
	int nseg; // segments along the march axis (marching sweeps; see k_pre_march)
};

QK_DEV auto sarr(SweepArgs const &a, int comp) -> double * { return a.scratch + static_cast<int64_t>(comp) * a.total_cells; }

// ---------------------------------------------------------------------------------------------- pencil pre-passes
// The flattening coefficient of direction d and the velocity difference D_d only couple cells ALONG d, so the three
// directions are three pencil passes that each stream the data once:
//   k_pre_x      U -> prim (valid+4), chi_x, m_x = min(chi_x(i-1), chi_x(i), chi_x(i+1)), D_x          [flat, LDS]
//   k_pre_march  Y then Z: rolling 5-cell pressure window in registers -> chi_d, m_d, D_d;  m <- min(m, m_d)
// after the Z pass m is the combined coefficient of FlattenShocks (hydro_system.hpp:655-669).
constexpr int PXB = 256, PXOUT = 250;

__global__ void __launch_bounds__(PXB) k_pre_x(const SGeom *geom, const qk_array4 *U_t, double *scratch, int64_t total, Eos eos, bool re)
{
	__shared__ double s_P[PXB], s_v[PXB], s_chi[PXB];
	const int b = blockIdx.y;
	const SGeom g = geom[b];
	RA4 U(U_t[b]);
	const int t = threadIdx.x;
	const int64_t f = static_cast<int64_t>(blockIdx.x) * PXOUT + t - 3;
	const bool inb = (f >= 0) && (f < g.ncell);
	const int64_t c = f < 0 ? 0 : (f >= g.ncell ? g.ncell - 1 : f);
	const bool owned = inb && (t >= 3) && (t < 3 + PXOUT);

	const int k = static_cast<int>(c / (static_cast<int64_t>(g.n[0]) * g.n[1]));
	const int r = static_cast<int>(c - static_cast<int64_t>(k) * g.n[0] * g.n[1]);
	const int j = r / g.n[0];
	const int i = r - j * g.n[0];
	const int64_t u = U.idx(g.glo[0] + i, g.glo[1] + j, g.glo[2] + k);
	const double rho = U.p[u + U.ns * RHO];
	const double px = U.p[u + U.ns * MX];
	const double py = U.p[u + U.ns * MY];
	const double pz = U.p[u + U.ns * MZ];
	const double E = U.p[u + U.ns * ENE];
	const double Eint_aux = U.p[u + U.ns * EINT];
	const double vx = px / rho;
	const double vy = py / rho;
	const double vz = pz / rho;
	const double kinetic_energy = 0.5 * rho * (vx * vx + vy * vy + vz * vz);
	const double Eint_cons = E - kinetic_energy;
	double *q = scratch + g.off + c;
	double Pphys;
	if (re) {
		const double e = Eint_cons / rho;
		Pphys = eos.pressure(rho, rho * e);
		if (owned) {
			q[(S_PRIM + PPRES) * total] = e;
			q[(S_PRIM + PEINT) * total] = Eint_aux / rho;
		}
	} else {
		Pphys = eos.isothermal ? rho * eos.cs_iso * eos.cs_iso : eos.pressure(rho, Eint_cons);
		if (owned) {
			q[(S_PRIM + PPRES) * total] = Pphys;
			q[(S_PRIM + PEINT) * total] = Eint_aux;
		}
	}
	if (eos.isothermal) {
		Pphys = rho * (eos.cs_iso * eos.cs_iso);
	}
	if (owned) {
		q[(S_PRIM + PRHO) * total] = rho;
		q[(S_PRIM + PVX) * total] = vx;
		q[(S_PRIM + PVY) * total] = vy;
		q[(S_PRIM + PVZ) * total] = vz;
	}
	s_P[t] = Pphys;
	s_v[t] = vx;
	__syncthreads();
	double chi = 1.0;
	if (t >= 2 && t < PXB - 2) {
		chi = flatteningChi(eos, s_P[t - 2], s_P[t - 1], Pphys, s_P[t + 1], s_P[t + 2], rho, s_v[t - 1], s_v[t + 1]);
	}
	s_chi[t] = chi;
	__syncthreads();
	if (owned) {
		q[(S_AUX + 0) * total] = smin(smin(s_chi[t - 1], chi), s_chi[t + 1]);
		q[(S_AUX + 1) * total] = smin(s_v[t + 1] - vx, vx - s_v[t - 1]);
	}
}

// nseg > 1: the march axis is cut into nseg segments, each marched by its own thread (with its own warm-up of the window) — levels with
// few columns (small AMR levels) would otherwise leave most of the chip idle behind one long dependent chain per column.  Every output
// cell belongs to exactly one segment; the arithmetic per cell is unchanged.
template <int DIR> __global__ void __launch_bounds__(256) k_pre_march(const qk_box *boxes, const SGeom *geom, double *scratch, int64_t T, Eos eos, bool re, int nseg)
{
	constexpr int OT = 3 - DIR;
	const int b = static_cast<int>(blockIdx.z) / nseg;
	const int seg = static_cast<int>(blockIdx.z) - b * nseg;
	const qk_box bx = boxes[b];
	const SGeom g = geom[b];
	const int i = bx.lo[0] - 1 + static_cast<int>(blockIdx.x * 64 + threadIdx.x);
	const int ot = bx.lo[OT] - 1 + static_cast<int>(blockIdx.y * 4 + threadIdx.y);
	if (i > bx.hi[0] + 1 || ot > bx.hi[OT] + 1) {
		return;
	}
	// output cells lo-1 .. hi+2 of the box, this segment's share [first, first + nout - 1]
	const int nall = bx.hi[DIR] - bx.lo[DIR] + 1 + 3;
	const int seglen = (nall + nseg - 1) / nseg;
	const int first = bx.lo[DIR] - 1 + seg * seglen;
	const int nout = min(seglen, bx.lo[DIR] - 1 + nall - first);
	if (nout <= 0) {
		return;
	}
	const int lo = first + 1;      // (for a whole box: lo = bx.lo, nvalid = box length)
	const int nvalid = nout - 3;
	const int64_t st[3] = {1, g.n[0], static_cast<int64_t>(g.n[0]) * g.n[1]};
	const int64_t ms = st[DIR];
	int pos[3];
	pos[0] = i;
	pos[OT] = ot;
	pos[DIR] = lo - 4;
	int64_t c = (pos[0] - g.glo[0]) * st[0] + (pos[1] - g.glo[1]) * st[1] + (pos[2] - g.glo[2]) * st[2];
	double *S = scratch + g.off;

	double P[5] = {0., 0., 0., 0., 0.}, v[5] = {0., 0., 0., 0., 0.}, rh[3] = {1., 1., 1.}, ch[3] = {1., 1., 1.};
	// p = lo-4+step is the newest cell of the window;  chi(p-2), then m and D of cell p-3
	for (int step = 0; step < nvalid + 9; ++step, c += ms) {
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			P[m] = P[m + 1];
			v[m] = v[m + 1];
		}
		rh[0] = rh[1];
		rh[1] = rh[2];
		const double rho = S[(S_PRIM + PRHO) * T + c];
		double Pm = S[(S_PRIM + PPRES) * T + c];
		if (re) {
			Pm = eos.pressure(rho, rho * Pm);
		}
		if (eos.isothermal) {
			Pm = rho * (eos.cs_iso * eos.cs_iso);
		}
		P[4] = Pm;
		rh[2] = rho;
		v[4] = S[(S_PRIM + PVX + DIR) * T + c];
		if (step < 4) {
			continue;
		}
		ch[0] = ch[1];
		ch[1] = ch[2];
		ch[2] = flatteningChi(eos, P[0], P[1], P[2], P[3], P[4], rh[0], v[1], v[3]);
		if (step < 6) {
			continue;
		}
		const int64_t cc = c - 3 * ms;
		const double m_in = S[(S_AUX + 0) * T + cc];
		S[(S_AUX + 0) * T + cc] = smin(smin(smin(m_in, ch[0]), ch[1]), ch[2]);
		S[(S_AUX + 1 + DIR) * T + cc] = smin(v[2] - v[1], v[1] - v[0]);
	}
}

// ---------------------------------------------------------------------------------------------- shared sweep pieces
template <int ORDER> QK_DEV void cellEdges(const double qm2, const double qm1, const double q0, const double qp1, const double qp2, double &am, double &ap)
{
	if (ORDER == 3) {
		ppmEdges(qm2, qm1, q0, qp1, qp2, am, ap);
	} else if (ORDER == 2) {
		plmEdges<QK_LIMITER_MINMOD>(qm1, q0, qp1, am, ap); // hydroFluxFunction uses minmod (QuokkaSimulation.hpp:1501)
	} else {
		am = q0;
		ap = q0;
	}
}

// hydro_system.hpp:679-685
QK_DEV void flattenEdges(double chi, double mean, double &am, double &ap)
{
	am = chi * am + (1. - chi) * mean;
	ap = chi * ap + (1. - chi) * mean;
}

// epilogue of one cell (AddInternalEnergyPdV + PredictStep + EnforceLimits + SyncDualEnergy)
// U holds the old state of the cell on entry
QK_DEV void updateCellFrom(SweepArgs const &a, Eos const &eos, int b, int i, int j, int k, double U[NVAR], const double rhs[NVAR], double div_v, double &sig0,
			   double &sig1)
{
	WA4 Un(a.U_out[b]);
	IA4 flag(a.redoFlag[b]);
	// hydro_system.hpp:797-812 (redoFlag == none branch)
	const double Pgas = consPressure(eos, U[RHO], U[MX], U[MY], U[MZ], U[ENE]);
	double r[NVAR];
#pragma unroll
	for (int n = 0; n < NVAR; ++n) {
		r[n] = rhs[n];
	}
	r[EINT] += -Pgas * div_v;
	// hydro_system.hpp:487-495
#pragma unroll
	for (int n = 0; n < NVAR; ++n) {
		U[n] = U[n] + a.dt * r[n];
	}
	const int bad = (U[RHO] > 0.) ? 0 : 1;
	flag(i, j, k) = bad;
	if (bad != 0) {
		atomicAdd(a.redo_count, 1ULL);
	} else {
		enforceLimits(eos, a.densityFloor, a.tempFloor, U);
		if (a.use_dual_energy != 0) {
			if (!syncDualEnergy(U)) {
				*a.error_flag = 1;
			}
		}
	}
	const int64_t cn = Un.idx(i, j, k);
#pragma unroll
	for (int n = 0; n < NVAR; ++n) {
		Un.p[cn + Un.ns * n] = U[n];
	}
	if (a.max_signal != nullptr) {
		sig0 = smax(sig0, signalSpeed(eos, 0, U[RHO], U[MX], U[MY], U[MZ], U[ENE]));
		sig1 = smax(sig1, signalSpeed(eos, 1, U[RHO], U[MX], U[MY], U[MZ], U[ENE]));
	}
}

QK_DEV void updateCell(SweepArgs const &a, Eos const &eos, int b, int i, int j, int k, const double rhs[NVAR], double div_v, double &sig0, double &sig1)
{
	RA4 Uo(a.U_old[b]);
	const int64_t co = Uo.idx(i, j, k);
	double U[NVAR];
#pragma unroll
	for (int n = 0; n < NVAR; ++n) {
		U[n] = Uo.p[co + Uo.ns * n];
	}
	updateCellFrom(a, eos, b, i, j, k, U, rhs, div_v, sig0, sig1);
}

// ---------------------------------------------------------------------------------------------- X sweep (flat + LDS)
constexpr int XB = 256;	 // threads per workgroup
constexpr int XOUT = 250; // cells updated per workgroup (3 halo cells on each side)

template <int ORDER, int STAGE> __global__ void __launch_bounds__(XB) k_sweep_x(SweepArgs a, Eos eos)
{
	__shared__ double s_q[NVAR][XB];  // primitives, later reused for the face fluxes
	__shared__ double s_e[NVAR][XB];  // right-edge states a_plus
	__shared__ double s_d[3][XB];	  // D_V, D_W, face velocity

	const int b = blockIdx.z;
	const qk_box bx = a.boxes[b];
	const SGeom g = a.geom[b];
	const int t = threadIdx.x;
	const int k = bx.lo[2] + blockIdx.y;
	if (k > bx.hi[2]) {
		return; // uniform for the whole workgroup
	}
	// flat slab: rows j = lo.y .. hi.y of plane k are contiguous
	const int64_t rowlen = g.n[0];
	const int64_t slab0 = rowlen * ((bx.lo[1] - g.glo[1]) + static_cast<int64_t>(g.n[1]) * (k - g.glo[2]));
	const int64_t slablen = rowlen * (bx.hi[1] - bx.lo[1] + 1);
	const int64_t f = static_cast<int64_t>(blockIdx.x) * XOUT + t - 3; // flat position inside the slab
	const bool inside = (f >= 0) && (f < slablen);
	const int64_t c = slab0 + (inside ? f : 0);
	const int jj = static_cast<int>((inside ? f : 0) / rowlen);
	const int i = g.glo[0] + static_cast<int>((inside ? f : 0) - jj * rowlen);
	const int j = bx.lo[1] + jj;

	const double *S = a.scratch + g.off;
	const int64_t T = a.total_cells;
	double q0[NVAR];
#pragma unroll
	for (int n = 0; n < NVAR; ++n) {
		q0[n] = S[(S_PRIM + n) * T + c];
		s_q[n][t] = q0[n];
	}
	const double chi = S[(S_AUX + 0) * T + c];
	const double dV = S[(S_AUX + 2) * T + c]; // view-j axis of X1 is y
	const double dW = S[(S_AUX + 3) * T + c]; // view-k axis is z
	s_d[0][t] = dV;
	s_d[1][t] = dW;
	__syncthreads();

	// reconstruct my cell (needs t-2 .. t+2)
	double am[NVAR], ap[NVAR];
	const int tm2 = max(t - 2, 0), tm1 = max(t - 1, 0), tp1 = min(t + 1, XB - 1), tp2 = min(t + 2, XB - 1);
#pragma unroll
	for (int n = 0; n < NVAR; ++n) {
		cellEdges<ORDER>(s_q[n][tm2], s_q[n][tm1], q0[n], s_q[n][tp1], s_q[n][tp2], am[n], ap[n]);
		flattenEdges(chi, q0[n], am[n], ap[n]);
		s_e[n][t] = ap[n];
	}
	__syncthreads();

	// flux at my left face
	double qL[NVAR];
#pragma unroll
	for (int n = 0; n < NVAR; ++n) {
		qL[n] = s_e[n][tm1];
	}
	const double du = q0[PVX] - s_q[PVX][tm1];
	const double dvl = s_d[0][tm1], dwl = s_d[1][tm1];
	double F[NVAR], vf;
	faceFlux<0, QK_RIEMANN_HLLC>(eos, a.reconstruct_eint, 3, qL, am, du, dvl, dV, dwl, dW, a.K_visc, F, vf);

	const bool validRow = inside;
	const bool isFace = validRow && (i >= bx.lo[0]) && (i <= bx.hi[0] + 1) && (t >= 3) && (t <= XB - 3);
	if (STAGE == 1) {
		if (isFace) {
			WA4 HF(a.halfFlux[b]);
			WA4 HV(a.halfVel[b]);
			const int64_t o = HF.idx(i, j, k);
#pragma unroll
			for (int n = 0; n < NVAR; ++n) {
				HF.p[o + HF.ns * n] = F[n];
			}
			HV(i, j, k) = vf;
		}
	} else {
		if (isFace) {
			WA4 HF(a.halfFlux[b]);
			WA4 HV(a.halfVel[b]);
			const int64_t o = HF.idx(i, j, k);
			// flux_rk2 = (0 + 0.5 F1) + 0.5 F2   (QuokkaSimulation.hpp:1106, :1220)
#pragma unroll
			for (int n = 0; n < NVAR; ++n) {
				F[n] = 0.5 * HF.p[o + HF.ns * n] + 0.5 * F[n];
			}
			vf = 0.5 * HV(i, j, k) + 0.5 * vf;
			if (a.store_rk2) {
				WA4 RF(a.rk2Flux[b]);
				const int64_t o2 = RF.idx(i, j, k);
#pragma unroll
				for (int n = 0; n < NVAR; ++n) {
					RF.p[o2 + RF.ns * n] = F[n];
				}
			}
		}
	}
	__syncthreads(); // everyone is done with s_q (primitives)
#pragma unroll
	for (int n = 0; n < NVAR; ++n) {
		s_q[n][t] = F[n];
	}
	s_d[2][t] = vf;
	__syncthreads();

	const bool isCell = validRow && (i >= bx.lo[0]) && (i <= bx.hi[0]) && (t >= 3) && (t < 3 + XOUT);
	if (isCell) {
		double *R = a.scratch + g.off + c;
#pragma unroll
		for (int n = 0; n < NVAR; ++n) {
			// hydro_system.hpp:469
			R[(S_RHS + n) * T] = a.inv_dx * (F[n] - s_q[n][tp1]);
		}
		// hydro_system.hpp:803
		R[(S_RHS + 6) * T] = (s_d[2][tp1] - vf) / a.dx;
	}
}

// ---------------------------------------------------------------------------------------------- Y / Z sweeps (marching)
template <int DIR, int ORDER, int STAGE, bool LAST> __global__ void __launch_bounds__(256) k_sweep_march(SweepArgs a, Eos eos)
{
	static_assert(DIR == 1 || DIR == 2, "marching sweeps are the strided directions");
	const int b = static_cast<int>(blockIdx.z) / a.nseg;
	const int seg = static_cast<int>(blockIdx.z) - b * a.nseg;
	const qk_box bx = a.boxes[b];
	const SGeom g = a.geom[b];
	constexpr int OT = (DIR == 1) ? 2 : 1; // the other transverse axis (besides x)
	const int i_raw = bx.lo[0] + blockIdx.x * 64 + threadIdx.x;
	const int ot_raw = bx.lo[OT] + blockIdx.y * 4 + threadIdx.y;
	// lanes beyond the box stay in the wave (clamped addresses, masked stores) so that wave reductions are well defined
	const bool live = (i_raw <= bx.hi[0]) && (ot_raw <= bx.hi[OT]);
	const int i = min(i_raw, bx.hi[0]);
	const int ot = min(ot_raw, bx.hi[OT]);
	double sig0 = 0., sig1 = 0.;
	const int64_t T = a.total_cells;
	const int64_t st[3] = {1, g.n[0], static_cast<int64_t>(g.n[0]) * g.n[1]};
	const int64_t ms = st[DIR]; // march stride
	// this segment's cells [lo, hi] of the box's march range; its faces lo .. hi+1 (a face between two segments is evaluated by both:
	// same value, stage-1 flux read-only in stage 2)
	const int seglen = (bx.hi[DIR] - bx.lo[DIR] + 1 + a.nseg - 1) / a.nseg;
	const int lo = bx.lo[DIR] + seg * seglen, hi = min(lo + seglen - 1, bx.hi[DIR]);
	const int nvalid = hi - lo + 1;
	if (nvalid <= 0) {
		return; // uniform for the workgroup
	}
	// scratch index of march position p = lo - 3
	int pos[3];
	pos[0] = i;
	pos[OT] = ot;
	pos[DIR] = lo - 3;
	int64_t c = (pos[0] - g.glo[0]) * st[0] + (pos[1] - g.glo[1]) * st[1] + (pos[2] - g.glo[2]) * st[2];
	const double *S = a.scratch + g.off;
	double *Sw = a.scratch + g.off;

	constexpr int AV = Axes<DIR>::v, AW = Axes<DIR>::w;
	double q[5][NVAR];
	double apPrev[NVAR], Fprev[NVAR];
	double vfPrev = 0., dVprev = 0., dWprev = 0.;
#pragma unroll
	for (int n = 0; n < NVAR; ++n) {
		apPrev[n] = 0.;
		Fprev[n] = 0.;
#pragma unroll
		for (int m = 0; m < 5; ++m) {
			q[m][n] = 0.;
		}
	}

	for (int step = 0; step < nvalid + 6; ++step, c += ms) {
		// shift the window and load prim(p)
#pragma unroll
		for (int n = 0; n < NVAR; ++n) {
			q[0][n] = q[1][n];
			q[1][n] = q[2][n];
			q[2][n] = q[3][n];
			q[3][n] = q[4][n];
			q[4][n] = S[(S_PRIM + n) * T + c];
		}
		if (step < 4) {
			continue;
		}
		// cell cc = p - 2 in [lo-1, hi+1]
		const int64_t cc = c - 2 * ms;
		const double chi = S[(S_AUX + 0) * T + cc];
		const double dV = S[(S_AUX + 1 + AV) * T + cc];
		const double dW = S[(S_AUX + 1 + AW) * T + cc];
		double am[NVAR], ap[NVAR];
#pragma unroll
		for (int n = 0; n < NVAR; ++n) {
			cellEdges<ORDER>(q[0][n], q[1][n], q[2][n], q[3][n], q[4][n], am[n], ap[n]);
			flattenEdges(chi, q[2][n], am[n], ap[n]);
		}
		if (step >= 5) {
			// face between cells cc-1 and cc, index = (march coordinate of cc)
			const double du = q[2][PVX + DIR] - q[1][PVX + DIR];
			double F[NVAR], vf;
			faceFlux<DIR, QK_RIEMANN_HLLC>(eos, a.reconstruct_eint, 3, apPrev, am, du, dVprev, dV, dWprev, dW, a.K_visc, F, vf);
			int fidx[3];
			fidx[0] = i;
			fidx[OT] = ot;
			fidx[DIR] = lo + (step - 5);
			if (STAGE == 1) {
				if (live) {
					WA4 HF(a.halfFlux[b]);
					WA4 HV(a.halfVel[b]);
					const int64_t o = HF.idx(fidx[0], fidx[1], fidx[2]);
#pragma unroll
					for (int n = 0; n < NVAR; ++n) {
						HF.p[o + HF.ns * n] = F[n];
					}
					HV(fidx[0], fidx[1], fidx[2]) = vf;
				}
			} else {
				WA4 HF(a.halfFlux[b]);
				WA4 HV(a.halfVel[b]);
				const int64_t o = HF.idx(fidx[0], fidx[1], fidx[2]);
#pragma unroll
				for (int n = 0; n < NVAR; ++n) {
					F[n] = 0.5 * HF.p[o + HF.ns * n] + 0.5 * F[n];
				}
				vf = 0.5 * HV(fidx[0], fidx[1], fidx[2]) + 0.5 * vf;
				if (a.store_rk2 && live) {
					WA4 RF(a.rk2Flux[b]);
					const int64_t o2 = RF.idx(fidx[0], fidx[1], fidx[2]);
#pragma unroll
					for (int n = 0; n < NVAR; ++n) {
						RF.p[o2 + RF.ns * n] = F[n];
					}
				}
			}
			if (step >= 6) {
				// update cell u = cc - 1 (march coordinate lo + step - 6)
				const int64_t cu = cc - ms;
				double rhs[NVAR];
#pragma unroll
				for (int n = 0; n < NVAR; ++n) {
					rhs[n] = S[(S_RHS + n) * T + cu] + a.inv_dx * (Fprev[n] - F[n]);
				}
				const double div_v = S[(S_RHS + 6) * T + cu] + (vf - vfPrev) / a.dx;
				if (LAST) {
					int u[3];
					u[0] = i;
					u[OT] = ot;
					u[DIR] = lo + (step - 6);
					if (live) {
						updateCell(a, eos, b, u[0], u[1], u[2], rhs, div_v, sig0, sig1);
					}
				} else if (live) {
#pragma unroll
					for (int n = 0; n < NVAR; ++n) {
						Sw[(S_RHS + n) * T + cu] = rhs[n];
					}
					Sw[(S_RHS + 6) * T + cu] = div_v;
				}
			}
#pragma unroll
			for (int n = 0; n < NVAR; ++n) {
				Fprev[n] = F[n];
			}
			vfPrev = vf;
		}
#pragma unroll
		for (int n = 0; n < NVAR; ++n) {
			apPrev[n] = ap[n];
		}
		dVprev = dV;
		dWprev = dW;
	}
	if (LAST && a.max_signal != nullptr) {
		// wave reduction (64 lanes), one atomic per wave; max is exact, so the result is deterministic
		for (int off = 32; off > 0; off >>= 1) {
			sig0 = smax(sig0, __shfl_xor(sig0, off));
			sig1 = smax(sig1, __shfl_xor(sig1, off));
		}
		if (threadIdx.x == 0) {
			atomicMaxNonNeg(&a.max_signal[0], sig0);
			atomicMaxNonNeg(&a.max_signal[1], sig1);
		}
	}
}

// segments along the march axis `dir` (transverse axes 0 and `ot`): enough threads to fill the chip when the level has few columns, at
// least 16 cells per segment (each segment re-marches 6 cells of warm-up); 1 for the large levels (one column per thread is enough)
auto marchSegments(const qk_level *lev, int dir, int ot) -> int
{
	if (const char *e = std::getenv("QK_MARCH_SEGMENTS")) {
		return std::max(1, std::atoi(e));
	}
	const int64_t cols = static_cast<int64_t>(lev->nboxes) * lev->maxlen[0] * lev->maxlen[ot];
	const int64_t want = (131072 + cols - 1) / std::max<int64_t>(cols, 1);
	const int cap = std::max(1, lev->maxlen[dir] / 16);
	return static_cast<int>(std::min<int64_t>(std::max<int64_t>(want, 1), std::min(cap, 16)));
}

auto buildGeom(qk_level *lev) -> int
{
	if (lev->d_sgeom != nullptr) {
		return QK_OK;
	}
	std::vector<SGeom> g(lev->nboxes);
	int64_t off = 0;
	for (int b = 0; b < lev->nboxes; ++b) {
		g[b].off = off;
		g[b].ncell = 1;
		for (int d = 0; d < 3; ++d) {
			g[b].glo[d] = lev->boxes[b].lo[d] - NG;
			g[b].n[d] = lev->boxes[b].hi[d] - lev->boxes[b].lo[d] + 1 + 2 * NG;
			g[b].ncell *= g[b].n[d];
		}
		off += g[b].ncell;
	}
	void *d = nullptr;
	QK_HIP_CHECK(lev->ctx, hipMalloc(&d, sizeof(SGeom) * lev->nboxes));
	QK_HIP_CHECK(lev->ctx, hipMemcpy(d, g.data(), sizeof(SGeom) * lev->nboxes, hipMemcpyHostToDevice));
	lev->d_sgeom = d;
	lev->sgeom_total_cells = off;
	return QK_OK;
}

template <int ORDER, int STAGE> void launchSweeps(qk_level *lev, hipStream_t s, SweepArgs a, Eos eos, const qk_hydro_stage_args *args)
{
	// X
	{
		SweepArgs ax = a;
		ax.halfFlux = args->halfFlux[0];
		ax.halfVel = args->halfVel[0];
		ax.rk2Flux = args->fluxRk2[0];
		ax.inv_dx = 1.0 / args->dx[0];
		ax.dx = args->dx[0];
		const int64_t slab = static_cast<int64_t>(lev->maxlen[0] + 2 * NG) * lev->maxlen[1];
		const dim3 grid(static_cast<unsigned>((slab + XOUT - 1) / XOUT), static_cast<unsigned>(lev->maxlen[2]), static_cast<unsigned>(lev->nboxes));
		ProfScope ps(lev->ctx, s, "k_sweep_x");
		hipLaunchKernelGGL((k_sweep_x<ORDER, STAGE>), grid, dim3(XB), 0, s, ax, eos);
	}
	// Y
	{
		SweepArgs ay = a;
		ay.halfFlux = args->halfFlux[1];
		ay.halfVel = args->halfVel[1];
		ay.rk2Flux = args->fluxRk2[1];
		ay.inv_dx = 1.0 / args->dx[1];
		ay.dx = args->dx[1];
		ay.nseg = marchSegments(lev, 1, 2);
		const dim3 grid((lev->maxlen[0] + 63) / 64, (lev->maxlen[2] + 3) / 4, lev->nboxes * ay.nseg);
		ProfScope ps(lev->ctx, s, "k_sweep_y");
		hipLaunchKernelGGL((k_sweep_march<1, ORDER, STAGE, false>), grid, dim3(64, 4), 0, s, ay, eos);
	}
	// Z (+ epilogue)
	{
		SweepArgs az = a;
		az.halfFlux = args->halfFlux[2];
		az.halfVel = args->halfVel[2];
		az.rk2Flux = args->fluxRk2[2];
		az.inv_dx = 1.0 / args->dx[2];
		az.dx = args->dx[2];
		az.nseg = marchSegments(lev, 2, 1);
		const dim3 grid((lev->maxlen[0] + 63) / 64, (lev->maxlen[1] + 3) / 4, lev->nboxes * az.nseg);
		ProfScope ps(lev->ctx, s, "k_sweep_z");
		hipLaunchKernelGGL((k_sweep_march<2, ORDER, STAGE, true>), grid, dim3(64, 4), 0, s, az, eos);
	}
}

} // namespace

extern "C" {

int64_t qk_hydro_stage_scratch_bytes(qk_level *lev, const qk_hydro_traits *t)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkTraits(lev->ctx, t); rc != QK_OK) {
		return rc;
	}
	int64_t cells = 0;
	for (int b = 0; b < lev->nboxes; ++b) {
		int64_t n = 1;
		for (int d = 0; d < 3; ++d) {
			n *= lev->boxes[b].hi[d] - lev->boxes[b].lo[d] + 1 + 2 * NG;
		}
		cells += n;
	}
	return cells * S_NCOMP * static_cast<int64_t>(sizeof(double));
}

int qk_hydro_stage_fused(qk_level *lev, qk_stream stream, const qk_hydro_traits *t, const qk_hydro_stage_args *args)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	qk_ctx *ctx = lev->ctx;
	if (int rc = checkTraits(ctx, t); rc != QK_OK) {
		return rc;
	}
	QK_REQUIRE(ctx, args != nullptr, "qk_hydro_stage_fused: NULL args");
	if (lev->nboxes == 0) {
		return QK_OK; // a rank without boxes on this level
	}
	if (t->nscalars != 0) {
		return setError(ctx, QK_ERR_UNSUPPORTED, "qk_hydro_stage_fused: no passive scalars (use the reference-shaped operators)");
	}
	if (t->ndim != 3) {
		return setError(ctx, QK_ERR_UNSUPPORTED, "qk_hydro_stage_fused: 3-D only (use the reference-shaped operators in 1-D)");
	}
	QK_REQUIRE(ctx, args->K_visc >= 0.0, "qk_hydro_stage_fused: negative artificial-viscosity coefficient");
	QK_REQUIRE(ctx, args->stage == 1 || args->stage == 2, "qk_hydro_stage_fused: stage must be 1 or 2");
	QK_REQUIRE(ctx, args->reconstruction_order >= 1 && args->reconstruction_order <= 3, "qk_hydro_stage_fused: reconstruction_order must be 1..3");
	QK_REQUIRE(ctx, args->U_in && args->U_old && args->U_out && args->redoFlag && args->d_redo_count && args->d_error_flag && args->scratch,
		   "qk_hydro_stage_fused: NULL array");
	for (int d = 0; d < 3; ++d) {
		QK_REQUIRE(ctx, args->halfFlux[d] && args->halfVel[d], "qk_hydro_stage_fused: NULL halfFlux/halfVel");
		QK_REQUIRE(ctx, args->store_flux_rk2 == 0 || args->stage != 2 || (args->fluxRk2[d] != nullptr && args->fluxRk2[d] != args->halfFlux[d]),
			   "qk_hydro_stage_fused: store_flux_rk2 needs fluxRk2[d], distinct from halfFlux[d]");
		QK_REQUIRE(ctx, lev->maxlen[d] >= 1, "qk_hydro_stage_fused: empty box");
	}
	QK_REQUIRE(ctx, args->scratch_bytes >= qk_hydro_stage_scratch_bytes(lev, t), "qk_hydro_stage_fused: scratch too small");

	if (int rc = buildGeom(lev); rc != QK_OK) {
		return rc;
	}
	auto s = static_cast<hipStream_t>(stream);
	const Eos eos(*t);
	const bool re = (t->reconstruct_eint != 0);
	auto *scratch = static_cast<double *>(args->scratch);
	const int64_t T = lev->sgeom_total_cells;
	const SGeom *geom = static_cast<const SGeom *>(lev->d_sgeom);
	const qk_box *boxes = lev->d_boxes;

	int64_t maxcell = 1;
	for (int d = 0; d < 3; ++d) {
		maxcell *= lev->maxlen[d] + 2 * NG;
	}

	// 1.-3. pencil pre-passes: primitives (valid + 4), combined flattening coefficient and velocity differences (valid + 1)
	{
		ProfScope ps(ctx, s, "k_pre_x");
		const dim3 grid(static_cast<unsigned>((maxcell + PXOUT - 1 + 3) / PXOUT), lev->nboxes, 1);
		hipLaunchKernelGGL(k_pre_x, grid, dim3(PXB), 0, s, geom, args->U_in, scratch, T, eos, re);
	}
	{
		ProfScope ps(ctx, s, "k_pre_y");
		const int nseg = marchSegments(lev, 1, 2);
		const dim3 grid((lev->maxlen[0] + 2 + 63) / 64, (lev->maxlen[2] + 2 + 3) / 4, lev->nboxes * nseg);
		hipLaunchKernelGGL(k_pre_march<1>, grid, dim3(64, 4), 0, s, boxes, geom, scratch, T, eos, re, nseg);
	}
	{
		ProfScope ps(ctx, s, "k_pre_z");
		const int nseg = marchSegments(lev, 2, 1);
		const dim3 grid((lev->maxlen[0] + 2 + 63) / 64, (lev->maxlen[1] + 2 + 3) / 4, lev->nboxes * nseg);
		hipLaunchKernelGGL(k_pre_march<2>, grid, dim3(64, 4), 0, s, boxes, geom, scratch, T, eos, re, nseg);
	}

	// 4. sweeps
	SweepArgs a{};
	a.boxes = boxes;
	a.geom = geom;
	a.scratch = scratch;
	a.total_cells = T;
	a.U_in = args->U_in;
	a.U_old = args->U_old;
	a.U_out = args->U_out;
	a.redoFlag = args->redoFlag;
	a.redo_count = reinterpret_cast<unsigned long long *>(args->d_redo_count);
	a.error_flag = args->d_error_flag;
	a.max_signal = args->d_max_signal;
	a.dt = args->dt;
	a.densityFloor = args->densityFloor;
	a.tempFloor = args->tempFloor;
	a.K_visc = args->K_visc;
	a.use_dual_energy = args->use_dual_energy;
	a.reconstruct_eint = re;
	a.store_rk2 = (args->store_flux_rk2 != 0);

#define QK_LAUNCH(ORDER)                                                                                                                             \
	if (args->stage == 1) {                                                                                                                      \
		launchSweeps<ORDER, 1>(lev, s, a, eos, args);                                                                                        \
	} else {                                                                                                                                     \
		launchSweeps<ORDER, 2>(lev, s, a, eos, args);                                                                                        \
	}
	if (args->reconstruction_order == 3) {
		QK_LAUNCH(3)
	} else if (args->reconstruction_order == 2) {
		QK_LAUNCH(2)
	} else {
		QK_LAUNCH(1)
	}
#undef QK_LAUNCH
	QK_HIP_CHECK(ctx, hipGetLastError());
	return QK_OK;
}

} // extern "C"
