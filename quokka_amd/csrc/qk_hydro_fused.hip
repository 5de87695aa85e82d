// qk_hydro_fused.hip — the throughput path: one RK stage of the hydro update as FOUR launches over all boxes of a level
//   k_pre3           U -> the combined flattening coefficient of FlattenShocks [ComputeFlatteningCoefficients<X1,X2,X3> + FlattenShocks]
//                    and the velocity differences D_x, D_y, D_z of the carbuncle switch, on valid + 1: x-y tile + march along z
//   k_sweep_x        PPM (or PLM / donor) reconstruction + shock flattening + HLLC + flux divergence + face-velocity divergence along x:
//                    one thread per cell of a row-contiguous slab, +-2 stencil, edge states and fluxes through LDS
//   k_sweep_march<Y> the same along y: lanes along x, each thread marches along y with a 5-cell primitive window in registers
//   k_sweep_march<Z> the same along z + the epilogue: P dV term, PredictStep, validity flag, EnforceLimits, SyncDualEnergy, CFL maxima
// Every sweep converts U to primitives itself (no primitive arrays in HBM); a 7-double right-hand-side accumulator travels X -> Y -> Z.
// Two ways to form the RK2 average (qk_hydro_stage_args::rk2_carry_rhs):
//   0  the reference's: flux_rk2 = 0.5 F1 + 0.5 F2 face by face — stage 1 stores F1 (7 doubles per face and direction), stage 2 reads
//      it back; bit-identical to the reference-shaped operators, needed when flux registers or the first-order correction consume flux_rk2;
//   1  carried half step: stage 1 stores, per CELL, S = U_old + (dt/2) rhs_1 (the right-hand side of stage 1 with its P dV term) and P(U_old)
//      — 7 doubles once instead of 3 x 7 face values —, stage 2 finishes U_new = S + (dt/2) rhs_2 with the P dV term on the stored pressure:
//      U_old + dt (rhs_1 + rhs_2) / 2 in exact arithmetic, a different rounding (~1e-16 per step; north_star allows 1e-12).  Stage 2 reads
//      neither the old state nor a second right-hand side: 29 instead of 35 doubles per cell in its final sweep.
// Measured characteristics (profiles/round2, round3): the X sweep is FP64-issue bound (~1160 VALU instructions per cell at
// 5 waves per SIMD), the marching sweeps (228-251 VGPRs, 2 waves per SIMD) move their real HBM traffic at ~5 TB/s.
// All arithmetic lives in qk_device.hpp, shared with the reference-shaped operators -> identical bits.
#include "qk_device.hpp"
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "qk_internal.hpp"

using namespace qk;

namespace
{

constexpr int NG = 4; // nghost_cc_ (reference src/simulation.hpp:363)

// Non-temporal hints on the streams a launch touches exactly once: the stores of the flux-divergence accumulator (X, Y), of the new state and of the
// carried half step (final sweep), and the loads of the accumulator (Y, final sweep).  Measured same-box (profiles/round4/ab12_*): Z -1.3 ... -1.5 %,
// Y -1 %, 512^3 headline +1.6 %.  Not on the state reads (neighbouring lanes and sweeps share their lines) and not on the pre-pass's outputs
// (measured: the pre-pass 5 - 8 % slower with them).  QK_NT=0: no hints, 1: stores only.
#ifndef QK_NT
#define QK_NT 2
#endif
template <class P> QK_DEV void streamStore(P *p, double v)
{
#if QK_NT >= 1
	__builtin_nontemporal_store(v, p);
#else
	*p = v;
#endif
}
template <class P> QK_DEV auto streamLoad(P *p) -> double
{
#if QK_NT >= 2
	return __builtin_nontemporal_load(p);
#else
	return *p;
#endif
}

// geometry of the ghost-4 scratch fab of one box
struct SGeom {
	int64_t off;   // offset of this box in cells (scratch arrays are [array][box][comp][cell])
	int glo[3];    // lower corner including ghosts
	int n[3];      // extent including ghosts
	int pitch;     // doubles between consecutive rows (>= n[0]; see rowPitchAligned)
	int64_t ncell; // pitch*n1*n2
};

// scratch arrays (in units of "components over total_cells")
enum { S_AUX = 0, S_RHS = 4 };
// S_AUX + 0: chi (combined), +1..3: D_x, D_y, D_z ;  S_RHS + 0..nv-1: flux divergence (nv = 6 + passive scalars), + nv: div v
constexpr auto scratchComps(int nscalars) -> int { return S_RHS + NVAR + nscalars + 1; }
constexpr int QK_FUSED_MAX_SCALARS = 3; // the fused sweeps are instantiated for 0..3 passive scalars (PassiveScalar 1, HydroContact 2, ...)

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
	double inv_dx0, dx0; // the x spacing (the Y sweep that carries the X sweep: FUSEX)
	double dx3[3]; // (FOFC pass: the cell-centred velocity divergence of a flagged cell)
	double dt;
	double densityFloor, tempFloor;
	double K_visc;
	int use_dual_energy;
	bool reconstruct_eint;
	bool same_old; // U_old is U_in (stage 1): the final sweep keeps the conserved state of its last three march positions in LDS instead of re-reading it
	bool store_rk2; // stage 2: write flux_rk2 = 0.5 F1 + 0.5 F2 to rk2Flux (never over F1: tile-boundary faces of the x sweep are evaluated twice)
	qk_array4 *rk2Flux;
	qk_array4 *rhs1; // carried half step (rk2_carry_rhs): per cell U_old + (dt/2) rhs_1 (nv components) + P(U_old), written by stage 1, read by stage 2
	const qk_carray4 *fluxMask; // carried form only, optional: cells whose faces keep F1 (stage 1 -> halfFlux) and receive flux_rk2 (stage 2 -> rk2Flux)
	int nseg; // segments along the march axis (marching sweeps; see k_pre_march)
	// the primitive hand-off between the stages of a step (qk_hydro_stage_args::prim_out / prim_in): the final sweep of stage 1 stores
	// (rho, v, P, E_int) of the cell it completes — consToPrim of the state it would have stored, whose quotients its limits already formed — and
	// every kernel of stage 2 reads its input as primitives: no conversion in the pre-pass, the three sweeps (4.9 per cell and stage), same bytes
	bool prim_in, prim_out;
};

QK_DEV auto sarr(SweepArgs const &a, int comp) -> double * { return a.scratch + static_cast<int64_t>(comp) * a.total_cells; }

// ---------------------------------------------------------------------------------------------- flattening pre-pass
// ONE kernel produces what the sweeps need besides the state itself: the combined flattening coefficient of FlattenShocks
// (hydro_system.hpp:655-669) and the velocity differences D_x, D_y, D_z of the carbuncle switch (:1019-1034) on valid+1.  Round 1 did this
// with three pencil passes that also materialised the primitive variables (272 B of HBM traffic per cell against 80 B here: U in, four
// doubles out; the sweeps now convert U to primitives themselves, 3 quotients sharing one reciprocal per cell and sweep).
//
// A cell's result needs pressure / velocity from a plus-shaped stencil: P at +-3 and v_d at +-2 along each axis d (chi_d at +-1, each
// from P +-2 and v_d +-1), nothing diagonal.  Mapping: a workgroup owns an x-y tile of PT_X x PT_Y cells and MARCHES along z —
//   * z direction: rolling register windows (P 5 deep, v_z 5, rho 3, chi_z 3), as the marching sweeps do;
//   * x and y directions: the plane's P, rho, v_x, v_y go through LDS with the plus-shaped halo (3 columns / 3 rows beyond the tile, no
//     corners), chi_x / chi_y of the tile and of its one-cell rim are exchanged through LDS as well; the in-plane results wait three
//     planes in a register FIFO until the z window has caught up with them.
// Halo cells are converted by the threads left over after every thread converted its own cell (900 conversions per 462 outputs and
// plane); their loads hit L2 (the neighbouring tiles read the same rows), so HBM sees each conserved value about once.
#ifndef QK_PT_Y
#define QK_PT_Y 7
#define QK_PT_THREADS 512
#endif
constexpr int PT_X = 66, PT_Y = QK_PT_Y, PT_THREADS = QK_PT_THREADS; // 2 x-tiles cover the 130 columns of a 128-cell box + rim
constexpr int PT_OWN = PT_X * PT_Y;		     // 462 threads own a column
constexpr int PT_HALO_Y = 6 * PT_X;		     // rows -3..-1 and PT_Y..PT_Y+2
constexpr int PT_HALO = PT_HALO_Y + 6 * PT_Y;	     // + columns -3..-1 and PT_X..PT_X+2 of the tile's rows
static_assert(PT_HALO <= PT_THREADS && PT_OWN <= PT_THREADS, "one halo cell per thread");

struct PlaneCell {
	double rho, vx, vy, vz, P;
};

// hydro_system.hpp:138-196 + the pressure ComputeFlatteningCoefficients works with (:560-586): one cell of U -> (rho, v, P)
struct PlaneRaw {
	double rho, mx, my, mz, E;
};
QK_DEV auto planeLoad(RA4 const &U, int64_t u) -> PlaneRaw
{
	PlaneRaw r;
	r.rho = U.p[u + U.ns * RHO];
	r.mx = U.p[u + U.ns * MX];
	r.my = U.p[u + U.ns * MY];
	r.mz = U.p[u + U.ns * MZ];
	r.E = U.p[u + U.ns * ENE];
	return r;
}
QK_DEV auto planeCell(Eos const &eos, bool re, PlaneRaw const &r) -> PlaneCell
{
	PlaneCell c;
	c.rho = r.rho;
	const Recip R = recipOf(c.rho);
	c.vx = divBy(r.mx, R);
	c.vy = divBy(r.my, R);
	c.vz = divBy(r.mz, R);
	if (eos.isothermal) {
		c.P = c.rho * (eos.cs_iso * eos.cs_iso);
		return c;
	}
	const double kinetic_energy = 0.5 * c.rho * (c.vx * c.vx + c.vy * c.vy + c.vz * c.vz);
	const double Eint_cons = r.E - kinetic_energy;
	if (re) {
		// the primitive is e = Eint / rho; the pressure is eos.pressure(rho, rho * e)
		const double e = divBy(Eint_cons, R);
		const double e2 = (c.rho == 0.0) ? 0.0 : divBy(c.rho * e, R);
		c.P = eos.gm1 * c.rho * e2;
	} else {
		const double e = (c.rho == 0.0) ? 0.0 : divBy(Eint_cons, R);
		c.P = eos.gm1 * c.rho * e;
	}
	return c;
}

// The flattening coefficient of a cell that IS compressive along the direction (v(+1) < v(-1)); it is 1 elsewhere (hydro_system.hpp:620-622), which
// the callers decide first — in smooth or expanding flow (all of the ambient medium of a young blast) the whole evaluation, K_S = rho c_s^2 with its
// square root included, is skipped by every lane of a wave.  Same operations as flatteningChiKS (qk_device.hpp), the three quotients on refined
// reciprocals (recipOf / divBy: the same bits as `/` for normal-range operands): |dP| / K_S shares 1 / K_S between the directions, the division by
// beta_max - beta_min is by a constant; (Zmax - Z) / (Zmax - Zmin) is a division by 0.5, exact as a multiplication.
QK_DEV auto chiCompressive(double Pm2, double Pm1, double Pp1, double Pp2, Recip const &RKS, Recip const &Rbeta) -> double
{
	constexpr double beta_max = 0.85;
	constexpr double Zmax = 0.75;
	constexpr double Zmin = 0.25;
	const double dP1 = fabs(Pp1 - Pm1);
	const double beta_denom = fabs(Pp2 - Pm2);
	const double beta = (beta_denom != 0) ? divBy(dP1, recipOf(beta_denom)) : 0;
	const double chi_min = smax0(smin1(divBy(beta_max - beta, Rbeta)));
	const double Z = divBy(dP1, RKS);
	return smax(chi_min, smin1((Zmax - Z) / (Zmax - Zmin)));
}

// Workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2; tiles that are neighbours in y read each other's halo rows.  The
// linear id is remapped so that every XCD works through one contiguous run of tiles (x fastest, then y, then box / segment): the halo rows
// of a tile are then found in the L2 of the XCD that just read them for the tile before.
// ndim < 3: the fab has no ghost cells in the inactive dimensions — rows / planes outside it take the neutral state and the flattening coefficient of
// an inactive direction is 1 (FlattenShocks takes the minimum over the AMREX_SPACEDIM active directions only, hydro_system.hpp:655-669).
// QK_PRE_PREFETCH (A/B knob, default 0): the next plane's values requested one plane ahead — 1 in both stages, 2 only in the stage that reads
// primitives (whose conversion-free body needs 112 registers instead of 118).  Measured in round 5 (profiles/round5/ab6_pre_prefetch.txt): the 20
// registers of the two in-flight cells spill 24 / 80 bytes per lane under the 128-register cap: 0.39 -> 0.42 ms (2), -> 0.53 ms (1) per launch.
#ifndef QK_PRE_PREFETCH
#define QK_PRE_PREFETCH 0
#endif
#ifndef QK_MARCH_PREFETCH
#define QK_MARCH_PREFETCH 1
#endif
#ifndef QK_PRE_HALO_DIET
#define QK_PRE_HALO_DIET 1 // (A/B knob)
#endif
template <bool PRIM>
__global__ void __launch_bounds__(PT_THREADS, 4) k_pre3(const qk_box *boxes, const SGeom *geom, const qk_array4 *U_t, double *scratch, int64_t T, Eos eos, bool re, int nseg,
							    int xt, int yt, int ndim)
{
	constexpr bool prim_in = PRIM;
	constexpr bool PF = (QK_PRE_PREFETCH == 1) || (QK_PRE_PREFETCH == 2 && PRIM); // the next plane's values requested one plane ahead
	const unsigned nblk = gridDim.x, lin = blockIdx.x;
	const unsigned q8 = nblk / 8, r8 = nblk % 8, xcd = lin % 8, slot = lin / 8;
	const unsigned logical = (xcd < r8) ? xcd * (q8 + 1) + slot : r8 * (q8 + 1) + (xcd - r8) * q8 + slot;
	const int bix = static_cast<int>(logical % xt), biy = static_cast<int>((logical / xt) % yt), biz = static_cast<int>(logical / (xt * yt));
	// LDS planes, indexed [row + 3][column + 3] (P), [row][column + 2] (v_x), [row + 2][column] (v_y)
	// (Two copies alternating from plane to plane, which makes the third barrier of a plane unnecessary, were measured in round 4: +-0 at 256^3,
	// 7 % slower at 512^3 — twice the LDS per workgroup.)
	__shared__ double s_P[PT_Y + 6][PT_X + 6];
	__shared__ double s_vx[PT_Y][PT_X + 4], s_vy[PT_Y + 4][PT_X];
	__shared__ double s_cx[PT_Y][PT_X + 2], s_cy[PT_Y + 2][PT_X]; // chi_x at columns -1..PT_X, chi_y at rows -1..PT_Y

	const int b = biz / nseg;
	const int seg = biz - b * nseg;
	const qk_box bx = boxes[b];
	const SGeom g = geom[b];
	RA4 U(U_t[b]);
	const int t = threadIdx.x;
	const int x0 = bx.lo[0] - 1 + bix * PT_X; // tile origin (first rim cell of the box for tile 0)
	const int y0 = bx.lo[1] - 1 + biy * PT_Y;
	if (x0 > bx.hi[0] + 1 || y0 > bx.hi[1] + 1) {
		return; // uniform for the workgroup
	}
	// output planes lo-1 .. hi+1 of the box, this segment's share [zfirst, zlast]
	const int nall = bx.hi[2] - bx.lo[2] + 3;
	const int seglen = (nall + nseg - 1) / nseg;
	const int zfirst = bx.lo[2] - 1 + seg * seglen;
	const int zlast = min(zfirst + seglen - 1, bx.hi[2] + 1);
	if (zlast < zfirst) {
		return;
	}
	// fab bounds (valid + 4 in the active dimensions): conversions outside feed only outputs outside valid + 1
	const int ngy = (ndim >= 2) ? NG : 0, ngz = (ndim == 3) ? NG : 0;
	const int flo[2] = {bx.lo[0] - NG, bx.lo[1] - ngy}, fhi[2] = {bx.hi[0] + NG, bx.hi[1] + ngy};
	const int fzlo = bx.lo[2] - ngz, fzhi = bx.hi[2] + ngz;

	const bool own = t < PT_OWN;
	const int ty = own ? t / PT_X : 0;
	const int tx = own ? t - ty * PT_X : 0;
	const int oi = x0 + tx, oj = y0 + ty;
	const bool ownIn = own && oi <= fhi[0] && oj >= flo[1] && oj <= fhi[1];
	const bool ownOut = own && oi <= bx.hi[0] + 1 && oj <= bx.hi[1] + 1;
	// the halo cell of this thread (threads PT_THREADS - PT_HALO .. PT_THREADS - 1)
	const int h = t - (PT_THREADS - PT_HALO);
	int hx = 0, hy = 0;
	if (h >= 0) {
		if (h < PT_HALO_Y) {
			const int r = h / PT_X;
			hx = h - r * PT_X;
			hy = (r < 3) ? r - 3 : PT_Y + r - 3;
		} else {
			const int r = (h - PT_HALO_Y) / 6, c = (h - PT_HALO_Y) - r * 6;
			hy = r;
			hx = (c < 3) ? c - 3 : PT_X + c - 3;
		}
	}
	const int hi_ = x0 + hx, hj = y0 + hy;
	const bool haloIn = (h >= 0) && hi_ >= flo[0] && hi_ <= fhi[0] && hj >= flo[1] && hj <= fhi[1];
	// rim coefficients (chi_x of columns -1 | PT_X, chi_y of rows -1 | PT_Y) are evaluated by the thread that converted that halo cell
	const bool rimX = (h >= 0) && (hx == -1 || hx == PT_X);
	const bool rimY = (h >= 0) && (hy == -1 || hy == PT_Y);
	const int rx = hx, ry = hy;

	double Pz[5] = {1., 1., 1., 1., 1.}, vzw[5] = {0., 0., 0., 0., 0.}, rhoz[3] = {1., 1., 1.}, chz[3] = {1., 1., 1.};
	const Recip Rbeta = recipOf(0.85 - 0.75); // beta_max - beta_min (hydro_system.hpp:596-597, :609)
	double fm[4] = {1., 1., 1., 1.}, fdx[4] = {0., 0., 0., 0.}, fdy[4] = {0., 0., 0., 0.}; // in-plane results of planes k-3 .. k
	double *Sout = scratch + g.off;
	// (requesting the next plane's conserved values one plane ahead was measured: +20 registers, spills under the 128-register cap that two
	// workgroups per CU need, 15 % slower — the kernel is issue-bound, not latency-bound)
	const PlaneRaw neutral{1., 0., 0., 0., prim_in ? 1. : 1. / eos.gm1};
	// prim_in: the array holds (rho, v_x, v_y, v_z, P, E_int) — what planeCell would form (stage 2 after a stage 1 with prim_out)
	auto cellOf = [&](PlaneRaw const &r) -> PlaneCell { return prim_in ? PlaneCell{r.rho, r.mx, r.my, r.mz, r.E} : planeCell(eos, re, r); };
	int64_t uo = ownIn ? U.idx(oi, oj, zfirst - 3) : 0;
	int64_t uh = haloIn ? U.idx(hi_, hj, zfirst - 3) : 0;
	PlaneRaw nextOwn = neutral, nextHalo = neutral;
	if constexpr (PF) {
		const bool k0In = (zfirst - 3 >= fzlo) && (zfirst - 3 <= fzhi);
		nextOwn = (ownIn && k0In) ? planeLoad(U, uo) : neutral;
		nextHalo = (haloIn && k0In && (zfirst - 3 >= zfirst)) ? planeLoad(U, uh) : neutral;
	}

	for (int k = zfirst - 3; k <= zlast + 3; ++k, uo += U.ks, uh += U.ks) {
		const bool inPlane = (k >= zfirst) && (k <= zlast); // uniform: this plane's x / y results are somebody's output
		const bool kIn = (k >= fzlo) && (k <= fzhi);	      // uniform: the plane exists in the fab
		PlaneRaw rawOwn, rawHalo = neutral;
		if constexpr (PF) {
			rawOwn = nextOwn;
			rawHalo = nextHalo;
			const int kn = k + 1;
			const bool knIn = (kn >= fzlo) && (kn <= fzhi) && (kn <= zlast + 3);
			nextOwn = (ownIn && knIn) ? planeLoad(U, uo + U.ks) : neutral;
			nextHalo = (haloIn && knIn && (kn >= zfirst) && (kn <= zlast)) ? planeLoad(U, uh + U.ks) : neutral;
		} else {
			rawOwn = (ownIn && kIn) ? planeLoad(U, uo) : neutral;
		}
		const PlaneCell c = cellOf(rawOwn);
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			Pz[m] = Pz[m + 1];
			vzw[m] = vzw[m + 1];
		}
		Pz[4] = c.P;
		vzw[4] = c.vz;
		// (rho c_s^2 of a cell — the denominator of the shock-strength ratio, the same for chi_x, chi_y in this plane and chi_z two planes on — is
		// evaluated only where a direction is compressive: the z window keeps rho, not K_S)
		rhoz[0] = rhoz[1];
		rhoz[1] = rhoz[2];
		rhoz[2] = c.rho;
#pragma unroll
		for (int m = 0; m < 3; ++m) {
			fm[m] = fm[m + 1];
			fdx[m] = fdx[m + 1];
			fdy[m] = fdy[m + 1];
		}
		if (inPlane) {
			PlaneCell hc{1., 0., 0., 0., 1.};
			if (own) {
				s_P[ty + 3][tx + 3] = c.P;
				s_vx[ty][tx + 2] = c.vx;
				s_vy[ty + 2][tx] = c.vy;
			}
			if (h >= 0) {
				if constexpr (PRIM && !PF && QK_PRE_HALO_DIET != 0) {
					// a halo cell of the primitive form is needed for its pressure; its v_x only in the tile's rows within two columns of the tile,
					// its v_y only in the tile's columns within two rows, its density only on the rim (the shock-strength ratio of chi there):
					// 2.3 loads per halo cell instead of 5 (the conserved form needs all five to form the pressure)
					if (haloIn && kIn) {
						hc.P = U.p[uh + U.ns * ENE];
						if (hy >= 0 && hy < PT_Y && hx >= -2 && hx < PT_X + 2) {
							hc.vx = U.p[uh + U.ns * MX];
						}
						if (hx >= 0 && hx < PT_X && hy >= -2 && hy < PT_Y + 2) {
							hc.vy = U.p[uh + U.ns * MY];
						}
						if (rimX || rimY) {
							hc.rho = U.p[uh + U.ns * RHO];
						}
					}
				} else {
					hc = cellOf(PF ? rawHalo : ((haloIn && kIn) ? planeLoad(U, uh) : neutral));
				}
				s_P[hy + 3][hx + 3] = hc.P;
				if (hy >= 0 && hy < PT_Y && hx >= -2 && hx < PT_X + 2) {
					s_vx[hy][hx + 2] = hc.vx;
				}
				if (hx >= 0 && hx < PT_X && hy >= -2 && hy < PT_Y + 2) {
					s_vy[hy + 2][hx] = hc.vy;
				}
			}
			__syncthreads();
			if (own) {
				const bool cx = s_vx[ty][tx + 3] < s_vx[ty][tx + 1];
				const bool cy = (ndim >= 2) && (s_vy[ty + 3][tx] < s_vy[ty + 1][tx]);
				double chx = 1.0, chy = 1.0;
				if (cx || cy) {
					const Recip RKS = recipOf(flatteningKS(eos, c.rho, c.P));
					if (cx) {
						const double *Pr = &s_P[ty + 3][tx + 3];
						chx = chiCompressive(Pr[-2], Pr[-1], Pr[1], Pr[2], RKS, Rbeta);
					}
					if (cy) {
						chy = chiCompressive(s_P[ty + 1][tx + 3], s_P[ty + 2][tx + 3], s_P[ty + 4][tx + 3], s_P[ty + 5][tx + 3], RKS, Rbeta);
					}
				}
				s_cx[ty][tx + 1] = chx;
				s_cy[ty + 1][tx] = chy;
			}
			if (rimX) {
				double chx = 1.0;
				if (s_vx[ry][rx + 3] < s_vx[ry][rx + 1]) {
					const double *Pr = &s_P[ry + 3][rx + 3];
					chx = chiCompressive(Pr[-2], Pr[-1], Pr[1], Pr[2], recipOf(flatteningKS(eos, hc.rho, hc.P)), Rbeta);
				}
				s_cx[ry][rx + 1] = chx;
			} else if (rimY) {
				double chy = 1.0;
				if ((ndim >= 2) && (s_vy[ry + 3][rx] < s_vy[ry + 1][rx])) {
					chy = chiCompressive(s_P[ry + 1][rx + 3], s_P[ry + 2][rx + 3], s_P[ry + 4][rx + 3], s_P[ry + 5][rx + 3],
							     recipOf(flatteningKS(eos, hc.rho, hc.P)), Rbeta);
				}
				s_cy[ry + 1][rx] = chy;
			}
			__syncthreads();
			if (own) {
				const double mx = smin(smin(s_cx[ty][tx], s_cx[ty][tx + 1]), s_cx[ty][tx + 2]);
				const double my = smin(smin(s_cy[ty][tx], s_cy[ty + 1][tx]), s_cy[ty + 2][tx]);
				fm[3] = smin(mx, my);
				fdx[3] = smin(s_vx[ty][tx + 3] - c.vx, c.vx - s_vx[ty][tx + 1]);
				fdy[3] = smin(s_vy[ty + 3][tx] - c.vy, c.vy - s_vy[ty + 1][tx]);
			}
			__syncthreads(); // the next plane overwrites the LDS planes
		}
		// z direction: chi_z of plane k-2 from P(k-4, k-3, k-1, k), rho c_s^2 (k-2), v_z(k-3), v_z(k-1)
		chz[0] = chz[1];
		chz[1] = chz[2];
		chz[2] = 1.0;
		if ((ndim == 3) && (vzw[3] < vzw[1])) {
			chz[2] = chiCompressive(Pz[0], Pz[1], Pz[3], Pz[4], recipOf(flatteningKS(eos, rhoz[0], Pz[2])), Rbeta);
		}
		const int ko = k - 3;
		if (ko >= zfirst && ko <= zlast && ownOut) {
			const int64_t cc = (oi - g.glo[0]) + static_cast<int64_t>(g.pitch) * ((oj - g.glo[1]) + static_cast<int64_t>(g.n[1]) * (ko - g.glo[2]));
			Sout[(S_AUX + 0) * T + cc] = smin(fm[0], smin(smin(chz[0], chz[1]), chz[2]));
			Sout[(S_AUX + 1) * T + cc] = fdx[0];
			Sout[(S_AUX + 2) * T + cc] = fdy[0];
			// D_z of plane ko = k-3: v_z(k-2) - v_z(k-3), v_z(k-3) - v_z(k-4)
			Sout[(S_AUX + 3) * T + cc] = smin(vzw[2] - vzw[1], vzw[1] - vzw[0]);
		}
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

// The carried form on a level that has refined children (qk_hydro_stage_args::flux_mask): the flux registers of the hierarchy need flux_rk2 = 0.5 F1 +
// 0.5 F2 on the coarse-fine faces — a few thousand faces of 50 million.  A face with a marked cell on either side keeps F1 in halfFlux (stage 1) and
// gets flux_rk2 written to rk2Flux (stage 2), exactly the values the reference's form stores on EVERY face; the cell updates stay carried.
using CA4 = A4<const char, qk_carray4>;
// The mask descriptor of a box may describe a WINDOW of the box only — the bounding box of its marked cells, the same memory and strides with begin /
// end cropped (an empty window: no cell) —: cells outside [begin, end) are unmarked and their bytes are never read.  The marked cells of a level are the
// two cell layers along the coarse-fine interfaces; with whole-box descriptors every face of the level waited for two byte loads before its
// flux could be dropped (the sweeps of BASELINE config 5's base level: X +9 %, Y +25 %, Z +17 % against the unigrid level).
QK_DEV auto maskByte(qk_carray4 const &d, int i, int j, int k) -> int
{
	if (i < d.begin[0] || i >= d.end[0] || j < d.begin[1] || j >= d.end[1] || k < d.begin[2] || k >= d.end[2]) {
		return 0;
	}
	return CA4(d)(i, j, k);
}
template <int STAGE, int NV> QK_DEV void maskedFaceFlux(SweepArgs const &a, int b, int i, int j, int k, const double F[NV])
{
	if (STAGE == 1) {
		WA4 HF(a.halfFlux[b]);
		const int64_t o = HF.idx(i, j, k);
#pragma unroll
		for (int n = 0; n < NV; ++n) {
			HF.p[o + HF.ns * n] = F[n];
		}
	} else {
		RA4 HF(a.halfFlux[b]);
		WA4 RF(a.rk2Flux[b]);
		const int64_t o = HF.idx(i, j, k), o2 = RF.idx(i, j, k);
#pragma unroll
		for (int n = 0; n < NV; ++n) {
			RF.p[o2 + RF.ns * n] = 0.5 * HF.p[o + HF.ns * n] + 0.5 * F[n]; // (0 + 0.5 F1) + 0.5 F2 (QuokkaSimulation.hpp:1106, :1220)
		}
	}
}

// ---------------------------------------------------------------------------------------------- first-order flux correction, fused
// A stage whose first pass flagged cells (PredictStep left rho <= 0 somewhere) is repeated by the SAME kernels with FOFC = true (qk_hydro_stage_args::
// fofc_pass): a face that touches a flagged cell takes the first-order flux of the OLD state — donor-cell states + LLF, computeFOHydroFluxes
// (QuokkaSimulation.hpp:1520-1568) evaluated on demand for just those faces — instead of the high-order one (replaceFluxes, :1324-1368; in stage 2 it
// replaces flux_rk2 as a whole); a flagged cell takes the cell-centred velocity divergence in its P dV term (hydro_system.hpp:804-808); the pass
// counts the cells that are still invalid and writes no flags (they are its input).  Same device functions as the reference-shaped operators: the
// result equals the operator path bit for bit.  Not instantiated for the carried-rhs form (its stage 2 has no F1 to average at the other faces) and
// only taken for K_visc == 0 (the viscosity term of a first-order face needs the transverse differences of the old state).
template <int NS> QK_DEV void primOfCell(RA4 const &U, Eos const &eos, bool re, int i, int j, int k, double q[NVAR + NS])
{
	const int64_t u = U.idx(i, j, k);
	double Uc[NVAR];
#pragma unroll
	for (int n = 0; n < NVAR; ++n) {
		Uc[n] = U.p[u + U.ns * n];
	}
	consToPrim(eos, re, Uc, q);
#pragma unroll
	for (int n = NVAR; n < NVAR + NS; ++n) {
		q[n] = U.p[u + U.ns * n];
	}
}
template <int DIR, int NS, bool TWOD, int NDIM>
QK_DEV void firstOrderFlux(Eos const &eos, bool re, const double qL[NVAR + NS], const double qR[NVAR + NS], double F[NVAR + NS], double &vf)
{
	Wave wv;
	faceFlux<DIR, QK_RIEMANN_LLF, TWOD>(eos, re, NDIM, qL, qR, 0., 0., 0., 0., 0., 0., F, vf, NS > 0 ? &wv : nullptr);
#pragma unroll
	for (int n = NVAR; n < NVAR + NS; ++n) {
		F[n] = scalarFlux<QK_RIEMANN_LLF>(wv, qL[n], qR[n]);
	}
}
// hydro_system.hpp:804-808: 0.5 * sum_d (v_d(+1) - v_d(-1)) / dx_d from the conserved variables (ComputeVelocityX1..3)
template <int NDIM> QK_DEV auto cellCentredDivV(RA4 const &U, int i, int j, int k, const double dx[3]) -> double
{
	auto vel = [&](int ii, int jj, int kk, int d) { return U(ii, jj, kk, MX + d) / U(ii, jj, kk, RHO); };
	double s = (vel(i + 1, j, k, 0) - vel(i - 1, j, k, 0)) / dx[0];
	if (NDIM >= 2) {
		s = s + (vel(i, j + 1, k, 1) - vel(i, j - 1, k, 1)) / dx[1];
	}
	if (NDIM == 3) {
		s = s + (vel(i, j, k + 1, 2) - vel(i, j, k - 1, 2)) / dx[2];
	}
	return 0.5 * s;
}

// epilogue of one cell (AddInternalEnergyPdV + PredictStep + EnforceLimits + SyncDualEnergy + the two CFL signal speeds), same operations
// as the per-cell functions of qk_device.hpp (which the reference-shaped operators call) with the ~30 divisions grouped by denominator:
// rho_old x4, rho_new x12, 2 rho_new x1, k_B and k_B_user (constants: their reciprocals are refined once per thread, EpiConst) — every further
// quotient is mul + 2 fma on the refined reciprocal, the same bits as `/` for normal-range operands (recipOf in qk_device.hpp).
struct EpiConst {
	Recip RkB, RkBu, Rmu; // 1 / k_B, 1 / k_B_user, 1 / (mu m_u)
};
QK_DEV auto epiConst(Eos const &eos) -> EpiConst
{
	EpiConst c;
	c.RkB = eos.RkB; // (formed on the host: qk::Eos)
	c.RkBu = eos.RkBu;
	c.Rmu = eos.Rmu;
	return c;
}
// Eos::tgasFromEint / eintFromTgas (EOS.hpp:74-159) with the shared reciprocals
QK_DEV auto epiTgas(Eos const &eos, EpiConst const &ec, Recip const &Rrho, double Eint) -> double
{
	if (eos.tmodel == 1) {
		return sqrt(sqrt(4.0 * Eint / eos.alpha));
	}
	const double e = divBy(Eint, Rrho);
	const double T = divBy(e * eos.mu * Eos::m_u * eos.gm1, ec.RkB);
	return divBy(T * Eos::k_B, ec.RkBu);
}
QK_DEV auto epiEint(Eos const &eos, EpiConst const &ec, double rho, double T) -> double
{
	if (eos.tmodel == 1) {
		return (eos.alpha / 4.0) * ((T * T) * (T * T));
	}
	const double p = divBy(rho * T * Eos::k_B, ec.Rmu);
	const double e = p / (eos.gm1 * rho);
	return divBy(e * rho * eos.kB_user, ec.RkB);
}

// U holds the old state of the cell on entry (CS == 2: not read — the carried half step replaces it)
// CS (carry stage): 0 the reference's flux average (rhs already holds div flux_rk2); 1: stage 1 of the carried form — besides its own update the cell
// stores S = U_old + (dt/2) r_1 and P(U_old) in `a.rhs1`; 2: stage 2 — `half` holds what stage 1 stored and the update is S + (dt/2) r_2
template <int NS, int CS, bool FOFC = false, int NDIM = 3>
QK_DEV void updateCellFrom(SweepArgs const &a, Eos const &eos, EpiConst const &ec, int b, int i, int j, int k, double U[NVAR + NS], const double rhs_sweeps[NVAR + NS],
			   double div_v, const double half[NVAR + NS + 1], double &sig0, double &sig1, bool haveP = false, double Pknown = 0.)
{
	WA4 Un(a.U_out[b]);
	IA4 flag(a.redoFlag[b]);
	if constexpr (FOFC) {
		if (flag(i, j, k) != 0) { // flagged by the first pass
			div_v = cellCentredDivV<NDIM>(RA4(a.U_old[b]), i, j, k, a.dx3);
		}
	}
	double rhs[NVAR + NS];
#pragma unroll
	for (int n = 0; n < NVAR + NS; ++n) {
		rhs[n] = rhs_sweeps[n];
	}
	// hydro_system.hpp:797-812 (redoFlag == none branch): P(U_old) = ComputePressure(cons)
	double Pgas;
	if (CS == 2) {
		Pgas = half[NVAR + NS];
	} else if (haveP) {
		// (uniform) the old state is the sweep's input state and the marching window holds this cell's primitives: its pressure there is
		// ComputePressure of these conserved values, formed by the same operations on the same operands (consToPrim, reconstruct_eint off)
		Pgas = Pknown;
	} else {
		const double rho = U[RHO];
		if (eos.isothermal) {
			Pgas = rho * eos.cs_iso * eos.cs_iso;
		} else {
			const Recip Ro = recipOf(rho);
			const double vx = divBy(U[MX], Ro), vy = divBy(U[MY], Ro), vz = divBy(U[MZ], Ro);
			const double thermal_energy = U[ENE] - 0.5 * rho * (vx * vx + vy * vy + vz * vz);
			const double e = (rho == 0.0) ? 0.0 : divBy(thermal_energy, Ro);
			Pgas = eos.gm1 * rho * e;
		}
	}
	double r[NVAR];
#pragma unroll
	for (int n = 0; n < NVAR; ++n) {
		r[n] = rhs[n];
	}
	r[EINT] += -Pgas * div_v;
	if (CS == 1) {
		WA4 R1(a.rhs1[b]);
		const int64_t c1 = R1.idx(i, j, k);
		const double hdt = 0.5 * a.dt;
#pragma unroll
		for (int n = 0; n < NVAR + NS; ++n) {
			streamStore(&R1.p[c1 + R1.ns * n], U[n] + hdt * ((n < NVAR) ? r[n] : rhs[n]));
		}
		streamStore(&R1.p[c1 + R1.ns * (NVAR + NS)], Pgas);
	}
	if (CS == 2) {
		const double hdt = 0.5 * a.dt;
#pragma unroll
		for (int n = 0; n < NVAR + NS; ++n) {
			U[n] = half[n] + hdt * ((n < NVAR) ? r[n] : rhs[n]);
		}
	} else {
		// hydro_system.hpp:487-495
#pragma unroll
		for (int n = 0; n < NVAR; ++n) {
			U[n] = U[n] + a.dt * r[n];
		}
#pragma unroll
		for (int n = NVAR; n < NVAR + NS; ++n) { // passive scalars: PredictStep loops over all hydro variables
			U[n] = U[n] + a.dt * rhs[n];
		}
	}
	const int bad = (U[RHO] > 0.) ? 0 : 1;
	if constexpr (!FOFC) {
		flag(i, j, k) = bad;
	}
	if (bad != 0) {
		atomicAdd(a.redo_count, 1ULL);
	}
	// A cell the FIRST pass leaves invalid is recomputed (correction pass or retry): its limits are skipped.  The correction pass is the stage's
	// last word: with abort_on_fofc_failure = 0 the reference goes on to EnforceLimits + SyncDualEnergy on EVERY cell (QuokkaSimulation.hpp:1180-1192,
	// :1267-1279) — the density floor is what repairs a cell that is still at rho <= 0 — and SyncDualEnergy aborts where it is not repaired
	// (hydro_system.hpp:834-836: the error flag here).
	const bool fix = FOFC || (bad == 0);
	// EnforceLimits (hydro_system.hpp:702-771) : density floor, then the temperature floors on E and on the auxiliary internal energy
	double rho_new = U[RHO];
	if (fix && U[RHO] < a.densityFloor) {
#pragma unroll
		for (int n = NVAR; n < NVAR + NS; ++n) { // hydro_system.hpp:713-722 (as the reference-shaped operator, qk_hydro_EnforceLimits)
			U[n] = (a.densityFloor == 0.0) ? 0.0 : U[n] * (U[RHO] / a.densityFloor);
		}
		rho_new = a.densityFloor;
		U[RHO] = rho_new;
	}
	const Recip Rn = recipOf(rho_new);
	const double px = U[MX], py = U[MY], pz = U[MZ];
	const double vx = divBy(px, Rn), vy = divBy(py, Rn), vz = divBy(pz, Rn);
	if (fix) {
		if ((rho_new > 2.2250738585072014e-308) && !eos.isothermal) {
			const double Ekin = 0.5 * rho_new * (vx * vx + vy * vy + vz * vz);
			const double Etot = U[ENE];
			if (epiTgas(eos, ec, Rn, Etot - Ekin) < a.tempFloor) {
				U[ENE] = Ekin + epiEint(eos, ec, rho_new, a.tempFloor);
			}
			if (epiTgas(eos, ec, Rn, U[EINT]) < a.tempFloor) {
				U[EINT] = epiEint(eos, ec, rho_new, a.tempFloor);
			}
		}
	}
	// SyncDualEnergy (:825-849)
	// (a / (2 rho) == 0.5 (a / rho): scaling by two is exact)
	const double Ekin2 = 0.5 * divBy(px * px + py * py + pz * pz, Rn);
	if constexpr (FOFC) {
		if (a.use_dual_energy != 0 && !(rho_new > 0.)) {
			*a.error_flag = 1; // "density is negative in SyncDualEnergy! abort!!"
		}
	}
	if (fix && a.use_dual_energy != 0 && (!FOFC || rho_new > 0.)) {
		const double Etot = U[ENE];
		const double Eint_aux = U[EINT];
		const double Eint_cons = Etot - Ekin2;
		if (Eint_cons > 1.0e-3 * Etot) {
			U[EINT] = Eint_cons;
		} else {
			U[EINT] = Eint_aux;
			U[ENE] = Eint_aux + Ekin2;
		}
	}
	const int64_t cn = Un.idx(i, j, k);
	if (a.prim_out) {
		// consToPrim of the state this cell would have stored: rho_new, the three quotients on its reciprocal formed above, the pressure of the
		// final energy (gamma law, reconstruct_eint off: qk_hydro_stage_fused checks); the auxiliary internal energy and the scalars as they are
		const double Eint_cons = U[ENE] - 0.5 * rho_new * (vx * vx + vy * vy + vz * vz);
		const double e = (rho_new == 0.0) ? 0.0 : divBy(Eint_cons, Rn);
		streamStore(&Un.p[cn + Un.ns * PRHO], rho_new);
		streamStore(&Un.p[cn + Un.ns * PVX], vx);
		streamStore(&Un.p[cn + Un.ns * PVY], vy);
		streamStore(&Un.p[cn + Un.ns * PVZ], vz);
		streamStore(&Un.p[cn + Un.ns * PPRES], eos.gm1 * rho_new * e);
		streamStore(&Un.p[cn + Un.ns * PEINT], U[EINT]);
#pragma unroll
		for (int n = NVAR; n < NVAR + NS; ++n) {
			streamStore(&Un.p[cn + Un.ns * n], U[n]);
		}
	} else {
#pragma unroll
		for (int n = 0; n < NVAR + NS; ++n) {
			streamStore(&Un.p[cn + Un.ns * n], U[n]);
		}
	}
	if (a.max_signal != nullptr) {
		// maxSignalSpeedLocal (:206-219) and ComputeMaxSignalSpeed (:227-250) of the new state
		double cs;
		if (eos.isothermal) {
			cs = eos.cs_iso;
		} else {
			const double thermal_energy = U[ENE] - 0.5 * rho_new * (vx * vx + vy * vy + vz * vz);
			const double e = (rho_new == 0.0) ? 0.0 : divBy(thermal_energy, Rn);
			const double P = eos.gm1 * rho_new * e;
			cs = sqrtN(divBy(eos.gamma * P, Rn));
		}
		sig0 = smax(sig0, cs + sqrtN(divBy(2.0 * Ekin2, Rn)));
		sig1 = smax(sig1, cs + sqrtN(vx * vx + vy * vy + vz * vz));
	}
}

// ---------------------------------------------------------------------------------------------- X sweep (flat + LDS)
#ifndef QK_XB
#define QK_XB 256 // (A/B knob)
#endif
constexpr int XB = QK_XB;	 // threads per workgroup
constexpr int XOUT = XB - 6; // cells updated per workgroup (3 halo cells on each side)

// NDIM: AMREX_SPACEDIM of the build.  In a 1-D build the x sweep is the only one and carries the epilogue (P dV, PredictStep, flags, limits, dual
// energy, CFL maxima) the z sweep carries in 3-D; in 2-D the y sweep does (k_sweep_march<1, ..., LAST, TWOD>).
template <int ORDER, int STAGE, int NS, bool CARRY, int NDIM = 3, bool FOFC = false> __global__ void __launch_bounds__(XB) k_sweep_x(SweepArgs a, Eos eos)
{
	static_assert(!(FOFC && CARRY), "the first-order flux correction pass exists for the reference's form of the RK2 average");
	constexpr int NV = NVAR + NS; // hydro variables + passive scalars
	constexpr int RHS_DIVV = S_RHS + NV;
	__shared__ double s_q[NV][XB];  // primitives, later reused for the face fluxes
	__shared__ double s_e[NV][XB];  // right-edge states a_plus
	__shared__ double s_d[3][XB];	  // D_V, D_W, face velocity

	// (no XCD-contiguous remap here — qk_device.hpp: measured 2 % slower; the kernel is bound by FP64 issue, not by the halo lines it re-fetches)
	const int b = blockIdx.z;
	const qk_box bx = a.boxes[b];
	const SGeom g = a.geom[b];
	const int t = threadIdx.x;
	const int k = bx.lo[2] + blockIdx.y;
	if (k > bx.hi[2]) {
		return; // uniform for the whole workgroup
	}
	// flat slab: rows j = lo.y .. hi.y of plane k are contiguous
	// (flat over the n[0] cells of a row — the pad columns of an aligned row pitch are skipped; the scratch index goes through the pitch)
	const int64_t rowlen = g.n[0];
	const int64_t slablen = rowlen * (bx.hi[1] - bx.lo[1] + 1);
	const int64_t f = static_cast<int64_t>(blockIdx.x) * XOUT + t - 3; // flat position inside the slab
	const bool inside = (f >= 0) && (f < slablen);
	const int jj = static_cast<int>((inside ? f : 0) / rowlen);
	const int i = g.glo[0] + static_cast<int>((inside ? f : 0) - jj * rowlen);
	const int j = bx.lo[1] + jj;
	const int64_t c = (i - g.glo[0]) + static_cast<int64_t>(g.pitch) * ((j - g.glo[1]) + static_cast<int64_t>(g.n[1]) * (k - g.glo[2]));

	const double *S = a.scratch + g.off;
	const int64_t T = a.total_cells;
	double q0[NV];
	{
		RA4 U(a.U_in[b]);
		const int64_t u = U.idx(i, j, k);
		double Uc[NVAR];
#pragma unroll
		for (int n = 0; n < NVAR; ++n) {
			Uc[n] = U.p[u + U.ns * n];
		}
		if (a.prim_in) { // (uniform) the input array holds the primitives
#pragma unroll
			for (int n = 0; n < NVAR; ++n) {
				q0[n] = Uc[n];
			}
		} else {
			consToPrim(eos, a.reconstruct_eint, Uc, q0);
		}
#pragma unroll
		for (int n = NVAR; n < NV; ++n) { // hydro_system.hpp:340-343: passive scalars are reconstructed as they are stored
			q0[n] = U.p[u + U.ns * n];
		}
	}
#pragma unroll
	for (int n = 0; n < NV; ++n) {
		s_q[n][t] = q0[n];
	}
	const double chi = S[(S_AUX + 0) * T + c];
	const double dV = S[(S_AUX + 2) * T + c]; // view-j axis of X1 is y
	const double dW = S[(S_AUX + 3) * T + c]; // view-k axis is z
	s_d[0][t] = dV;
	s_d[1][t] = dW;
	__syncthreads();

	// reconstruct my cell (needs t-2 .. t+2)
	double am[NV], ap[NV];
	const int tm2 = max(t - 2, 0), tm1 = max(t - 1, 0), tp1 = min(t + 1, XB - 1), tp2 = min(t + 2, XB - 1);
#pragma unroll
	for (int n = 0; n < NV; ++n) {
		cellEdges<ORDER>(s_q[n][tm2], s_q[n][tm1], q0[n], s_q[n][tp1], s_q[n][tp2], am[n], ap[n]);
		flattenEdges(chi, q0[n], am[n], ap[n]);
		s_e[n][t] = ap[n];
	}
	__syncthreads();

	// flux at my left face
	double qL[NV];
#pragma unroll
	for (int n = 0; n < NV; ++n) {
		qL[n] = s_e[n][tm1];
	}
	const double du = q0[PVX] - s_q[PVX][tm1];
	const double dvl = s_d[0][tm1], dwl = s_d[1][tm1];
	double F[NV], vf;
	{
		Wave wv;
		faceFlux<0, QK_RIEMANN_HLLC>(eos, a.reconstruct_eint, NDIM, qL, am, du, dvl, dV, dwl, dW, a.K_visc, F, vf, NS > 0 ? &wv : nullptr);
#pragma unroll
		for (int n = NVAR; n < NV; ++n) { // hydro_system.hpp:1062-1076, HLLC.hpp:126-136
			F[n] = scalarFlux<QK_RIEMANN_HLLC>(wv, qL[n], am[n]);
		}
	}

	const bool validRow = inside;
	const bool isFace = validRow && (i >= bx.lo[0]) && (i <= bx.hi[0] + 1) && (t >= 3) && (t <= XB - 3);
	// FOFC pass: does this face touch a cell the first pass flagged? (redoFlag carries one filled ghost cell)
	bool firstOrder = false;
	if constexpr (FOFC) {
		if (isFace) {
			CIA4 flag(a.redoFlag[b]);
			firstOrder = (flag(i - 1, j, k) != 0) || (flag(i, j, k) != 0);
		}
	}
	auto replaceByFirstOrder = [&]() { // replaceFluxes (QuokkaSimulation.hpp:1324-1368) for this face, the first-order flux evaluated on demand
		double qLo[NV], qRo[NV];
		if (STAGE == 1) { // the old state is the input state: its primitives are in LDS
#pragma unroll
			for (int n = 0; n < NV; ++n) {
				qLo[n] = s_q[n][tm1];
				qRo[n] = q0[n];
			}
		} else {
			RA4 Uold(a.U_old[b]);
			primOfCell<NS>(Uold, eos, a.reconstruct_eint, i - 1, j, k, qLo);
			primOfCell<NS>(Uold, eos, a.reconstruct_eint, i, j, k, qRo);
		}
		firstOrderFlux<0, NS, false, NDIM>(eos, a.reconstruct_eint, qLo, qRo, F, vf);
	};
	if (CARRY) {
		// carried right-hand side: neither stage touches the face arrays — except on the marked faces of a level with refined children
		if (a.fluxMask != nullptr && isFace) {
			const qk_carray4 md = a.fluxMask[b];
			if ((maskByte(md, i - 1, j, k) | maskByte(md, i, j, k)) != 0) {
				maskedFaceFlux<STAGE, NV>(a, b, i, j, k, F);
			}
		}
	} else if (STAGE == 1 && FOFC) {
		// (halfFlux keeps the UNCORRECTED stage-1 flux the first pass stored: flux_rk2 is formed from it, QuokkaSimulation.hpp:1105-1108)
		if (firstOrder) {
			replaceByFirstOrder();
		}
	} else if (STAGE == 1) {
		if (isFace) {
			WA4 HF(a.halfFlux[b]);
			WA4 HV(a.halfVel[b]);
			const int64_t o = HF.idx(i, j, k);
#pragma unroll
			for (int n = 0; n < NV; ++n) {
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
			for (int n = 0; n < NV; ++n) {
				F[n] = 0.5 * HF.p[o + HF.ns * n] + 0.5 * F[n];
			}
			vf = 0.5 * HV(i, j, k) + 0.5 * vf;
			if (FOFC && firstOrder) {
				replaceByFirstOrder(); // flux_rk2 of this face as a whole
			}
			if (a.store_rk2) {
				WA4 RF(a.rk2Flux[b]);
				const int64_t o2 = RF.idx(i, j, k);
#pragma unroll
				for (int n = 0; n < NV; ++n) {
					RF.p[o2 + RF.ns * n] = F[n];
				}
			}
		}
	}
	__syncthreads(); // everyone is done with s_q (primitives)
#pragma unroll
	for (int n = 0; n < NV; ++n) {
		s_q[n][t] = F[n];
	}
	s_d[2][t] = vf;
	__syncthreads();

	const bool isCell = validRow && (i >= bx.lo[0]) && (i <= bx.hi[0]) && (t >= 3) && (t < 3 + XOUT);
	if constexpr (NDIM == 1) {
		// the only sweep of a 1-D build: finish the cell here (same per-cell function as the z sweep of a 3-D build)
		double sig0 = 0., sig1 = 0.;
		if (isCell) {
			double rhs[NV], r1[NV + 1];
#pragma unroll
			for (int n = 0; n < NV; ++n) {
				rhs[n] = a.inv_dx * (F[n] - s_q[n][tp1]); // hydro_system.hpp:469
				r1[n] = 0.;
			}
			r1[NV] = 0.;
			const double div_v = (s_d[2][tp1] - vf) / a.dx; // :803
			RA4 Uold(a.U_old[b]);
			const int64_t co = Uold.idx(i, j, k);
			double Uo[NV];
#pragma unroll
			for (int n = 0; n < NV; ++n) {
				Uo[n] = Uold.p[co + Uold.ns * n];
			}
			if (CARRY && STAGE == 2) {
				RA4 R1(a.rhs1[b]);
				const int64_t c1 = R1.idx(i, j, k);
#pragma unroll
				for (int n = 0; n < NV + 1; ++n) {
					r1[n] = R1.p[c1 + R1.ns * n];
				}
			}
			const EpiConst ec = epiConst(eos);
			updateCellFrom<NS, CARRY ? STAGE : 0, FOFC, 1>(a, eos, ec, b, i, j, k, Uo, rhs, div_v, r1, sig0, sig1);
		}
		if (a.max_signal != nullptr) { // every lane takes part in the wave reduction (no early exit above for lanes of a live workgroup)
			for (int off = 32; off > 0; off >>= 1) {
				sig0 = smax(sig0, __shfl_xor(sig0, off));
				sig1 = smax(sig1, __shfl_xor(sig1, off));
			}
			if ((threadIdx.x & 63) == 0) {
				atomicMaxNonNeg(&a.max_signal[0], sig0);
				atomicMaxNonNeg(&a.max_signal[1], sig1);
			}
		}
	} else if (isCell) {
		double *R = a.scratch + g.off + c;
#pragma unroll
		for (int n = 0; n < NV; ++n) {
			// hydro_system.hpp:469
			streamStore(&R[(S_RHS + n) * T], a.inv_dx * (F[n] - s_q[n][tp1]));
		}
		// hydro_system.hpp:803
		R[RHS_DIVV * T] = (s_d[2][tp1] - vf) / a.dx;
	}
}

// ---------------------------------------------------------------------------------------------- Y / Z sweeps (marching)
#ifndef QK_MARCH_BY
#define QK_MARCH_BY 4
#endif
constexpr int MARCH_BY = QK_MARCH_BY; // rows of the other transverse axis per workgroup (64 x MARCH_BY threads)
// TWOD: the y sweep of an AMREX_SPACEDIM == 2 build — the X2 view is the index swap of ArrayView_2d.hpp (view-j = x, view-k = z), and it is the last sweep
// FUSEX (the Y sweep of a 3-D level, carried form without flux mask): the X sweep folded into the march.  The lanes of a wave run along x and hold the
// primitives of their cells of the row the step completes: the +-2 stencil of the reconstruction, the edge state of the left neighbour and the flux of
// the right face are in the neighbouring lanes and travel through a few hundred bytes of wave-private LDS (no barrier: a wave's LDS operations
// are executed in order; the fences only keep the compiler from moving a read above the write it depends on).  The X flux divergence starts the
// accumulator in registers: no X launch, no 72 + 56 B per cell read and written by it, no 56 B read back by Y.
constexpr int XW = 64 + 8; // slots of a row buffer: 64 cells + 2 halo cells either side (+ padding)
#ifndef QK_XROWS
#define QK_XROWS 16
#endif
constexpr int XROWS = QK_XROWS; // rows per batch of wave-edge faces (2 * XROWS lanes of the wave work in a batch pass)
static_assert(XROWS == 8 || XROWS == 16 || XROWS == 32, "the batch maps rows and sides onto the lanes of one wave");
QK_DEV void waveFence()
{
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	__builtin_amdgcn_wave_barrier();
}
template <int DIR, int ORDER, int STAGE, bool LAST, int NS, bool CARRY, bool TWOD = false, bool FOFC = false, bool FUSEX = false>
__global__ void __launch_bounds__(64 * MARCH_BY) k_sweep_march(SweepArgs a, Eos eos)
{
	static_assert(!FUSEX || (DIR == 1 && !LAST && !TWOD && !FOFC && CARRY), "the X sweep rides on the Y sweep of a 3-D level in the carried form");
	static_assert(!(FOFC && CARRY), "the first-order flux correction pass exists for the reference's form of the RK2 average");
	static_assert(!TWOD || (DIR == 1 && LAST), "the 2-D build has one marching sweep: y, carrying the epilogue");
	constexpr int NV = NVAR + NS; // hydro variables + passive scalars
	constexpr int RHS_DIVV = S_RHS + NV;
	static_assert(DIR == 1 || DIR == 2, "marching sweeps are the strided directions");
	// (the two 64-cell chunks of a row share a cache line, and so do consecutive rows: qk_device.hpp)
	const BlockId blk = xcdContiguousBlock();
	const int bix = blk.x, biy = blk.y, biz = blk.z;
	const int b = biz / a.nseg;
	const int seg = biz - b * a.nseg;
	const qk_box bx = a.boxes[b];
	const SGeom g = a.geom[b];
	constexpr int OT = (DIR == 1) ? 2 : 1; // the other transverse axis (besides x)
	const int i_raw = bx.lo[0] + bix * 64 + threadIdx.x;
	const int ot_raw = bx.lo[OT] + biy * MARCH_BY + threadIdx.y;
	// lanes beyond the box stay in the wave (clamped addresses, masked stores) so that wave reductions are well defined
	const bool live = (i_raw <= bx.hi[0]) && (ot_raw <= bx.hi[OT]);
	const int i = min(i_raw, bx.hi[0]);
	const int ot = min(ot_raw, bx.hi[OT]);
	double sig0 = 0., sig1 = 0.;
	const int64_t T = a.total_cells;
	const int64_t st[3] = {1, g.pitch, static_cast<int64_t>(g.pitch) * g.n[1]};
	const int64_t ms = st[DIR]; // march stride
	// this segment's cells [lo, hi] of the box's march range; its faces lo .. hi+1 (a face between two segments is evaluated by both:
	// same value, stage-1 flux read-only in stage 2)
	const int seglen = (bx.hi[DIR] - bx.lo[DIR] + 1 + a.nseg - 1) / a.nseg;
	const int lo = bx.lo[DIR] + seg * seglen, hi = min(lo + seglen - 1, bx.hi[DIR]);
	const int nvalid = hi - lo + 1;
	if (nvalid <= 0) {
		return; // uniform for the workgroup
	}
	if constexpr (FUSEX) {
		if (bx.lo[0] + bix * 64 > bx.hi[0]) {
			return; // (uniform) a 64-cell chunk beyond a box narrower than the level's widest: nothing to do, and its edge faces would lie outside the fab
		}
	}
	// scratch index of march position p = lo - 3
	int pos[3];
	pos[0] = i;
	pos[OT] = ot;
	pos[DIR] = lo - 3;
	int64_t c = (pos[0] - g.glo[0]) * st[0] + (pos[1] - g.glo[1]) * st[1] + (pos[2] - g.glo[2]) * st[2];
	const double *S = a.scratch + g.off;
	double *Sw = a.scratch + g.off;
	RA4 Uin(a.U_in[b]);
	RA4 Uold(a.U_old[b]);
	const EpiConst ec = epiConst(eos);
	int64_t u = Uin.idx(pos[0], pos[1], pos[2]);
	const int64_t ums = (DIR == 1) ? Uin.js : Uin.ks;

	constexpr int AV = Axes<DIR, TWOD>::v, AW = Axes<DIR, TWOD>::w;
	double q[5][NV];
	double apPrev[NV], Fprev[NV];
	double vfPrev = 0., dVprev = 0., dWprev = 0.;
	// The cell a step completes (march position p - 3) was loaded as the newest cell three steps earlier: when the old state IS the input
	// state (stage 1) its conserved values wait in a per-lane ring of three LDS slots (no barriers: a lane only touches its own slots).
	// (Measured and rejected in round 4, profiles/round4/ab6_*: the newest cell's conserved values requested ONE STEP AHEAD through LDS-direct loads —
	// three global_load_lds_dwordx4 per wave and step into a 3-KiB slot, read out at the top of the next step, no register held meanwhile.  Y sweep
	// 0.722 ms with it, 0.721 ms without, on the same box: the sweep is bound by the rate at which the box's HBM delivers its bytes, not by the
	// latency of that load.  The Z sweep, at 248 VGPRs, spilled with it: 0.92 -> 1.11 ms.)
	// the carried form on a level with refined children: can this column (i, ot) hold marked cells, and where along the march (maskByte)
	bool maskCol = false;
	int maskLo = 0, maskHi = -1;
	if (CARRY && a.fluxMask != nullptr) {
		const qk_carray4 md = a.fluxMask[b];
		maskCol = live && i >= md.begin[0] && i < md.end[0] && ot >= md.begin[OT] && ot < md.end[OT];
		maskLo = md.begin[DIR];
		maskHi = md.end[DIR] - 1;
	}
	constexpr bool RING = LAST && (STAGE == 1);
	__shared__ double s_ring[RING ? 3 : 1][RING ? NV : 1][RING ? 64 * MARCH_BY : 1];
	__shared__ double s_fx[FUSEX ? MARCH_BY : 1][FUSEX ? NV + 2 : 1][FUSEX ? XW : 1];
	// FUSEX, the two faces of a row a wave cannot form from its own lanes (64 cells have 65 faces, and the edge state left of lane 0 belongs to a cell
	// outside the wave): formed 32 rows at a time — lanes 0..31 the left faces, 32..63 the right faces of the next 32 rows — and parked here with
	// the primitives of the two cells beyond either end of the row, which the reconstruction of the end lanes reads
	// entries of a row: 0, 1 the cells x0 - 2, x0 - 1; 2, 3 the cells x0 + 64, x0 + 65; 4 the right face (x0 + 64); 5 the left face (x0).  A row's entries are
	// copied into six spare slots of the wave's row buffer one step ahead (stageEdgeRow), off the dependency chain of the step that reads them.
	__shared__ double s_edge[FUSEX ? MARCH_BY : 1][FUSEX ? XROWS : 1][6][FUSEX ? NV + 1 : 1];
	double chiPrev = 1.;
	const int lself = (FUSEX && threadIdx.x == 0) ? 69 : static_cast<int>(threadIdx.x) + 2;
	const int lnext = (FUSEX && threadIdx.x == 63) ? 68 : static_cast<int>(threadIdx.x) + 3;
	// lanes 0..5 copy entry `lane` of row r of the batch into slots 0, 1, 66, 67 (halo cells), 68 (right face), 69 (left face) of the wave's row buffer
	auto stageEdgeRow = [&](int r) {
		if constexpr (FUSEX) {
			const int t6 = static_cast<int>(threadIdx.x);
			if (t6 < 6) {
				const int slot = (t6 < 2) ? t6 : 64 + t6;
				const double *src = s_edge[threadIdx.y][r][t6];
#pragma unroll
				for (int n = 0; n < NV + 1; ++n) {
					s_fx[threadIdx.y][n][slot] = src[n];
				}
			}
		}
	};
	const int tid = threadIdx.y * 64 + threadIdx.x;
	const bool ring = RING && a.same_old;
	int slot = 0;
	double Uo[NV];
#pragma unroll
	for (int n = 0; n < NV; ++n) {
		apPrev[n] = 0.;
		Fprev[n] = 0.;
#pragma unroll
		for (int m = 0; m < 5; ++m) {
			q[m][n] = 0.;
		}
	}

	// QK_MARCH_PREFETCH (A/B knob, default 1): the newest cell of the NEXT step is requested one step ahead (the fab holds the cell one beyond the last one a
	// march converts: 4 ghost cells, 3 of them read), so that a wave does not park on the round trip of the cell it is about to convert.  Six more doubles in
	// flight: not for the fused XY sweep, which sits at the 256-register limit of two waves per SIMD.  Same loads, same arithmetic: no bit changes.
	// Measured (profiles/round6/ab6_march_prefetch.txt): the Z sweep 0.873 -> 0.837 ms at 256^3, 6.61 -> 6.31 ms at 512^3 (234 -> 246 registers, no
	// scratch); requesting the pre-pass results (chi, D_v, D_w) of the next cell as well gives the gain back (252 registers).
	constexpr bool PFQ = (QK_MARCH_PREFETCH != 0) && !FUSEX && NS == 0; // (with passive scalars the six doubles push 21 instantiations past 256 registers: one wave per SIMD)
	double Un[NVAR];
	if constexpr (PFQ) {
#pragma unroll
		for (int n = 0; n < NVAR; ++n) {
			Un[n] = Uin.p[u + Uin.ns * n];
		}
	}
	for (int step = 0; step < nvalid + 6; ++step, c += ms, u += ums) {
		// shift the window; U(p) -> primitives of the newest cell
#pragma unroll
		for (int n = 0; n < NV; ++n) {
			q[0][n] = q[1][n];
			q[1][n] = q[2][n];
			q[2][n] = q[3][n];
			q[3][n] = q[4][n];
		}
		{
			double Uc[NVAR];
			if constexpr (PFQ) {
#pragma unroll
				for (int n = 0; n < NVAR; ++n) {
					Uc[n] = Un[n];
					Un[n] = Uin.p[u + ums + Uin.ns * n];
				}
			} else {
#pragma unroll
				for (int n = 0; n < NVAR; ++n) {
					Uc[n] = Uin.p[u + Uin.ns * n];
				}
			}
			if (a.prim_in) { // (uniform) the input array holds the primitives
#pragma unroll
				for (int n = 0; n < NVAR; ++n) {
					q[4][n] = Uc[n];
				}
			} else {
				consToPrim(eos, a.reconstruct_eint, Uc, q[4]);
			}
#pragma unroll
			for (int n = NVAR; n < NV; ++n) { // passive scalars are reconstructed as they are stored
				q[4][n] = Uin.p[u + Uin.ns * n];
			}
			if constexpr (RING) {
				if (ring) {
#pragma unroll
					for (int n = 0; n < NV; ++n) {
						Uo[n] = s_ring[slot][n][tid];
						s_ring[slot][n][tid] = (n < NVAR) ? Uc[n] : q[4][n];
					}
					slot = (slot == 2) ? 0 : slot + 1;
				}
			}
		}
		if (step < 4) {
			continue;
		}
		// cell cc = p - 2 in [lo-1, hi+1]
		const int64_t cc = c - 2 * ms;
		const double chi = S[(S_AUX + 0) * T + cc];
		const double dV = S[(S_AUX + 1 + AV) * T + cc];
		const double dW = S[(S_AUX + 1 + AW) * T + cc];
		// The accumulator of cell cc-1 (and in stage 2 the stage-1 flux of this step's face) are requested BEFORE the reconstruction and the
		// Riemann solve instead of where they are used: behind the face-flux stores they could not be hoisted by the compiler (may-alias),
		// and a wave parked on them for a full memory round trip per step (SQ_WAIT_ANY 59 % of the wave cycles at 2 waves per SIMD).
		double rhs_in[NV + 1], F1[NV + 1];
		int fidx[3];
		fidx[0] = i;
		fidx[OT] = ot;
		fidx[DIR] = lo + (step - 5);
		if (step >= 6) {
			const int64_t cu = cc - ms;
			if constexpr (FUSEX) {
				// the X sweep of the row this step completes (cell cu = cc - 1: primitives q[1], chi of the step before, D_z = dVprev)
				double(*sx)[XW] = s_fx[threadIdx.y];
				const int tx = static_cast<int>(threadIdx.x);
				const int l = tx + 2;
				const int rr = (step - 6) & (XROWS - 1);
				if (rr == 0) { // (uniform) the wave-edge faces and halo primitives of the next XROWS rows
					const int side = (tx / XROWS) & 1; // (lanes beyond 2 * XROWS repeat the work of the first ones: same values to the same slots)
					const int brow = tx & (XROWS - 1);
					const int mrow = min(lo + (step - 6) + brow, hi);
					const int f = bx.lo[0] + bix * 64 + 64 * side; // the face: between cells f - 1 and f
					int64_t ub = Uin.idx(f - 3, mrow, ot);
					double qb[6][NV];
#pragma unroll
					for (int m = 0; m < 6; ++m, ++ub) {
						double Uc[NVAR];
#pragma unroll
						for (int n = 0; n < NVAR; ++n) {
							Uc[n] = Uin.p[ub + Uin.ns * n];
						}
						if (a.prim_in) {
#pragma unroll
							for (int n = 0; n < NVAR; ++n) {
								qb[m][n] = Uc[n];
							}
						} else {
							consToPrim(eos, a.reconstruct_eint, Uc, qb[m]);
						}
#pragma unroll
						for (int n = NVAR; n < NV; ++n) {
							qb[m][n] = Uin.p[ub + Uin.ns * n];
						}
					}
					const int64_t cb = (f - 1 - g.glo[0]) + (mrow - g.glo[1]) * st[1] + (ot - g.glo[2]) * st[2];
					const double chiL = S[cb], chiR = S[cb + 1];
					const double DyL = S[(S_AUX + 2) * T + cb], DyR = S[(S_AUX + 2) * T + cb + 1];
					const double DzL = S[(S_AUX + 3) * T + cb], DzR = S[(S_AUX + 3) * T + cb + 1];
					double qLb[NV], qRb[NV], Fb[NV], vfb;
#pragma unroll
					for (int n = 0; n < NV; ++n) {
						double am_, ap_;
						cellEdges<ORDER>(qb[0][n], qb[1][n], qb[2][n], qb[3][n], qb[4][n], am_, ap_);
						flattenEdges(chiL, qb[2][n], am_, ap_);
						qLb[n] = ap_;
						cellEdges<ORDER>(qb[1][n], qb[2][n], qb[3][n], qb[4][n], qb[5][n], am_, ap_);
						flattenEdges(chiR, qb[3][n], am_, ap_);
						qRb[n] = am_;
					}
					{
						Wave wv;
						faceFlux<0, QK_RIEMANN_HLLC>(eos, a.reconstruct_eint, 3, qLb, qRb, qb[3][PVX] - qb[2][PVX], DyL, DyR, DzL, DzR, a.K_visc, Fb, vfb,
									     NS > 0 ? &wv : nullptr);
#pragma unroll
						for (int n = NVAR; n < NV; ++n) {
							Fb[n] = scalarFlux<QK_RIEMANN_HLLC>(wv, qLb[n], qRb[n]);
						}
					}
					double(*ed)[NV + 1] = s_edge[threadIdx.y][brow];
					double *ef = ed[5 - side];
#pragma unroll
					for (int n = 0; n < NV; ++n) {
						ef[n] = Fb[n];
					}
					ef[NV] = vfb;
					// halo cells of the row: x0 - 2, x0 - 1 (left face: cells f - 2, f - 1) and x0 + 64, x0 + 65 (right face: cells f, f + 1)
#pragma unroll
					for (int n = 0; n < NV; ++n) {
						ed[2 * side][n] = (side != 0) ? qb[3][n] : qb[1][n];
						ed[2 * side + 1][n] = (side != 0) ? qb[4][n] : qb[2][n];
					}
					waveFence();
					stageEdgeRow(0);
				}
				const double Dy = S[(S_AUX + 2) * T + cu];
#pragma unroll
				for (int n = 0; n < NV; ++n) {
					sx[n][l] = q[1][n];
				}
				waveFence();
				double amx[NV], apx[NV];
#pragma unroll
				for (int n = 0; n < NV; ++n) {
					cellEdges<ORDER>(sx[n][l - 2], sx[n][l - 1], q[1][n], sx[n][l + 1], sx[n][l + 2], amx[n], apx[n]);
					flattenEdges(chiPrev, q[1][n], amx[n], apx[n]);
				}
				const double uLeft = sx[PVX][l - 1];
				waveFence();
#pragma unroll
				for (int n = 0; n < NV; ++n) {
					sx[n][l] = apx[n];
				}
				sx[NV][l] = Dy;
				sx[NV + 1][l] = dVprev;
				waveFence();
				double qLx[NV], Fx[NV], vfx;
#pragma unroll
				for (int n = 0; n < NV; ++n) {
					qLx[n] = sx[n][l - 1];
				}
				const double dvl = sx[NV][l - 1], dwl = sx[NV + 1][l - 1];
				{
					Wave wv;
					faceFlux<0, QK_RIEMANN_HLLC>(eos, a.reconstruct_eint, 3, qLx, amx, q[1][PVX] - uLeft, dvl, Dy, dwl, dVprev, a.K_visc, Fx, vfx, NS > 0 ? &wv : nullptr);
#pragma unroll
					for (int n = NVAR; n < NV; ++n) {
						Fx[n] = scalarFlux<QK_RIEMANN_HLLC>(wv, qLx[n], amx[n]);
					}
				}
				waveFence();
#pragma unroll
				for (int n = 0; n < NV; ++n) {
					sx[n][l] = Fx[n];
				}
				sx[NV][l] = vfx;
				waveFence();
				// every lane reads the fluxes of its two faces back: its own slot and its right neighbour's — but lane 0 its left face from slot 69 and
				// lane 63 its right face from slot 68, where the batch's faces were staged a step ago (no branch, no select on 64-bit values)
#pragma unroll
				for (int n = 0; n < NV; ++n) {
					rhs_in[n] = a.inv_dx0 * (sx[n][lself] - sx[n][lnext]); // hydro_system.hpp:469
				}
				rhs_in[NV] = (sx[NV][lnext] - sx[NV][lself]) / a.dx0; // :803
				waveFence();
				if (rr + 1 < XROWS) {
					stageEdgeRow(rr + 1);
				}
			} else {
#pragma unroll
			for (int n = 0; n < NV + 1; ++n) {
				rhs_in[n] = streamLoad(&S[(S_RHS + n) * T + cu]);
			}
			}
			if (LAST && !ring && !(CARRY && STAGE == 2)) { // the old state of the cell this step completes (stage 2 of the carried form: S replaces it)
				int uc[3];
				uc[0] = i;
				uc[OT] = ot;
				uc[DIR] = lo + (step - 6);
				const int64_t co = Uold.idx(uc[0], uc[1], uc[2]);
#pragma unroll
				for (int n = 0; n < NV; ++n) {
					Uo[n] = Uold.p[co + Uold.ns * n];
				}
			}
			if (CARRY && LAST && STAGE == 2) { // the half step and the pressure stage 1 stored for this cell (F1[] is free in this mode)
				RA4 R1(a.rhs1[b]);
				int uc[3];
				uc[0] = i;
				uc[OT] = ot;
				uc[DIR] = lo + (step - 6);
				const int64_t c1 = R1.idx(uc[0], uc[1], uc[2]);
#pragma unroll
				for (int n = 0; n < NV + 1; ++n) {
					F1[n] = R1.p[c1 + R1.ns * n];
				}
			}
		}
		if (!CARRY && STAGE == 2 && step >= 5) {
			RA4 HF(a.halfFlux[b]);
			RA4 HV(a.halfVel[b]);
			const int64_t o = HF.idx(fidx[0], fidx[1], fidx[2]);
#pragma unroll
			for (int n = 0; n < NV; ++n) {
				F1[n] = HF.p[o + HF.ns * n];
			}
			F1[NV] = HV(fidx[0], fidx[1], fidx[2]);
		}
		double am[NV], ap[NV];
#pragma unroll
		for (int n = 0; n < NV; ++n) {
			cellEdges<ORDER>(q[0][n], q[1][n], q[2][n], q[3][n], q[4][n], am[n], ap[n]);
			flattenEdges(chi, q[2][n], am[n], ap[n]);
		}
		if (step >= 5) {
			// face between cells cc-1 and cc, index = (march coordinate of cc)
			const double du = q[2][PVX + DIR] - q[1][PVX + DIR];
			double F[NV], vf;
			{
				Wave wv;
				faceFlux<DIR, QK_RIEMANN_HLLC, TWOD>(eos, a.reconstruct_eint, TWOD ? 2 : 3, apPrev, am, du, dVprev, dV, dWprev, dW, a.K_visc, F, vf,
								     NS > 0 ? &wv : nullptr);
#pragma unroll
				for (int n = NVAR; n < NV; ++n) {
					F[n] = scalarFlux<QK_RIEMANN_HLLC>(wv, apPrev[n], am[n]);
				}
			}
			// FOFC pass: a face that touches a cell the first pass flagged takes the first-order flux of the old state (see firstOrderFlux)
			bool firstOrder = false;
			if constexpr (FOFC) {
				CIA4 flag(a.redoFlag[b]);
				int fm[3] = {fidx[0], fidx[1], fidx[2]};
				fm[DIR] -= 1;
				firstOrder = (flag(fm[0], fm[1], fm[2]) != 0) || (flag(fidx[0], fidx[1], fidx[2]) != 0);
			}
			auto replaceByFirstOrder = [&]() {
				double qLo[NV], qRo[NV];
				if (STAGE == 1) { // the old state is the input state: the window holds its primitives (cells cc - 1 and cc)
#pragma unroll
					for (int n = 0; n < NV; ++n) {
						qLo[n] = q[1][n];
						qRo[n] = q[2][n];
					}
				} else {
					int fm[3] = {fidx[0], fidx[1], fidx[2]};
					fm[DIR] -= 1;
					primOfCell<NS>(Uold, eos, a.reconstruct_eint, fm[0], fm[1], fm[2], qLo);
					primOfCell<NS>(Uold, eos, a.reconstruct_eint, fidx[0], fidx[1], fidx[2], qRo);
				}
				firstOrderFlux<DIR, NS, TWOD, TWOD ? 2 : 3>(eos, a.reconstruct_eint, qLo, qRo, F, vf);
			};
			if (CARRY) {
				// carried right-hand side: no face arrays — except on the marked faces of a level with refined children
				if (maskCol && fidx[DIR] >= maskLo && fidx[DIR] - 1 <= maskHi) { // the face's cells fidx - 1 and fidx along the march
					CA4 M(a.fluxMask[b]);
					int fm[3] = {fidx[0], fidx[1], fidx[2]};
					fm[DIR] -= 1;
					const int m0 = (fm[DIR] >= maskLo) ? M(fm[0], fm[1], fm[2]) : 0;
					const int m1 = (fidx[DIR] <= maskHi) ? M(fidx[0], fidx[1], fidx[2]) : 0;
					if ((m0 | m1) != 0) {
						maskedFaceFlux<STAGE, NV>(a, b, fidx[0], fidx[1], fidx[2], F);
					}
				}
			} else if (STAGE == 1 && FOFC) {
				if (firstOrder) { // (halfFlux keeps the uncorrected stage-1 flux of the first pass)
					replaceByFirstOrder();
				}
			} else if (STAGE == 1) {
				if (live) {
					WA4 HF(a.halfFlux[b]);
					WA4 HV(a.halfVel[b]);
					const int64_t o = HF.idx(fidx[0], fidx[1], fidx[2]);
#pragma unroll
					for (int n = 0; n < NV; ++n) {
						HF.p[o + HF.ns * n] = F[n];
					}
					HV(fidx[0], fidx[1], fidx[2]) = vf;
				}
			} else {
#pragma unroll
				for (int n = 0; n < NV; ++n) {
					F[n] = 0.5 * F1[n] + 0.5 * F[n];
				}
				vf = 0.5 * F1[NV] + 0.5 * vf;
				if (FOFC && firstOrder) {
					replaceByFirstOrder(); // flux_rk2 of this face as a whole
				}
				if (a.store_rk2 && live) {
					WA4 RF(a.rk2Flux[b]);
					const int64_t o2 = RF.idx(fidx[0], fidx[1], fidx[2]);
#pragma unroll
					for (int n = 0; n < NV; ++n) {
						RF.p[o2 + RF.ns * n] = F[n];
					}
				}
			}
			if (step >= 6) {
				// update cell u = cc - 1 (march coordinate lo + step - 6)
				const int64_t cu = cc - ms;
				double rhs[NV];
#pragma unroll
				for (int n = 0; n < NV; ++n) {
					rhs[n] = rhs_in[n] + a.inv_dx * (Fprev[n] - F[n]);
				}
				const double div_v = rhs_in[NV] + (vf - vfPrev) / a.dx;
				if (LAST) {
					int u[3];
					u[0] = i;
					u[OT] = ot;
					u[DIR] = lo + (step - 6);
					if (live) {
						updateCellFrom<NS, CARRY ? STAGE : 0, FOFC, TWOD ? 2 : 3>(a, eos, ec, b, u[0], u[1], u[2], Uo, rhs, div_v, F1, sig0, sig1,
													    (STAGE == 1) && a.same_old && !a.reconstruct_eint && !a.prim_in, q[1][PPRES]);
					}
				} else if (live) {
#pragma unroll
					for (int n = 0; n < NV; ++n) {
						streamStore(&Sw[(S_RHS + n) * T + cu], rhs[n]);
					}
					streamStore(&Sw[RHS_DIVV * T + cu], div_v);
				}
			}
#pragma unroll
			for (int n = 0; n < NV; ++n) {
				Fprev[n] = F[n];
			}
			vfPrev = vf;
		}
#pragma unroll
		for (int n = 0; n < NV; ++n) {
			apPrev[n] = ap[n];
		}
		dVprev = dV;
		dWprev = dW;
		chiPrev = chi;
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
inline auto smallLevelCells() -> int64_t
{
	static const int64_t v = [] {
		const char *e = std::getenv("QK_SMALL_LEVEL_CELLS");
		return (e != nullptr) ? static_cast<int64_t>(std::atoll(e)) : static_cast<int64_t>(2) << 20;
	}();
	return v;
}

inline auto smallMinLen() -> int
{
	static const int v = [] {
		const char *e = std::getenv("QK_SMALL_MINLEN");
		return (e != nullptr) ? std::max(1, std::atoi(e)) : 4;
	}();
	return v;
}

auto marchSegments(const qk_level *lev, int dir, int ot) -> int
{
	if (const char *e = std::getenv("QK_MARCH_SEGMENTS")) {
		return std::max(1, std::atoi(e));
	}
	const int64_t cols = static_cast<int64_t>(lev->nboxes) * lev->maxlen[0] * lev->maxlen[ot];
	const int64_t want = (131072 + cols - 1) / std::max<int64_t>(cols, 1);
	// a small level (the refined levels of a young hierarchy: a few 32^3 boxes) is bound by the length of one thread's march, not by the
	// redundant warm-up cells: segments of 4 cells there (8 -> 4: +2 % on the young Sedov hierarchy, 2 no better)
	const int minlen = (cols * lev->maxlen[dir] < smallLevelCells()) ? smallMinLen() : 16;
	const int cap = std::max(1, lev->maxlen[dir] / minlen);
	return static_cast<int>(std::min<int64_t>(std::max<int64_t>(want, 1), std::min(cap, 16)));
}

// Row pitch of the scratch arrays.  A dense row of a 128-cell box is 136 doubles = 8.5 cache lines and its first valid cell sits 32 bytes into a
// line: a wave's 512-byte access straddles 5 lines, neighbouring rows and chunks share lines.  With QK_ROW_ALIGN (default 1) rows are 16-double
// multiples apart and every box starts 12 doubles past a line boundary, so that valid cell 0 of EVERY row of every component begins a 128-byte
// line: the marching sweeps (which never touch the x ghost cells of the scratch arrays) move whole lines only.  The state arrays get the same
// layout from the hosts (any stride is legal in a qk_array4); QK_ROW_ALIGN=0 restores dense rows for A/B runs.
inline auto rowAlign() -> bool
{
	static const bool v = [] {
		const char *e = std::getenv("QK_ROW_ALIGN"); // 0: dense everywhere, 1 (default): state + scratch aligned, 2: state only, 3: scratch only
		return e == nullptr || std::atoi(e) == 1 || std::atoi(e) == 3;
	}();
	return v;
}

// geometry of the scratch arrays of a level: per box (offset, extents, pitch), total doubles per component
auto scratchGeom(const qk_level *lev, std::vector<SGeom> &g) -> int64_t
{
	g.resize(lev->nboxes);
	const bool al = rowAlign();
	int64_t off = 0;
	for (int b = 0; b < lev->nboxes; ++b) {
		for (int d = 0; d < 3; ++d) {
			g[b].glo[d] = lev->boxes[b].lo[d] - NG;
			g[b].n[d] = lev->boxes[b].hi[d] - lev->boxes[b].lo[d] + 1 + 2 * NG;
		}
		g[b].pitch = al ? (g[b].n[0] + 15) / 16 * 16 : g[b].n[0];
		if (al) {
			off = (off + 15) / 16 * 16 + (16 - NG); // cell NG of a row (the first valid one) on a line boundary
		}
		g[b].off = off;
		g[b].ncell = static_cast<int64_t>(g[b].pitch) * g[b].n[1] * g[b].n[2];
		off += g[b].ncell;
	}
	return al ? (off + 15) / 16 * 16 : off;
}

auto buildGeom(qk_level *lev) -> int
{
	if (lev->d_sgeom != nullptr) {
		return QK_OK;
	}
	std::vector<SGeom> g;
	const int64_t total = scratchGeom(lev, g);
	void *d = nullptr;
	QK_HIP_CHECK(lev->ctx, hipMalloc(&d, sizeof(SGeom) * lev->nboxes));
	QK_HIP_CHECK(lev->ctx, hipMemcpy(d, g.data(), sizeof(SGeom) * lev->nboxes, hipMemcpyHostToDevice));
	lev->d_sgeom = d;
	lev->sgeom_total_cells = total;
	return QK_OK;
}

template <int ORDER, int STAGE, int NS, bool CARRY, bool FOFC = false> void launchSweeps(qk_level *lev, hipStream_t s, SweepArgs a, Eos eos, const qk_hydro_stage_args *args)
{
	// the X sweep inside the Y sweep (FUSEX): 3-D, carried form without a flux mask, every box a whole number of 64-cell waves wide
	bool fusex = false;
	if constexpr (CARRY && !FOFC && NS == 0) { // (with passive scalars the fused kernel needs more than 256 registers: one wave per SIMD)
		const char *e = std::getenv("QK_FUSEX"); // (read per launch: tests and A/B runs switch it inside one process)
		const int want = (e != nullptr) ? std::atoi(e) : 1;
		fusex = want != 0 && lev->ndim == 3 && a.fluxMask == nullptr;
		for (int b = 0; fusex && b < lev->nboxes; ++b) {
			fusex = (lev->boxes[b].hi[0] - lev->boxes[b].lo[0] + 1) % 64 == 0;
		}
	}
	// X
	if (!fusex) {
		SweepArgs ax = a;
		ax.halfFlux = args->halfFlux[0];
		ax.halfVel = args->halfVel[0];
		ax.rk2Flux = args->fluxRk2[0];
		ax.inv_dx = 1.0 / args->dx[0];
		ax.dx = args->dx[0];
		const int64_t slab = static_cast<int64_t>(lev->maxlen[0] + 2 * NG) * lev->maxlen[1];
		const dim3 grid(static_cast<unsigned>((slab + XOUT - 1) / XOUT), static_cast<unsigned>(lev->maxlen[2]), static_cast<unsigned>(lev->nboxes));
		ProfScope ps(lev->ctx, s, "k_sweep_x");
		if (lev->ndim == 3) {
			hipLaunchKernelGGL((k_sweep_x<ORDER, STAGE, NS, CARRY, 3, FOFC>), grid, dim3(XB), 0, s, ax, eos);
		} else if constexpr (!CARRY) { // (1-D / 2-D builds: the reference's form of the RK2 average only)
			if (lev->ndim == 2) {
				hipLaunchKernelGGL((k_sweep_x<ORDER, STAGE, NS, false, 2, FOFC>), grid, dim3(XB), 0, s, ax, eos);
			} else {
				hipLaunchKernelGGL((k_sweep_x<ORDER, STAGE, NS, false, 1, FOFC>), grid, dim3(XB), 0, s, ax, eos); // + epilogue: the stage is complete
			}
		}
	}
	if (lev->ndim == 1) {
		return;
	}
	if (lev->ndim == 2) { // Y of a 2-D build (index-swap view) + epilogue
		if constexpr (!CARRY) {
			SweepArgs ay = a;
			ay.same_old = (args->U_in == args->U_old);
			ay.halfFlux = args->halfFlux[1];
			ay.halfVel = args->halfVel[1];
			ay.rk2Flux = args->fluxRk2[1];
			ay.inv_dx = 1.0 / args->dx[1];
			ay.dx = args->dx[1];
			ay.nseg = marchSegments(lev, 1, 2);
			const dim3 grid((lev->maxlen[0] + 63) / 64, (lev->maxlen[2] + MARCH_BY - 1) / MARCH_BY, lev->nboxes * ay.nseg);
			ProfScope ps(lev->ctx, s, "k_sweep_y");
			hipLaunchKernelGGL((k_sweep_march<1, ORDER, STAGE, true, NS, false, true, FOFC>), grid, dim3(64, MARCH_BY), 0, s, ay, eos);
		}
		return;
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
		ay.inv_dx0 = 1.0 / args->dx[0];
		ay.dx0 = args->dx[0];
		const dim3 grid((lev->maxlen[0] + 63) / 64, (lev->maxlen[2] + MARCH_BY - 1) / MARCH_BY, lev->nboxes * ay.nseg);
		if constexpr (CARRY && !FOFC && NS == 0) {
			if (fusex) {
				ProfScope ps(lev->ctx, s, "k_sweep_xy");
				hipLaunchKernelGGL((k_sweep_march<1, ORDER, STAGE, false, NS, CARRY, false, FOFC, true>), grid, dim3(64, MARCH_BY), 0, s, ay, eos);
			}
		}
		if (!fusex) {
			ProfScope ps(lev->ctx, s, "k_sweep_y");
			hipLaunchKernelGGL((k_sweep_march<1, ORDER, STAGE, false, NS, CARRY, false, FOFC>), grid, dim3(64, MARCH_BY), 0, s, ay, eos);
		}
	}
	// Z (+ epilogue)
	{
		SweepArgs az = a;
		az.same_old = (args->U_in == args->U_old);
		az.halfFlux = args->halfFlux[2];
		az.halfVel = args->halfVel[2];
		az.rk2Flux = args->fluxRk2[2];
		az.inv_dx = 1.0 / args->dx[2];
		az.dx = args->dx[2];
		az.nseg = marchSegments(lev, 2, 1);
		const dim3 grid((lev->maxlen[0] + 63) / 64, (lev->maxlen[1] + MARCH_BY - 1) / MARCH_BY, lev->nboxes * az.nseg);
		ProfScope ps(lev->ctx, s, "k_sweep_z");
		hipLaunchKernelGGL((k_sweep_march<2, ORDER, STAGE, true, NS, CARRY, false, FOFC>), grid, dim3(64, MARCH_BY), 0, s, az, eos);
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
	std::vector<SGeom> g;
	const int64_t cells = scratchGeom(lev, g);
	return cells * scratchComps(t->nscalars) * static_cast<int64_t>(sizeof(double));
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
	if (int rc = needsLibraryEos(ctx, t, "qk_hydro_stage_fused (EnforceLimits)"); rc != QK_OK) {
		return rc;
	}
	QK_REQUIRE(ctx, args != nullptr, "qk_hydro_stage_fused: NULL args");
	if (lev->nboxes == 0) {
		return QK_OK; // a rank without boxes on this level
	}
	if (t->nscalars > QK_FUSED_MAX_SCALARS || t->nmscalars != 0) {
		return setError(ctx, QK_ERR_UNSUPPORTED, "qk_hydro_stage_fused: up to 3 passive scalars, no mass scalars (use the reference-shaped operators)");
	}
	if (t->ndim != lev->ndim) {
		return setError(ctx, QK_ERR_INVALID, "qk_hydro_stage_fused: traits.ndim differs from the level's dimension");
	}
	if (t->ndim != 3 && args->rk2_carry_rhs != 0) {
		return setError(ctx, QK_ERR_UNSUPPORTED, "qk_hydro_stage_fused: rk2_carry_rhs is instantiated for 3-D builds");
	}
	QK_REQUIRE(ctx, args->K_visc >= 0.0, "qk_hydro_stage_fused: negative artificial-viscosity coefficient");
	QK_REQUIRE(ctx, args->stage == 1 || args->stage == 2, "qk_hydro_stage_fused: stage must be 1 or 2");
	QK_REQUIRE(ctx, args->reconstruction_order >= 1 && args->reconstruction_order <= 3, "qk_hydro_stage_fused: reconstruction_order must be 1..3");
	QK_REQUIRE(ctx, args->U_in && args->U_old && args->U_out && args->redoFlag && args->d_redo_count && args->d_error_flag && args->scratch,
		   "qk_hydro_stage_fused: NULL array");
	QK_REQUIRE(ctx, args->fofc_pass == 0 || (args->K_visc == 0.0 && (args->rk2_carry_rhs == 0 || args->stage == 1)),
		   "qk_hydro_stage_fused: the fused first-order flux correction pass needs K_visc == 0 and, in the carried-rhs form, stage 1 (use the reference-shaped operators otherwise)");
	QK_REQUIRE(ctx, args->rk2_carry_rhs == 0 || (args->rhs1 != nullptr && args->store_flux_rk2 == 0),
		   "qk_hydro_stage_fused: rk2_carry_rhs needs rhs1 and excludes store_flux_rk2 (flux_rk2 is formed on the faces flux_mask marks, nowhere else)");
	if (args->rk2_carry_rhs != 0 && args->flux_mask != nullptr) {
		for (int d = 0; d < t->ndim; ++d) {
			QK_REQUIRE(ctx, args->halfFlux[d] != nullptr && args->fluxRk2[d] != nullptr && args->fluxRk2[d] != args->halfFlux[d],
				   "qk_hydro_stage_fused: flux_mask needs halfFlux[d] and a distinct fluxRk2[d]");
		}
	}
	for (int d = 0; d < t->ndim; ++d) {
		QK_REQUIRE(ctx, args->rk2_carry_rhs != 0 || (args->halfFlux[d] && args->halfVel[d]), "qk_hydro_stage_fused: NULL halfFlux/halfVel");
		QK_REQUIRE(ctx, args->store_flux_rk2 == 0 || args->stage != 2 || (args->fluxRk2[d] != nullptr && args->fluxRk2[d] != args->halfFlux[d]),
			   "qk_hydro_stage_fused: store_flux_rk2 needs fluxRk2[d], distinct from halfFlux[d]");
		QK_REQUIRE(ctx, lev->maxlen[d] >= 1, "qk_hydro_stage_fused: empty box");
	}
	QK_REQUIRE(ctx, args->scratch_bytes >= qk_hydro_stage_scratch_bytes(lev, t), "qk_hydro_stage_fused: scratch too small");
	if (args->prim_in != 0 || args->prim_out != 0) {
		QK_REQUIRE(ctx, t->reconstruct_eint == 0 && !Eos(*t).isothermal && t->eos_temperature_model == 0,
			   "qk_hydro_stage_fused: the primitive hand-off is the gamma-law, reconstruct_eint = 0 form");
		QK_REQUIRE(ctx, args->fofc_pass == 0, "qk_hydro_stage_fused: the primitive hand-off has no correction pass (redo the step without it)");
		QK_REQUIRE(ctx, (args->prim_out == 0 || args->stage == 1) && (args->prim_in == 0 || (args->stage == 2 && args->U_in != args->U_old)),
			   "qk_hydro_stage_fused: prim_out belongs to stage 1, prim_in to stage 2 (whose old state is a different array)");
	}

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

	// 1. flattening coefficient and velocity differences (valid + 1), one pass over U
	{
		ProfScope ps(ctx, s, "k_pre");
		const int xt = (lev->maxlen[0] + 2 + PT_X - 1) / PT_X, yt = (lev->maxlen[1] + 2 + PT_Y - 1) / PT_Y;
		// Segments along z.  Every segment pays 6 planes of warm-up (own columns only), a CU holds two workgroups (512 slots on the chip), and
		// short segments keep the workgroups of neighbouring tiles at the same z, where their halo rows are still in the XCD's L2.  Measured
		// (profiles/round4/ab7_prepass_segments.txt): 256^3 in 128^3 boxes (304 tiles) 3 / 4 / 5 / 6 segments 0.419 / 0.428 / 0.405 / 0.423 ms —
		// the order of (idle share of the last round of workgroups) x (warm-up share); 512^3 (2432 tiles) 1 / 2 / 3 / 4 / 5 / 6 segments 3.10 /
		// 3.00 / 2.96 / 2.88 / 2.99 / 3.02 ms.  So: segments of about 30 planes; with few rounds of workgroups the neighbour count that wastes
		// least.  (A small level: shorter segments, see marchSegments.)
		const bool small = static_cast<int64_t>(lev->nboxes) * lev->maxlen[0] * lev->maxlen[1] * lev->maxlen[2] < smallLevelCells();
		const int64_t tiles = static_cast<int64_t>(xt) * yt * lev->nboxes;
		const int nplanes = lev->maxlen[2] + 2;
		int nseg;
		if (small) {
			nseg = static_cast<int>(std::min<int64_t>((1024 + tiles - 1) / tiles, std::max(1, nplanes / smallMinLen())));
		} else {
			const int base = std::max(1, std::min(8, (nplanes + 15) / 30));
			nseg = base;
			const double slots = 512.0;
			if (static_cast<double>(tiles) * base / slots < 6.0) {
				double best = 1e300;
				for (int c = std::max(1, base - 1); c <= base + 1; ++c) {
					const double rounds = static_cast<double>(tiles) * c / slots;
					const int seglen = (nplanes + c - 1) / c;
					const double cost = (std::ceil(rounds) / rounds) * (static_cast<double>(seglen + 6) / seglen);
					if (cost < best) {
						best = cost;
						nseg = c;
					}
				}
			}
		}
		if (const char *e = std::getenv("QK_PRE_SEGMENTS")) {
			nseg = std::max(1, std::atoi(e));
		}
		const dim3 grid(static_cast<unsigned>(xt) * yt * lev->nboxes * nseg);
		if (args->prim_in != 0) {
			hipLaunchKernelGGL(k_pre3<true>, grid, dim3(PT_THREADS), 0, s, boxes, geom, args->U_in, scratch, T, eos, re, nseg, xt, yt, t->ndim);
		} else {
			hipLaunchKernelGGL(k_pre3<false>, grid, dim3(PT_THREADS), 0, s, boxes, geom, args->U_in, scratch, T, eos, re, nseg, xt, yt, t->ndim);
		}
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
	a.rhs1 = args->rhs1;
	a.fluxMask = (args->rk2_carry_rhs != 0) ? args->flux_mask : nullptr;
	a.prim_in = (args->prim_in != 0);
	a.prim_out = (args->prim_out != 0);
	for (int d = 0; d < 3; ++d) {
		a.dx3[d] = args->dx[d];
	}

#define QK_LAUNCH_NS(ORDER, NS)                                                                                                                      \
	if (args->fofc_pass != 0) {                                                                                                                  \
		if (args->stage == 1) {                                                                                                              \
			launchSweeps<ORDER, 1, NS, false, true>(lev, s, a, eos, args);                                                               \
		} else {                                                                                                                             \
			launchSweeps<ORDER, 2, NS, false, true>(lev, s, a, eos, args);                                                               \
		}                                                                                                                                    \
	} else if (args->rk2_carry_rhs != 0) {                                                                                                              \
		if (args->stage == 1) {                                                                                                              \
			launchSweeps<ORDER, 1, NS, true>(lev, s, a, eos, args);                                                                      \
		} else {                                                                                                                             \
			launchSweeps<ORDER, 2, NS, true>(lev, s, a, eos, args);                                                                      \
		}                                                                                                                                    \
	} else if (args->stage == 1) {                                                                                                               \
		launchSweeps<ORDER, 1, NS, false>(lev, s, a, eos, args);                                                                             \
	} else {                                                                                                                                     \
		launchSweeps<ORDER, 2, NS, false>(lev, s, a, eos, args);                                                                             \
	}
#define QK_LAUNCH(ORDER)                                                                                                                             \
	switch (t->nscalars) {                                                                                                                       \
	case 0:                                                                                                                                      \
		QK_LAUNCH_NS(ORDER, 0) break;                                                                                                        \
	case 1:                                                                                                                                      \
		QK_LAUNCH_NS(ORDER, 1) break;                                                                                                        \
	case 2:                                                                                                                                      \
		QK_LAUNCH_NS(ORDER, 2) break;                                                                                                        \
	default:                                                                                                                                     \
		QK_LAUNCH_NS(ORDER, 3) break;                                                                                                        \
	}
	if (args->reconstruction_order == 3) {
		QK_LAUNCH(3)
	} else if (args->reconstruction_order == 2) {
		QK_LAUNCH(2)
	} else {
		QK_LAUNCH(1)
	}
#undef QK_LAUNCH_NS
#undef QK_LAUNCH
	QK_HIP_CHECK(ctx, hipGetLastError());
	return QK_OK;
}

} // extern "C"
