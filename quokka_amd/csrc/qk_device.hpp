// qk_device.hpp — per-cell / per-face device arithmetic of the hydro hot path, written for gfx950.
//
// Every function is the device-side counterpart of one reference routine (cited per function) and
// keeps the reference's floating-point association order; the translation unit is compiled with
// -ffp-contract=off.  Differences from the reference that are value-preserving by construction:
//   * nvar = 6, no passive / mass scalars: quokka::valarray<6> ops are written out per component
//     and the multiplications by the literal zeros of D_L/D_R/D_star (HLLC.hpp:116-118) are dropped
//     (x + 0.0 == x up to the sign of a zero);
//   * HLLC selects the Riemann-fan side FIRST and evaluates only that side's F / F* (the reference
//     evaluates all four and selects, HLLC.hpp:132-150) — same values, about half the divisions;
//   * std::pow(cs, 2) (hydro_system.hpp:602) is cs*cs;
//   * the divisions of the Riemann path that share a denominator share its refined reciprocal (recipOf / divBy below):
//     the same operations on the same operands as the compiler's expansion of `/`, i.e. the same bits for normal-range
//     operands; a zero quotient may differ in sign, a zero / subnormal / infinite denominator gives NaN instead of +-inf.  Since round 4
//     EVERY division of the Riemann path takes this form (divN / recipExact for denominators used once: 7-8 instructions against the 11 of
//     `/`), the division by the run-time constant kB_user goes through a reciprocal formed on the host, and the square roots are sqrtN
//     (15 instructions against 22: no range scaling for arguments below 2^-767) — 7 square roots and 10 lone divisions per face.
#ifndef QK_DEVICE_HPP_
#define QK_DEVICE_HPP_

#include <hip/hip_runtime.h>

#include "quokka_amd.h"

#define QK_DEV __device__ __forceinline__

namespace qk
{

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2) in the order x, y, z of the grid, so neighbouring workgroups — which
// share cache lines: rows start 32 bytes into a 128-byte line and are 8.5 lines long, slabs and tiles read each other's halo — land on different
// XCDs and fetch the shared lines from HBM twice.  The linear workgroup id is remapped so that one XCD works through a contiguous run of
// workgroups.  Returns the block index this workgroup should work on.  Same-box A/B: the marching hydro sweeps Y -2.8 %, Z -3.9 % at 256^3 (the
// Z sweep 0.39 -> 0.41 of the HBM roofline at 512^3), the marching radiation sweeps -1 ... -2.5 %; the LDS slab sweeps along x (hydro and
// radiation), which are bound by FP64 issue, became 2 ... 5 % SLOWER with it and keep the hardware's order.
struct BlockId {
	int x, y, z;
};
QK_DEV auto xcdContiguousBlock() -> BlockId
{
	const unsigned gx = gridDim.x, gy = gridDim.y, nblk = gx * gy * gridDim.z;
	const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
	const unsigned q8 = nblk / 8, r8 = nblk % 8, xcd = lin % 8, turn = lin / 8;
	const unsigned logical = (xcd < r8) ? xcd * (q8 + 1) + turn : r8 * (q8 + 1) + (xcd - r8) * q8 + turn;
	return {static_cast<int>(logical % gx), static_cast<int>((logical / gx) % gy), static_cast<int>(logical / (gx * gy))};
}

constexpr int NVAR = 6;
// HydroSystem::consVarIndex / primVarIndex (hydro_system.hpp:54-72)
enum { RHO = 0, MX = 1, MY = 2, MZ = 3, ENE = 4, EINT = 5 };
enum { PRHO = 0, PVX = 1, PVY = 2, PVZ = 3, PPRES = 4, PEINT = 5 };

// A pointer that was LOADED from memory (the data pointer of an array descriptor) is a generic pointer to the compiler: every access through
// it becomes a FLAT instruction, which counts in vmcnt AND lgkmcnt and may return out of order, so hipcc waits for everything
// (s_waitcnt vmcnt(0) lgkmcnt(0)) before the first use of any such load — including the stores a loop issued just before.  All arrays of the
// C-ABI are device allocations (global_load / global_store, in-order vmcnt).
// The pointer is therefore kept with its address space (1 = global) in the accessor.
template <typename T, typename D> struct A4 {
	using GT = __attribute__((address_space(1))) T;
	GT *p;
	int64_t js, ks, ns;
	int bx, by, bz;
	QK_DEV explicit A4(D const &d) : p((GT *)d.p), js(d.jstride), ks(d.kstride), ns(d.nstride), bx(d.begin[0]), by(d.begin[1]), bz(d.begin[2]) {}
	QK_DEV auto idx(int i, int j, int k) const -> int64_t { return (i - bx) + js * (j - by) + ks * (k - bz); }
	QK_DEV auto operator()(int i, int j, int k, int n = 0) const -> GT & { return p[idx(i, j, k) + ns * n]; }
	// generic pointer to one element (for the atomic* functions of the HIP headers, which take generic pointers)
	QK_DEV auto ptr(int i, int j, int k, int n = 0) const -> T * { return (T *)&p[idx(i, j, k) + ns * n]; }
};
using RA4 = A4<const double, qk_array4>;
using WA4 = A4<double, qk_array4>;
using IA4 = A4<int, qk_iarray4>;
using CIA4 = A4<const int, qk_iarray4>;

// loop index -> view index (ArrayView_3d.hpp:22-28) expressed as a unit offset: a view step of
// (di, dj, dk) is the array step (i + di*e[dir]) + ..., i.e. axis `dir` is the normal, the next two
// (cyclically) are view-j and view-k.
// TWOD: the X2 view of an AMREX_SPACEDIM == 2 build is an index SWAP (ArrayView_2d.hpp:13-17, velocity components hydro_system.hpp:963-966):
// view-j is the x axis, view-k stays z.
template <int DIR, bool TWOD = false> struct Axes {
	static constexpr int n = DIR;					       // array axis of view-i (normal)
	static constexpr int v = (TWOD && DIR == 1) ? 0 : (DIR + 1) % 3;       // array axis of view-j
	static constexpr int w = (TWOD && DIR == 1) ? 2 : (DIR + 2) % 3;       // array axis of view-k
};

QK_DEV auto unit(int axis, int comp) -> int { return axis == comp ? 1 : 0; }

// Correctly rounded FP64 division with the refined reciprocal of the denominator SHARED between numerators.
// hipcc expands `n / d` to  div_scale x2, rcp, two Newton steps (4 fma), q = n * r, e = fma(-d, q, n), div_fmas(e, r, q),
// div_fixup  — 11 instructions, one of them quarter rate, in one dependency chain.  The Riemann solver divides 4 numerators
// by rho_L, 4 by rho_R, 2 + 2 by (gamma-1) rho and 6 by S_K - S*: with the reciprocal refined once per denominator every
// further quotient is mul + 2 fma, the SAME three operations on the SAME operands as the tail of the expansion, hence the
// same bits whenever div_scale does not rescale and div_fixup does not intervene: denominator and quotient in the normal
// range (roughly |d|, |n / d| in [2^-1020, 2^1020] with exponents of n and d less than 768 apart).  Outside it: a zero
// quotient may come out as +0 where IEEE gives -0; a zero, infinite or subnormal DENOMINATOR (rho, (gamma-1) rho, S_K - S*)
// yields NaN where IEEE gives +-inf or a rounded subnormal quotient — such a state is invalid in the reference as well (the
// cell is flagged and the step retried), it just fails with a different non-number.  A per-face range guard with a plain-
// division fallback was measured: the second code path costs more than the shared reciprocals save.
struct Recip {
	double d, r;
};
QK_DEV auto recipOf(double d) -> Recip
{
	Recip R;
	R.d = d;
	const double r0 = __builtin_amdgcn_rcp(d);
	double e = __builtin_fma(-d, r0, 1.0);
	const double r1 = __builtin_fma(r0, e, r0);
	e = __builtin_fma(-d, r1, 1.0);
	R.r = __builtin_fma(r1, e, r1);
	return R;
}
QK_DEV auto divBy(double n, Recip const &R) -> double
{
	const double q = n * R.r;
	const double e = __builtin_fma(-R.d, q, n);
	return __builtin_fma(e, R.r, q);
}

// Correctly rounded FP64 square root without the range scaling of hipcc's expansion.  `sqrt(x)` compiles to 22 VALU instructions: compare
// against 2^-767, select a scale, ldexp, v_rsq_f64, two multiplies and seven fma of Goldschmidt / Newton refinement, ldexp back, and the class test
// that returns x itself for +-0 and +inf.  The seven-instruction tail around the refinement only serves arguments below 2^-767 (1e-231): no
// density, c_s^2 or v^2 of a valid state comes near.  This is the same refinement on the same operands (the same bits for x >= 2^-767) plus the
// class test — sqrt(0) = 0 matters: |v| of gas at rest —, 15 instructions; a PPM + HLLC face takes seven square roots.
QK_DEV auto sqrtN(double x) -> double
{
	const double y = __builtin_amdgcn_rsq(x);
	double g = x * y;
	double h = y * 0.5;
	const double r = __builtin_fma(-h, g, 0.5);
	g = __builtin_fma(g, r, g);
	double d = __builtin_fma(-g, g, x);
	h = __builtin_fma(h, r, h);
	g = __builtin_fma(d, h, g);
	d = __builtin_fma(-g, g, x);
	g = __builtin_fma(d, h, g);
	return __builtin_amdgcn_class(x, 0x260) ? x : g; // +-0, +inf
}
// 1 / d, correctly rounded (recipOf + the two-fma tail of divBy with numerator 1)
QK_DEV auto recipExact(double d) -> double
{
	const double r0 = __builtin_amdgcn_rcp(d);
	double e = __builtin_fma(-d, r0, 1.0);
	const double r1 = __builtin_fma(r0, e, r0);
	e = __builtin_fma(-d, r1, 1.0);
	const double r = __builtin_fma(r1, e, r1);
	e = __builtin_fma(-d, r, 1.0);
	return __builtin_fma(e, r, r);
}
// n / d for a single numerator: the reciprocal refined for this quotient alone (8 instructions against the 11 of `/`: no div_scale / div_fixup)
QK_DEV auto divN(double n, double d) -> double { return divBy(n, recipOf(d)); }

// quokka::EOS<problem_t>, gamma-law, direct association (oracle/eos.hpp variant 0)
struct Eos {
	double gamma, gm1, cs_iso, mu, kB_ratio_num, kB_user; // mu = mean_molecular_weight / m_u
	bool isothermal;
	int tmodel;   // temperature hooks: 0 gamma-law, 1 E_int = (alpha / 4) T^4
	double alpha;
	Recip RkBu;   // 1 / kB_user: a run-time constant, its reciprocal (correctly rounded by the constructor's IEEE division) arrives in scalar registers
	Recip RkB, Rmu; // 1 / k_B, 1 / (mu m_u): likewise (the matter-radiation exchange and the temperature floors divide by them in every cell)
	static constexpr double k_B = 1.380649e-16;
	static constexpr double m_u = 1.6605390666e-24;

	__host__ __device__ explicit Eos(qk_hydro_traits const &t)
	    : gamma(t.gamma), gm1(t.gamma - 1.0), cs_iso(t.cs_isothermal), mu(t.mean_molecular_weight / m_u), kB_ratio_num(k_B),
	      kB_user(t.boltzmann_constant), isothermal(t.gamma == 1.0), tmodel(t.eos_temperature_model), alpha(t.eos_alpha),
	      RkBu{t.boltzmann_constant, 1.0 / t.boltzmann_constant}, RkB{k_B, 1.0 / k_B},
	      Rmu{(t.mean_molecular_weight / m_u) * m_u, 1.0 / ((t.mean_molecular_weight / m_u) * m_u)}
	{
	}
	// EOS.hpp:304-348 : e = Eint/rho (0 if rho == 0) ; p = (gamma-1) rho e
	QK_DEV auto pressure(double rho, double Eint) const -> double
	{
		const double e = (rho == 0.0) ? 0.0 : Eint / rho;
		return gm1 * rho * e;
	}
	// EOS.hpp:350-383 : cs = sqrt(gamma p / rho)
	QK_DEV auto soundSpeed(double rho, double P) const -> double { return sqrt(gamma * P / rho); }
	// EOS.hpp:161-200 : Eint = (p / ((gamma-1) rho)) * rho
	QK_DEV auto eintFromPres(double rho, double P) const -> double { return (P / (gm1 * rho)) * rho; }
	// EOS.hpp:74-114
	QK_DEV auto tgasFromEint(double rho, double Eint) const -> double
	{
		if (tmodel == 1) { // (4 E / alpha)^(1/4) as two correctly rounded square roots (the reference calls std::pow(x, 1./4.))
			return sqrt(sqrt(4.0 * Eint / alpha));
		}
		const double e = Eint / rho;
		const double T = e * mu * m_u * gm1 / k_B;
		return T * k_B / kB_user;
	}
	// EOS.hpp:116-159
	QK_DEV auto eintFromTgas(double rho, double T) const -> double
	{
		if (tmodel == 1) {
			return (alpha / 4.0) * ((T * T) * (T * T));
		}
		const double p = rho * T * k_B / (mu * m_u);
		const double e = p / (gm1 * rho);
		return e * rho * kB_user / k_B;
	}
};

// hydro_system.hpp:349-372 ComputePressure(cons, i,j,k) from the six conserved values
QK_DEV auto consPressure(Eos const &eos, double rho, double px, double py, double pz, double E) -> double
{
	const double vx = px / rho;
	const double vy = py / rho;
	const double vz = pz / rho;
	const double kinetic_energy = 0.5 * rho * (vx * vx + vy * vy + vz * vz);
	const double thermal_energy = E - kinetic_energy;
	if (eos.isothermal) {
		return rho * eos.cs_iso * eos.cs_iso;
	}
	return eos.pressure(rho, thermal_energy);
}

QK_DEV auto sgn(double v) -> int { return static_cast<int>(0.0 < v) - static_cast<int>(v < 0.0); }
// std::min / std::max as ONE v_min_f64 / v_max_f64 each.  `(b < a) ? b : a` compiles to v_cmp_lt_f64 + s_nop + 2 x v_cndmask_b32 (a 64-bit
// select is two 32-bit ones), four issue slots where the hardware has a one-slot instruction; the sweeps are issue-bound and a PPM
// reconstruction + HLLC solve holds ~60 min/max per cell (profiles/round2/ubench_*.txt: v_min_f64 1.9 ns, the select form 6.3 ns per
// wave-instruction per SIMD).  Written as inline assembly because fmin()/fmax() add a canonicalising v_max_f64(x, x) per operand in IEEE
// mode (the variant round 1 measured as slower).  Same value as std::min / std::max for ordered operands; differences, none of which a
// valid state reaches: a tie between +0 and -0 may return the other zero, and a NaN in the FIRST argument is dropped (std::min returns it)
// — the second-argument NaN cases the reference relies on (min(1., 0./0.) = 1. in the carbuncle / low-Mach factors) behave identically.
QK_DEV auto smin(double a, double b) -> double
{
	double r;
	asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
QK_DEV auto smax(double a, double b) -> double
{
	double r;
	asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
// min / max against the inline constants 0 and 1 (no register for the constant)
QK_DEV auto smin0(double a) -> double
{
	double r;
	asm("v_min_f64 %0, %1, 0" : "=v"(r) : "v"(a));
	return r;
}
QK_DEV auto smax0(double a) -> double
{
	double r;
	asm("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(a));
	return r;
}
QK_DEV auto smin1(double a) -> double
{
	double r;
	asm("v_min_f64 %0, 1.0, %1" : "=v"(r) : "v"(a));
	return r;
}
// (v < lo) ? lo : (hi < v) ? hi : v for lo <= hi
QK_DEV auto clampd(double v, double lo, double hi) -> double { return smin(smax(v, lo), hi); }

// hydro_system.hpp:138-196 ConservedToPrimitive for one cell: q = (rho, vx, vy, vz, P | e, Eint | e_aux); the quotients by rho share its
// refined reciprocal (same bits as `/` for a normal-range rho, see recipOf)
QK_DEV void consToPrim(Eos const &eos, bool reconstruct_eint, const double U[NVAR], double q[NVAR])
{
	const double rho = U[RHO];
	const Recip R = recipOf(rho);
	const double vx = divBy(U[MX], R);
	const double vy = divBy(U[MY], R);
	const double vz = divBy(U[MZ], R);
	const double kinetic_energy = 0.5 * rho * (vx * vx + vy * vy + vz * vz);
	const double Eint_cons = U[ENE] - kinetic_energy;
	q[PRHO] = rho;
	q[PVX] = vx;
	q[PVY] = vy;
	q[PVZ] = vz;
	if (reconstruct_eint) {
		q[PPRES] = divBy(Eint_cons, R);
		q[PEINT] = divBy(U[EINT], R);
	} else {
		if (eos.isothermal) {
			q[PPRES] = rho * eos.cs_iso * eos.cs_iso;
		} else {
			const double e = (rho == 0.0) ? 0.0 : divBy(Eint_cons, R);
			q[PPRES] = eos.gm1 * rho * e;
		}
		q[PEINT] = U[EINT];
	}
}

// hyperbolic_system.hpp:58-66
QK_DEV auto MC(double a, double b) -> double { return 0.5 * (sgn(a) + sgn(b)) * smin(0.5 * fabs(a + b), smin(2.0 * fabs(a), 2.0 * fabs(b))); }
QK_DEV auto minmod(double a, double b) -> double { return 0.5 * (sgn(a) + sgn(b)) * smin(fabs(a), fabs(b)); }

// hyperbolic_system.hpp:337-433 : PPM edges of cell i from q(i-2..i+2).
//   am -> rightState(i) (left edge of the cell), ap -> leftState(i+1) (right edge)
QK_DEV void ppmEdges(double qm2, double qm1, double q0, double qp1, double qp2, double &am, double &ap)
{
	// std::minmax({q0, qm1, qp1})
	const double lo = smin(smin(q0, qm1), qp1);
	const double hi = smax(smax(q0, qm1), qp1);
	const double coef_1 = (7. / 12.);
	const double coef_2 = (-1. / 12.);
	const double a_minus = (coef_1 * q0 + coef_2 * qp1) + (coef_1 * qm1 + coef_2 * qm2);
	const double a_plus = (coef_1 * qp1 + coef_2 * qp2) + (coef_1 * q0 + coef_2 * qm1);
	double new_a_minus = clampd(a_minus, lo, hi);
	double new_a_plus = clampd(a_plus, lo, hi);
	const double a = q0;
	const double dq_minus = (a - new_a_minus);
	const double dq_plus = (new_a_plus - a);
	const double qa = dq_plus * dq_minus;
	if (qa <= 0.0) {
		const double dq0 = MC(qp1 - q0, q0 - qm1);
		new_a_minus = a - 0.5 * dq0;
		new_a_plus = a + 0.5 * dq0;
	} else {
		if (fabs(dq_minus) >= 2.0 * fabs(dq_plus)) {
			new_a_minus = a - 2.0 * dq_plus;
		}
		if (fabs(dq_plus) >= 2.0 * fabs(dq_minus)) {
			new_a_plus = a + 2.0 * dq_minus;
		}
	}
	am = new_a_minus;
	ap = new_a_plus;
}

// hyperbolic_system.hpp:218-247 : PLM. For cell-centred bookkeeping: the right edge of cell i is
// leftState(i+1) = q(i) + 0.25*lim(q(i+1)-q(i), q(i)-q(i-1)); the left edge is
// rightState(i) = q(i) - 0.25*lim(q(i+1)-q(i), q(i)-q(i-1)).
template <int LIMITER> QK_DEV void plmEdges(double qm1, double q0, double qp1, double &am, double &ap)
{
	const double slope = (LIMITER == QK_LIMITER_MINMOD) ? minmod(qp1 - q0, q0 - qm1) : MC(qp1 - q0, q0 - qm1);
	ap = q0 + 0.25 * slope;
	am = q0 - 0.25 * slope;
}

// hydro_system.hpp:588-622 : flattening coefficient from P(i-2..i+2), rho(i), vn(i-1), vn(i+1).  K_S = rho c_s^2 depends on the centre cell
// only (:600-604) and is shared between the directions by the fused pre-pass.
QK_DEV auto flatteningKS(Eos const &eos, double rho, double P) -> double
{
	if (eos.isothermal) {
		return rho * eos.cs_iso * eos.cs_iso;
	}
	const double cs = eos.soundSpeed(rho, P);
	return (cs * cs) * rho;
}
QK_DEV auto flatteningChiKS(double Pm2, double Pm1, double Pp1, double Pp2, double K_S, double vm1, double vp1) -> double
{
	constexpr double beta_max = 0.85;
	constexpr double beta_min = 0.75;
	constexpr double Zmax = 0.75;
	constexpr double Zmin = 0.25;
	const double beta_denom = fabs(Pp2 - Pm2);
	const double beta = (beta_denom != 0) ? (fabs(Pp1 - Pm1) / beta_denom) : 0;
	const double chi_min = smax0(smin1((beta_max - beta) / (beta_max - beta_min)));
	const double Z = fabs(Pp1 - Pm1) / K_S;
	const double chi_shock = smax(chi_min, smin1((Zmax - Z) / (Zmax - Zmin)));
	return (vp1 < vm1) ? chi_shock : 1.0;
}
QK_DEV auto flatteningChi(Eos const &eos, double Pm2, double Pm1, double P, double Pp1, double Pp2, double rho, double vm1, double vp1) -> double
{
	return flatteningChiKS(Pm2, Pm1, Pp1, Pp2, flatteningKS(eos, rho, P), vm1, vp1);
}

// HydroState.hpp:10-23 (no scalars, no B field)
struct HState {
	double rho, u, v, w, P, cs, E, Eint;
};

// hydro_system.hpp:881-1003 : build the canonical (normal-first) Riemann state from one reconstructed
// primitive state q[6] = (rho, vx, vy, vz, P|e, Eint|e_aux).  DIR fixes (u,v,w) <- (vN, vV, vW) with the
// 3-D mapping X1:(x,y,z) X2:(y,z,x) X3:(z,x,y)  (:954-976).
// Rrho = recipOf(rho), Rg = recipOf((gamma-1) rho): shared with the divisions of hllc()
template <int DIR, bool TWOD = false> QK_DEV auto makeState(Eos const &eos, bool reconstruct_eint, const double q[NVAR], Recip const &Rrho, Recip const &Rg) -> HState
{
	HState s;
	const double rho = q[PRHO];
	const double vx = q[PVX], vy = q[PVY], vz = q[PVZ];
	const double ke = 0.5 * rho * (vx * vx + vy * vy + vz * vz);
	double P, Eint, cs, E;
	if (eos.isothermal) {
		P = rho * (eos.cs_iso * eos.cs_iso);
		cs = eos.cs_iso;
		// E_L, Eint_L stay NaN in the reference (:898-905); their fluxes are zeroed (:1084-1087)
		E = __builtin_nan("");
		Eint = __builtin_nan("");
	} else {
		if (reconstruct_eint) {
			// eos.pressure(rho, Eint): e = Eint / rho (0 if rho == 0), p = (gamma-1) rho e
			const double e = (rho == 0.0) ? 0.0 : divBy(q[PPRES] * rho, Rrho);
			P = eos.gm1 * rho * e;
			Eint = rho * q[PEINT];
		} else {
			P = q[PPRES];
			Eint = q[PEINT];
		}
		cs = sqrtN(divBy(eos.gamma * P, Rrho)); // eos.soundSpeed
		E = divBy(P, Rg) * rho + ke;	     // eos.eintFromPres
	}
	s.rho = rho;
	s.u = q[PVX + Axes<DIR, TWOD>::n];
	s.v = q[PVX + Axes<DIR, TWOD>::v];
	s.w = q[PVX + Axes<DIR, TWOD>::w];
	s.P = P;
	s.cs = cs;
	s.E = E;
	s.Eint = Eint;
	return s;
}

// What a passive scalar needs from the Riemann solve of its face: scalars ride on the same waves, with D = 0 in F = u U + P D
// (HLLC.hpp:126-136, LLF.hpp:30-41), and take the same artificial viscosity (hydro_system.hpp:1054-1076).
struct Wave {
	bool left, star;  // HLLC: side of the contact, inside the fan
	double u, S_K, S_star; // HLLC: normal velocity and outer wave speed of that side, contact speed
	Recip RD;	  // HLLC: 1 / (S_K - S_star)
	double uL, uR, hS; // LLF: normal velocities, 0.5 * Sp
	double viscosity;
};

// flux of a passive scalar with face states (qL, qR) through a face whose Riemann solve left `wv` behind
template <int RIEMANN> QK_DEV auto scalarFlux(Wave const &wv, double qL, double qR) -> double
{
	double Fc;
	if (RIEMANN == QK_RIEMANN_HLLD) {
		Fc = 0.0; // HLLD.hpp:331-332: the flux array is zero beyond the energy
	} else if (RIEMANN == QK_RIEMANN_HLLC) {
		const double U = wv.left ? qL : qR;
		const double F_K = wv.u * U;
		const double N = wv.S_star * (wv.S_K * U - F_K);
		Fc = wv.star ? divBy(N, wv.RD) : F_K;
	} else {
		Fc = 0.5 * (wv.uL * qL + wv.uR * qR) - wv.hS * (qR - qL);
	}
	return Fc + wv.viscosity * (qL - qR);
}

// HLLC.hpp:22-153. F[6] in canonical order (rho, mom_n, mom_v, mom_w, E, Eint).
QK_DEV void hllc(Eos const &eos, HState const &sL, HState const &sR, double du, double dw, double F[NVAR], Recip const &RL, Recip const &RR,
		 Recip const &GL, Recip const &GR, Wave *wv = nullptr)
{
	const double wl = sqrtN(sL.rho);
	const double wr = sqrtN(sR.rho);
	const double norm = recipExact(wl + wr);
	const double u_tilde = (wl * sL.u + wr * sR.u) * norm;
	const double dU = sL.u - sR.u;
	double S_L, S_R;
	if (!eos.isothermal) {
		const double v_tilde = (wl * sL.v + wr * sR.v) * norm;
		const double w_tilde = (wl * sL.w + wr * sR.w) * norm;
		const double vsq_tilde = u_tilde * u_tilde + v_tilde * v_tilde + w_tilde * w_tilde;
		const double H_L = divBy(sL.E + sL.P, RL);
		const double H_R = divBy(sR.E + sR.P, RR);
		const double H_tilde = (wl * H_L + wr * H_R) * norm;
		// ComputeOtherDerivatives (EOS.hpp:246-302) with the direct gamma-law forms:
		//   dedr = 0, dedp = 1/dpde = 1/((gamma-1) rho), drdp = 1/((p/rho) * k_B / k_B_user), G = (gamma+1)/2
		const double dedp_L = divBy(1.0, GL);
		const double dedp_R = divBy(1.0, GR);
		const double drdp_L = recipExact(divBy(divBy(sL.P, RL) * Eos::k_B, eos.RkBu));
		const double drdp_R = recipExact(divBy(divBy(sR.P, RR) * Eos::k_B, eos.RkBu));
		const double G = 0.5 * (1.0 + eos.gamma);
		const double eL = divBy(sL.Eint, RL);
		const double eR = divBy(sR.Eint, RR);
		const double C_tilde_rho = 0.5 * (eL + eR); // + rho*dedr with dedr = 0
		const double C_tilde_P = 0.5 * (eL * drdp_L + eR * drdp_R + sL.rho * dedp_L + sR.rho * dedp_R);
		const double cs_exp = H_tilde - 0.5 * vsq_tilde - C_tilde_rho;
		double cs_tilde;
		if (cs_exp <= 0) {
			cs_tilde = 0.5 * (sL.cs + sR.cs);
		} else {
			cs_tilde = sqrtN(divN(cs_exp, C_tilde_P));
		}
		const double s_NL = 0.5 * G * smax0(dU);
		const double s_NR = s_NL;
		S_L = smin(sL.u - (sL.cs + s_NL), u_tilde - (cs_tilde + s_NL));
		S_R = smax(sR.u + (sR.cs + s_NR), u_tilde + (cs_tilde + s_NR));
	} else {
		const double cs_tilde = 0.5 * (sL.cs + sR.cs);
		const double G_L = 0.5 * (1.0 + 1.);
		const double s_NL = 0.5 * G_L * smax0(dU);
		const double s_NR = s_NL;
		S_L = smin(sL.u - (sL.cs + s_NL), u_tilde - (cs_tilde + s_NL));
		S_R = smax(sR.u + (sR.cs + s_NR), u_tilde + (cs_tilde + s_NR));
	}

	// :91-93 carbuncle switch
	const double cs_max = smax(sL.cs, sR.cs);
	const double tp = smin1(divN(cs_max - smin0(du), cs_max - smin0(dw)));
	const double theta = tp * tp * tp * tp;

	// :97-98
	const double S_star =
	    divN(theta * (sR.P - sL.P) + (sL.rho * sL.u * (S_L - sL.u) - sR.rho * sR.u * (S_R - sR.u)), sL.rho * (S_L - sL.u) - sR.rho * (S_R - sR.u));

	// :102-107
	const double vmag_L = sqrtN(sL.u * sL.u + sL.v * sL.v + sL.w * sL.w);
	const double vmag_R = sqrtN(sR.u * sR.u + sR.v * sR.v + sR.w * sR.w);
	const double chi = smin1(divN(smax(vmag_L, vmag_R), cs_max));
	const double phi = chi * (2. - chi);
	const double P_LR = 0.5 * (sL.P + sR.P) + 0.5 * phi * (sL.rho * (S_L - sL.u) * (S_star - sL.u) + sR.rho * (S_R - sR.u) * (S_star - sR.u));

	// :142-150 fan selection, done before evaluating the fluxes
	const bool caseA = (S_L > 0.0);
	const bool caseB = !caseA && ((S_star > 0.0) && (S_L <= 0.0));
	const bool caseC = !caseA && !caseB && ((S_star <= 0.0) && (S_R >= 0.0));
	const bool left = caseA || caseB;
	const bool star = caseB || caseC;

	const double rho = left ? sL.rho : sR.rho;
	const double u = left ? sL.u : sR.u;
	const double v = left ? sL.v : sR.v;
	const double w = left ? sL.w : sR.w;
	const double P = left ? sL.P : sR.P;
	const double E = left ? sL.E : sR.E;
	const double Eint = left ? sL.Eint : sR.Eint;
	const double S_K = left ? S_L : S_R;

	// U_K (:120-121), F_K = u U_K + P D_K (:132-133)
	const double U0 = rho, U1 = rho * u, U2 = rho * v, U3 = rho * w, U4 = E, U5 = Eint;
	const double F0 = u * U0;
	const double F1 = u * U1 + P;
	const double F2 = u * U2;
	const double F3 = u * U3;
	const double F4 = u * U4 + P * u;
	const double F5 = u * U5;

	// F*_K = (S* (S_K U_K - F_K) + (S_K P_LR) D*) / (S_K - S*)   (:135-136)
	const double SKP = S_K * P_LR;
	const double den = S_K - S_star;
	const double N0 = S_star * (S_K * U0 - F0);
	const double N1 = S_star * (S_K * U1 - F1) + SKP;
	const double N2 = S_star * (S_K * U2 - F2);
	const double N3 = S_star * (S_K * U3 - F3);
	const double N4 = S_star * (S_K * U4 - F4) + SKP * S_star;
	const double N5 = S_star * (S_K * U5 - F5);
	const Recip RD = recipOf(den);
	const double G0 = divBy(N0, RD);
	const double G1 = divBy(N1, RD);
	const double G2 = divBy(N2, RD);
	const double G3 = divBy(N3, RD);
	const double G4 = divBy(N4, RD);
	const double G5 = divBy(N5, RD);

	F[0] = star ? G0 : F0;
	F[1] = star ? G1 : F1;
	F[2] = star ? G2 : F2;
	F[3] = star ? G3 : F3;
	F[4] = star ? G4 : F4;
	F[5] = star ? G5 : F5;
	if (wv != nullptr) {
		wv->left = left;
		wv->star = star;
		wv->u = u;
		wv->S_K = S_K;
		wv->S_star = S_star;
		wv->RD = RD;
	}
}

// HLLD (Miyoshi & Kusano 2005; reference src/hydro/HLLD.hpp) as the reference USES it: only through its MHD stub, which hands the solver
// bx = by = bz = 0 "for testing purposes" (hydro_system.hpp:987-1003, :1044-1048).  This is that B = 0 specialisation, derived by putting zero
// fields into the solver and dropping what then multiplies or adds an exact zero (x + 0, x - 0, 0 * finite; the sign of a zero result may differ):
//   * the fast magnetosonic speed collapses to sqrt(0.5 (gamma P + sqrt((gamma P)^2)) / rho);
//   * both Alfven speeds coincide with the contact speed S_M, so the double-star states never reach the interface and the five-wave fan is the
//     three-wave one: F = f_L | f_L + S_L (U*_L - U_L) | f_R + S_R (U*_R - U_R) | f_R;
//   * the star states keep the transverse velocities of their side, E* = (a E - P u + p* S_M) / (S_K - S_M) with a = S_K - u.
// The general solver stays restated in oracle/hydro.hpp; tests/test_hydro_ops_gpu.py::test_hlld_stub_fluxes_bit_exact_random_3d pins this function
// to it bit for bit.  Association of every retained operation follows the reference (the reciprocal of S_K - S_M is formed first and multiplied,
// HLLD.hpp:137-140).  F[6] in canonical order (rho, mom_n, mom_v, mom_w, E, 0): no flux of the auxiliary internal energy, none of the passive
// scalars (HLLD.hpp:331-332).
struct HlldSide {
	double U[5], f[5]; // conserved state and physical flux (rho, m_n, m_v, m_w, E)
	double a;	   // S_K - u_K
};
QK_DEV auto hlldFastSpeed(double gamma, HState const &s) -> double
{
	const double gp = gamma * s.P;
	return sqrt(0.5 * (gp + sqrt(gp * gp)) / s.rho);
}
QK_DEV void hlldSide(double gamma, HState const &s, HlldSide &k)
{
	const double ke = 0.5 * s.rho * (s.u * s.u + (s.v * s.v + s.w * s.w));
	k.U[0] = s.rho;
	k.U[1] = s.u * s.rho;
	k.U[2] = s.v * s.rho;
	k.U[3] = s.w * s.rho;
	k.U[4] = ke + s.P / (gamma - 1.0);
	k.f[0] = k.U[1];
	k.f[1] = k.U[1] * s.u + s.P;
	k.f[2] = k.U[2] * s.u;
	k.f[3] = k.U[3] * s.u;
	k.f[4] = s.u * (k.U[4] + s.P);
}
// f_K + S_K (U*_K - U_K): the flux on the K side of the contact
QK_DEV void hlldStarFlux(HState const &s, HlldSide const &k, double S_K, double S_M, double rhoStar, double invSM, double pStar, double F[NVAR])
{
	double Ustar[5];
	Ustar[0] = rhoStar;
	Ustar[1] = rhoStar * S_M;
	Ustar[2] = rhoStar * s.v;
	Ustar[3] = rhoStar * s.w;
	Ustar[4] = (k.a * k.U[4] - s.P * s.u + pStar * S_M) * invSM;
#pragma unroll
	for (int n = 0; n < 5; ++n) {
		F[n] = k.f[n] + S_K * (Ustar[n] - k.U[n]);
	}
}
QK_DEV void hlldHydro(HState const &sL, HState const &sR, const double gamma, double F[NVAR])
{
	HlldSide L, R;
	hlldSide(gamma, sL, L);
	hlldSide(gamma, sR, R);
	const double cL = hlldFastSpeed(gamma, sL), cR = hlldFastSpeed(gamma, sR);
	const double S_L = smin(sL.u - cL, sR.u - cR);
	const double S_R = smax(sL.u + cL, sR.u + cR);
	L.a = S_L - sL.u;
	R.a = S_R - sR.u;
	const double S_M = (R.a * R.U[1] - L.a * L.U[1] + (sL.P - sR.P)) / (R.a * R.U[0] - L.a * L.U[0]);
	const double invL = 1.0 / (S_L - S_M), invR = 1.0 / (S_R - S_M);
	const double rhoStarL = L.U[0] * L.a * invL, rhoStarR = R.U[0] * R.a * invR;
	const double pStar = 0.5 * ((sL.P - L.U[0] * L.a * (S_M - sL.u)) + (sR.P - R.U[0] * R.a * (S_M - sR.u)));
	F[5] = 0.0;
	if (S_L >= 0.0) {
#pragma unroll
		for (int n = 0; n < 5; ++n) {
			F[n] = L.f[n];
		}
	} else if (S_R <= 0.0) {
#pragma unroll
		for (int n = 0; n < 5; ++n) {
			F[n] = R.f[n];
		}
	} else if (S_M >= 0.0) {
		hlldStarFlux(sL, L, S_L, S_M, rhoStarL, invL, pStar, F);
		if (!(rhoStarL > 0.0)) { // the left Alfven speed S_M - 0 / sqrt(rho*_L) is not a number: every component of the reference's sum is NaN
#pragma unroll
			for (int n = 0; n < 5; ++n) {
				F[n] = __builtin_nan("");
			}
		}
	} else {
		hlldStarFlux(sR, R, S_R, S_M, rhoStarR, invR, pStar, F);
	}
}

// LLF.hpp:16-43
QK_DEV void llf(HState const &sL, HState const &sR, double F[NVAR], Wave *wv = nullptr)
{
	const double Sp = smax(fabs(sL.u) + sL.cs, fabs(sR.u) + sR.cs);
	if (wv != nullptr) {
		wv->uL = sL.u;
		wv->uR = sR.u;
		wv->hS = 0.5 * Sp;
	}
	const double UL[NVAR] = {sL.rho, sL.rho * sL.u, sL.rho * sL.v, sL.rho * sL.w, sL.E, sL.Eint};
	const double UR[NVAR] = {sR.rho, sR.rho * sR.u, sR.rho * sR.v, sR.rho * sR.w, sR.E, sR.Eint};
	double FL[NVAR], FR[NVAR];
	FL[0] = sL.u * UL[0];
	FL[1] = sL.u * UL[1] + sL.P;
	FL[2] = sL.u * UL[2];
	FL[3] = sL.u * UL[3];
	FL[4] = sL.u * UL[4] + sL.P * sL.u;
	FL[5] = sL.u * UL[5];
	FR[0] = sR.u * UR[0];
	FR[1] = sR.u * UR[1] + sR.P;
	FR[2] = sR.u * UR[2];
	FR[3] = sR.u * UR[3];
	FR[4] = sR.u * UR[4] + sR.P * sR.u;
	FR[5] = sR.u * UR[5];
	const double hS = 0.5 * Sp;
#pragma unroll
	for (int n = 0; n < NVAR; ++n) {
		F[n] = 0.5 * (FL[n] + FR[n]) - hS * (UR[n] - UL[n]);
	}
}

// hydro_system.hpp:1037-1110 : Riemann solve + artificial viscosity + momentum un-permutation + face velocity.
// qL/qR: reconstructed primitive states at the face; du, dvl.. the velocity differences of :1019-1034.
// Fout[6] is in ARRAY component order (rho, px, py, pz, E, Eint).
template <int DIR, int RIEMANN, bool TWOD = false>
QK_DEV void faceFlux(Eos const &eos, bool reconstruct_eint, int ndim, const double qL[NVAR], const double qR[NVAR], double du, double dvl, double dvr,
		     double dwl, double dwr, double K_visc, double Fout[NVAR], double &v_norm, Wave *wv = nullptr)
{
	const Recip RL = recipOf(qL[PRHO]), RR = recipOf(qR[PRHO]);
	const Recip GL = recipOf(eos.gm1 * qL[PRHO]), GR = recipOf(eos.gm1 * qR[PRHO]);
	const HState sL = makeState<DIR, TWOD>(eos, reconstruct_eint, qL, RL, GL);
	const HState sR = makeState<DIR, TWOD>(eos, reconstruct_eint, qR, RR, GR);
	double dw = 0.;
	if (ndim >= 2) {
		dw = smin(dvl, dvr);
	}
	if (ndim == 3) {
		dw = smin(smin(dwl, dwr), dw);
	}
	double Fc[NVAR];
	if (RIEMANN == QK_RIEMANN_HLLC) {
		hllc(eos, sL, sR, du, dw, Fc, RL, RR, GL, GR, wv);
	} else if (RIEMANN == QK_RIEMANN_HLLD) {
		hlldHydro(sL, sR, eos.gamma, Fc); // hydro_system.hpp:987-1003, :1047: the reference hands HLLD bx = by = bz = 0
	} else {
		llf(sL, sR, Fc, wv);
	}
	// :1054-1076 artificial viscosity (momentum components are overwritten below, :1079-1081)
	double div_v = du;
	if (ndim >= 2) {
		div_v = div_v + 0.5 * (dvl + dvr);
	}
	if (ndim == 3) {
		div_v = div_v + 0.5 * (dwl + dwr);
	}
	const double viscosity = K_visc * smax0(-div_v);
	if (wv != nullptr) {
		wv->viscosity = viscosity;
	}
	double F[NVAR];
	F[RHO] = Fc[0] + viscosity * (sL.rho - sR.rho);
	F[ENE] = Fc[4] + viscosity * (sL.E - sR.E);
	F[EINT] = Fc[5] + viscosity * (sL.Eint - sR.Eint);
	F[MX + Axes<DIR, TWOD>::n] = Fc[1];
	F[MX + Axes<DIR, TWOD>::v] = Fc[2];
	F[MX + Axes<DIR, TWOD>::w] = Fc[3];
	if (eos.isothermal) {
		F[ENE] = 0;
		F[EINT] = 0;
	}
	// :1090
	v_norm = (F[RHO] >= 0.) ? divBy(F[RHO], RR) : divBy(F[RHO], RL);
#pragma unroll
	for (int n = 0; n < NVAR; ++n) {
		Fout[n] = F[n];
	}
}

// hydro_system.hpp:702-771 EnforceLimits for one cell (no scalars). U in array order.
QK_DEV void enforceLimits(Eos const &eos, double densityFloor, double tempFloor, double U[NVAR])
{
	double rho_new = U[RHO];
	if (U[RHO] < densityFloor) {
		rho_new = densityFloor;
		U[RHO] = rho_new;
	}
	if ((rho_new > 2.2250738585072014e-308) && !eos.isothermal) {
		const double vx1 = U[MX] / rho_new;
		const double vx2 = U[MY] / rho_new;
		const double vx3 = U[MZ] / rho_new;
		const double Ekin = 0.5 * rho_new * (vx1 * vx1 + vx2 * vx2 + vx3 * vx3);
		const double Etot = U[ENE];
		const double primTemp = eos.tgasFromEint(rho_new, (Etot - Ekin));
		if (primTemp < tempFloor) {
			const double prim_eint = eos.eintFromTgas(rho_new, tempFloor);
			U[ENE] = Ekin + prim_eint;
		}
		const double auxEint = U[EINT];
		const double auxTemp = eos.tgasFromEint(rho_new, auxEint);
		if (auxTemp < tempFloor) {
			U[EINT] = eos.eintFromTgas(rho_new, tempFloor);
		}
	}
}

// hydro_system.hpp:825-849 SyncDualEnergy for one cell; returns false if rho <= 0 (fatal in the reference)
QK_DEV auto syncDualEnergy(double U[NVAR]) -> bool
{
	const double eta = 1.0e-3;
	const double rho = U[RHO];
	if (rho <= 0.) {
		return false;
	}
	const double px = U[MX], py = U[MY], pz = U[MZ];
	const double Etot = U[ENE];
	const double Eint_aux = U[EINT];
	const double Ekin = (px * px + py * py + pz * pz) / (2.0 * rho);
	const double Eint_cons = Etot - Ekin;
	if (Eint_cons > eta * Etot) {
		U[EINT] = Eint_cons;
	} else {
		U[EINT] = Eint_aux;
		U[ENE] = Eint_aux + Ekin;
	}
	return true;
}

// which = 0: maxSignalSpeedLocal (hydro_system.hpp:206-219); which = 1: ComputeMaxSignalSpeed (:227-250)
QK_DEV auto signalSpeed(Eos const &eos, int which, double rho, double px, double py, double pz, double E) -> double
{
	double cs;
	if (eos.isothermal) {
		cs = eos.cs_iso;
	} else {
		// ComputeSoundSpeed(cons,i,j,k) (:374-394): P from (rho, E - KE) then cs(rho, P)
		const double vx = px / rho;
		const double vy = py / rho;
		const double vz = pz / rho;
		const double kinetic_energy = 0.5 * rho * (vx * vx + vy * vy + vz * vz);
		const double thermal_energy = E - kinetic_energy;
		const double P = eos.pressure(rho, thermal_energy);
		cs = eos.soundSpeed(rho, P);
	}
	if (which == 0) {
		const double kinetic_energy = (px * px + py * py + pz * pz) / (2.0 * rho);
		const double abs_vel = sqrt(2.0 * kinetic_energy / rho);
		return cs + abs_vel;
	}
	const double vx = px / rho;
	const double vy = py / rho;
	const double vz = pz / rho;
	const double vel_mag = sqrt(vx * vx + vy * vy + vz * vz);
	return cs + vel_mag;
}


// atomic max on a non-negative double through its (order-preserving) bit pattern.  NaNs never win a
// std::max(result, v) comparison in the reference's serial reduction, so they are skipped here too.
QK_DEV void atomicMaxNonNeg(double *addr, double v)
{
	if (v > 0.0) {
		atomicMax(reinterpret_cast<unsigned long long *>(addr), static_cast<unsigned long long>(__double_as_longlong(v)));
	}
}

} // namespace qk

#endif // QK_DEVICE_HPP_
