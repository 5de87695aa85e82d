// qk_cooling_device.hpp — optically-thin cooling from Cloudy tables, per cell (host + device).
//
// What the reference does (src/cooling/TabulatedCooling.hpp): the Strang-split source integrates dE_int/dt = n_H^2 (Gamma - Lambda)(n_H, T) over
// dt/2 in every cell with an adaptive Heun integrator (src/math/ODEIntegrate.hpp); T follows from E_int through the tabulated mean molecular
// weight mu(n_H, T), an implicit relation solved with Alefeld, Potra & Shi's Algorithm 748 (ACM TOMS 21, 327 (1995); src/math/root_finding.hpp)
// to a relative bracket width of 1e-5 — the returned temperature is the MIDPOINT of the final bracket, so the sequence of bracketing points is
// part of the result and is followed here step for step (same interpolation points, same guards, same order of the floating-point operations).
// Functions are QK_HD: the library's kernel (qk_cooling.hip) and the host mirror's quokka::TabulatedCooling functions, which the problem files call
// from their own device lambdas (host/compat/tabulated_cooling.hpp), share them.
#ifndef QK_COOLING_DEVICE_HPP_
#define QK_COOLING_DEVICE_HPP_

#include <cfloat>
#include <cmath>

#ifndef QK_HD
#if defined(__HIPCC__)
#define QK_HD __host__ __device__ inline
#else
#define QK_HD inline
#endif
#endif

namespace qk
{
namespace cool
{

// the tables as the kernels see them: log10 n_H (n_nH values, uniformly spaced), log10 T (n_T values), and three n_nH x n_T tables with the
// n_H index running fastest — the layout of the reference's transposed TableData (CloudyDataReader.cpp:206-222)
struct Tables {
	const double *log_nH;
	const double *log_T;
	const double *cool; // FastMath::log10 of Lambda / CoolUnit
	const double *heat;
	const double *mmw; // dimensionless mean molecular weight
	int n_nH, n_T;
	double T_min, T_max;
	double mmw_min, mmw_max;
	// constants of fundamental_constants.H the relations use: m_p + m_e and k_B (cgs)
	double m_H, k_B;
	// filled by prepare(): the ends and the spacing of the two axes — what interpolate2d recomputes from the axis arrays in every call
	double xi, xf, yi, yf;
	double dx, dy, rdx, rdy; // spacing and (device) its refined reciprocal
	int prepared;
};

// "abundances ism" of Cloudy: n_He / n_H = 0.098 (TabulatedCooling.hpp:32)
constexpr double H_mass_fraction = 1. / (1. + 0.098 * 3.971);

QK_HD auto clampd(double v, double lo, double hi) -> double { return (v < lo) ? lo : (hi < v) ? hi : v; }
QK_HD auto clampi(int v, int lo, int hi) -> int { return (v < lo) ? lo : (hi < v) ? hi : v; }
QK_HD auto signOf(double v) -> int { return (0.0 < v) - (v < 0.0); }
QK_HD auto isNan(double v) -> bool { return v != v; }

// FastMath::pow10 (src/math/FastMath.hpp:42-49,68-72): 2^x with the fractional part of x taken linearly
QK_HD auto fastPow10(double x) -> double
{
	constexpr double LOG10OLOG2 = 3.321928094887362626;
	const double x2 = LOG10OLOG2 * x;
	const int flr = static_cast<int>(floor(x2)); // (the reference stores std::floor in an int)
	const double remainder = x2 - flr;
	const double mantissa = 0.5 * (remainder + 1);
	return ldexp(mantissa, flr + 1);
}
// FastMath::log10 (FastMath.hpp:33-40,62-66): frexp's exponent + the mantissa taken linearly
QK_HD auto fastLog10(double x) -> double
{
	constexpr double LOG2OLOG10 = 0.301029995663981195;
	int n = 0;
	const double y = frexp(x, &n);
	return LOG2OLOG10 * (2 * (y - 1) + n);
}

// Quotients.  The reference divides; a correctly rounded quotient does not depend on how it is formed.  On the device a divisor shared by several
// quotients (the cell volume of the four interpolation weights, the axis spacing) is inverted once — v_rcp_f64 refined by two Newton steps — and
// each quotient is the product with that reciprocal corrected by its fma residual: the IEEE quotient for operands in the normal range (all of them
// here are differences of table ordinates, O(0.01 ... 10)), 3 instructions instead of the ~35 of a division.  The host divides.
struct Den {
	double d, r;
};
QK_HD auto denOf(double d) -> Den
{
	Den D;
	D.d = d;
#if defined(__HIP_DEVICE_COMPILE__)
	const double r0 = __builtin_amdgcn_rcp(d);
	double e = __builtin_fma(-d, r0, 1.0);
	const double r1 = __builtin_fma(r0, e, r0);
	e = __builtin_fma(-d, r1, 1.0);
	D.r = __builtin_fma(r1, e, r1);
#else
	D.r = 0.0;
#endif
	return D;
}
QK_HD auto over(double n, Den const &D) -> double
{
#if defined(__HIP_DEVICE_COMPILE__)
	const double q = n * D.r;
	const double e = __builtin_fma(-D.d, q, n);
	return __builtin_fma(e, D.r, q);
#else
	return n / D.d;
#endif
}

QK_HD auto quo(double n, double d) -> double { return over(n, denOf(d)); } // a lone quotient: 11 instructions on the device

// the ends and the spacing of the axes (Interpolate2D.hpp:18-24), once per kernel thread / host call instead of once per look-up
QK_HD void prepare(Tables &t)
{
	t.xi = t.log_nH[0];
	t.xf = t.log_nH[t.n_nH - 1];
	t.yi = t.log_T[0];
	t.yf = t.log_T[t.n_T - 1];
	t.dx = (t.xf - t.xi) / static_cast<double>(t.n_nH - 1);
	t.dy = (t.yf - t.yi) / static_cast<double>(t.n_T - 1);
	t.rdx = denOf(t.dx).r;
	t.rdy = denOf(t.dy).r;
	t.prepared = 1;
}
QK_HD auto preparedCopy(Tables const &t) -> Tables
{
	Tables p = t;
	if (p.prepared != 1) {
		prepare(p);
	}
	return p;
}

// interpolate2d (src/math/Interpolate2D.hpp:14-80) on a uniformly spaced table, in three parts: the position on one axis (clamped ordinate, the
// two indices, their ordinates), the four weights of a point, the weighted sum over a table — so that the density axis is resolved once per cell,
// and the weights once per point for the cooling and the heating table.  The reference's degenerate-cell tests compare the first ORDINATE `yi` with
// the upper INDEX `iiy`; they are kept as written, since they decide the weights on the last row / column of the table.
struct AxisPos {
	int i0, i1;
	double v, v1, v2; // the clamped ordinate and the ordinates of the two indices
};
QK_HD auto axisPos(const double *av, int n, double lo, double hi, double da, double rda, double a) -> AxisPos
{
	AxisPos p;
	p.v = clampd(a, lo, hi);
	const Den D{da, rda};
	p.i0 = clampi(static_cast<int>(floor(over(p.v - lo, D))), 0, n - 1);
	p.i1 = (p.i0 == n - 1) ? p.i0 : p.i0 + 1;
	p.v1 = av[p.i0];
	p.v2 = av[p.i1];
	return p;
}
struct Weights {
	int ix, iix, iy, iiy;
	double w11, w12, w21, w22;
};
QK_HD auto weightsOf(Tables const &t, AxisPos const &X, AxisPos const &Y) -> Weights
{
	Weights W;
	W.ix = X.i0;
	W.iix = X.i1;
	W.iy = Y.i0;
	W.iiy = Y.i1;
	W.w11 = 0;
	W.w12 = 0;
	W.w21 = 0;
	W.w22 = 0;
	const double x = X.v, x1 = X.v1, x2 = X.v2, y = Y.v, y1 = Y.v1, y2 = Y.v2;
	const bool yEdge = (t.yi == static_cast<double>(W.iiy)); // sic
	if (W.ix != W.iix && W.iy != W.iiy) {
		const Den vol = denOf((x2 - x1) * (y2 - y1));
		W.w11 = over((x2 - x) * (y2 - y), vol);
		W.w12 = over((x2 - x) * (y - y1), vol);
		W.w21 = over((x - x1) * (y2 - y), vol);
		W.w22 = over((x - x1) * (y - y1), vol);
	} else if (W.ix == W.iix && !yEdge) {
		const Den vol = denOf(y2 - y1);
		W.w11 = over(y2 - y, vol);
		W.w12 = over(y - y1, vol);
	} else if (W.ix != W.iix && yEdge) {
		const Den vol = denOf(x2 - x1);
		W.w11 = over(x2 - x, vol);
		W.w21 = over(x - x1, vol);
	} else {
		W.w11 = 1.0;
	}
	return W;
}
QK_HD auto weighted(Tables const &t, const double *table, Weights const &W) -> double
{
	const int nx = t.n_nH;
	const double A = table[W.ix + nx * W.iy];
	const double B = table[W.ix + nx * W.iiy];
	const double C = table[W.iix + nx * W.iy];
	const double D = table[W.iix + nx * W.iiy];
	return W.w11 * A + W.w12 * B + W.w21 * C + W.w22 * D;
}
QK_HD auto densityAxis(Tables const &t, double log_nH) -> AxisPos { return axisPos(t.log_nH, t.n_nH, t.xi, t.xf, t.dx, t.rdx, log_nH); }
QK_HD auto temperatureAxis(Tables const &t, double log_T) -> AxisPos { return axisPos(t.log_T, t.n_T, t.yi, t.yf, t.dy, t.rdy, log_T); }
// (t prepared)
QK_HD auto lookup(Tables const &t, const double *table, AxisPos const &X, double log_T) -> double
{
	return weighted(t, table, weightsOf(t, X, temperatureAxis(t, log_T)));
}

// What the functions below recompute from the density alone, formed once per cell by the integrator (the same expressions: the same bits):
// rho X, log10 n_H and its position on the density axis, the energies at the ends of the temperature axis, k_B rho / m_H
struct CellCool {
	double rho, gamma, rhoH, log_nH, Emin, Emax, kBn;
	AxisPos X;
};

// cloudy_cooling_function (TabulatedCooling.hpp:82-99): net heating rate per volume, (rho X)^2 (10^heat - 10^cool); one set of weights serves both
// tables (t prepared)
QK_HD auto netHeatingAt(Tables const &t, double rhoH, AxisPos const &X, double T) -> double
{
	const Weights W = weightsOf(t, X, temperatureAxis(t, log10(T)));
	const double logCool = weighted(t, t.cool, W);
	const double logHeat = weighted(t, t.heat, W);
	const double netLambda = fastPow10(logHeat) - fastPow10(logCool);
	return (rhoH * rhoH) * netLambda;
}
QK_HD auto netHeating(Tables const &t0, double rho, double T) -> double
{
	const Tables t = preparedCopy(t0);
	const double rhoH = rho * H_mass_fraction;
	const double nH = rhoH / t.m_H;
	return netHeatingAt(t, rhoH, densityAxis(t, log10(nH)), T);
}

// ComputeEgasFromTgas (TabulatedCooling.hpp:101-115) (t prepared)
QK_HD auto egasFromTgasAt(Tables const &t, double rho, AxisPos const &X, double Tgas, double gamma) -> double
{
	const double mu = lookup(t, t.mmw, X, log10(Tgas));
	const double n = rho / (t.m_H * mu);
	const double Pgas = n * t.k_B * Tgas;
	return Pgas / (gamma - 1.);
}
QK_HD auto egasFromTgas(Tables const &t0, double rho, double Tgas, double gamma) -> double
{
	const Tables t = preparedCopy(t0);
	const double rhoH = rho * H_mass_fraction;
	const double nH = rhoH / t.m_H;
	return egasFromTgasAt(t, rho, densityAxis(t, log10(nH)), Tgas, gamma);
}
// (t prepared)
QK_HD auto cellCool(Tables const &t, double rho, double gamma) -> CellCool
{
	CellCool c;
	c.rho = rho;
	c.gamma = gamma;
	c.rhoH = rho * H_mass_fraction;
	c.log_nH = log10(c.rhoH / t.m_H);
	c.X = densityAxis(t, c.log_nH);
	c.Emin = egasFromTgasAt(t, rho, c.X, t.T_min, gamma);
	c.Emax = egasFromTgasAt(t, rho, c.X, t.T_max, gamma);
	c.kBn = t.k_B * (rho / t.m_H);
	return c;
}

// ---- Algorithm 748: a bracket [a, b] with f(a) f(b) < 0 and the two points that left it last (d, e)
struct Bracket748 {
	double a, b, fa, fb, d, fd, e, fe;

	// secant point, pulled to the midpoint when it falls within 5 eps of an end (root_finding.hpp:134-145)
	QK_HD auto secantPoint() const -> double
	{
		const double tol = DBL_EPSILON * 5;
		const double c = a - quo(fa, fb - fa) * (b - a);
		if ((c <= a + fabs(a) * tol) || (c >= b - fabs(b) * tol)) {
			return (a + b) / 2;
		}
		return c;
	}
	// quotient that returns r instead of overflowing (root_finding.hpp:120-132)
	QK_HD static auto guardedQuotient(double num, double denom, double r) -> double
	{
		if (fabs(denom) < 1) {
			if (fabs(denom * DBL_MAX) <= fabs(num)) {
				return r;
			}
		}
		return quo(num, denom);
	}
	// zero of the parabola through (a, b, d) by `count` Newton steps from the end where it has the sign of the curvature (root_finding.hpp:147-176)
	QK_HD auto parabolaPoint(unsigned count) const -> double
	{
		const double B = guardedQuotient(fb - fa, b - a, DBL_MAX);
		double A = guardedQuotient(fd - fb, d - b, DBL_MAX);
		A = guardedQuotient(A - B, d - a, 0.0);
		if (A == 0) {
			return secantPoint();
		}
		double c = (signOf(A) * signOf(fa) > 0) ? a : b;
		for (unsigned i = 1; i <= count; ++i) {
			c -= guardedQuotient(fa + (B + A * (c - b)) * (c - a), B + A * (2 * c - a - b), 1 + c - a);
		}
		if ((c <= a) || (c >= b)) {
			c = secantPoint();
		}
		return c;
	}
	// inverse cubic interpolation through (a, b, d, e) (root_finding.hpp:178-208)
	QK_HD auto inverseCubicPoint() const -> double
	{
		// (nine quotients over six distinct denominators)
		const Den Dfdfb = denOf(fd - fb), Dfbfa = denOf(fb - fa), Dfdfa = denOf(fd - fa);
		const double q11 = quo((d - e) * fd, fe - fd);
		const double q21 = over((b - d) * fb, Dfdfb);
		const double q31 = over((a - b) * fa, Dfbfa);
		const double d21 = over((b - d) * fd, Dfdfb);
		const double d31 = over((a - b) * fb, Dfbfa);
		const double q22 = quo((d21 - q11) * fb, fe - fb);
		const double q32 = over((d31 - q21) * fa, Dfdfa);
		const double d32 = over((d31 - q21) * fd, Dfdfa);
		const double q33 = quo((d32 - q22) * fa, fe - fa);
		double c = q31 + q32 + q33 + a;
		if ((c <= a) || (c >= b)) {
			c = parabolaPoint(3);
		}
		return c;
	}
	// two function values closer than 32 denormal minima: the cubic's divided differences would overflow (root_finding.hpp:272-274)
	QK_HD auto valuesCoincide() const -> bool
	{
		const double min_diff = DBL_MIN * 32;
		return (fabs(fa - fb) < min_diff) || (fabs(fa - fd) < min_diff) || (fabs(fa - fe) < min_diff) || (fabs(fb - fd) < min_diff) ||
		       (fabs(fb - fe) < min_diff) || (fabs(fd - fe) < min_diff);
	}
	// evaluate f at c (kept 2 eps inside the bracket) and keep the half that still brackets the root (root_finding.hpp:84-118)
	template <class F> QK_HD void shrinkAt(F const &f, double c)
	{
		const double tol = DBL_EPSILON * 2;
		if ((b - a) < 2 * tol * a) {
			c = a + (b - a) / 2;
		} else if (c <= a + fabs(a) * tol) {
			c = a + fabs(a) * tol;
		} else if (c >= b - fabs(b) * tol) {
			c = b - fabs(b) * tol;
		}
		const double fc = f(c);
		if (fc == 0) {
			a = c;
			fa = 0;
			d = 0;
			fd = 0;
			return;
		}
		if (signOf(fa) * signOf(fc) < 0) {
			d = b;
			fd = fb;
			b = c;
			fb = fc;
		} else {
			d = a;
			fd = fa;
			a = c;
			fa = fc;
		}
	}
	QK_HD auto narrow(double eps) const -> bool { return fabs(a - b) <= (eps * fmin(fabs(a), fabs(b))); }
};

// toms748_solve(f, ax, bx, tol, max_iter) (root_finding.hpp:212-340): on return [lo, hi] brackets the root, iterations = function evaluations used
// (the two at the ends included).  ax < bx and f(ax) f(bx) <= 0 are the caller's responsibility, as in the reference (which asserts them).
template <class F> QK_HD void solve748(F const &f, double ax, double bx, double eps, int max_iter, double &lo, double &hi, int &iterations)
{
	int budget = max_iter - 2;
	int count = budget;
	Bracket748 s;
	s.a = ax;
	s.b = bx;
	s.fa = f(ax);
	s.fb = f(bx);
	if (s.narrow(eps) || (s.fa == 0) || (s.fb == 0)) {
		if (s.fa == 0) {
			s.b = s.a;
		} else if (s.fb == 0) {
			s.a = s.b;
		}
		lo = s.a;
		hi = s.b;
		iterations = 2;
		return;
	}
	s.fe = s.e = s.fd = 1e5F;
	s.d = 0; // (never read before the first shrinkAt sets it)
	// two opening steps: secant, then the parabola
	s.shrinkAt(f, s.secantPoint());
	--count;
	if (count != 0 && (s.fa != 0) && !s.narrow(eps)) {
		const double c = s.parabolaPoint(2);
		s.e = s.d;
		s.fe = s.fd;
		s.shrinkAt(f, c);
		--count;
	}
	while (count != 0 && (s.fa != 0) && !s.narrow(eps)) {
		const double a0 = s.a, b0 = s.b;
		// two interpolation steps: the cubic where its divided differences exist, else the parabola (2, then 3 Newton steps)
		double c = s.valuesCoincide() ? s.parabolaPoint(2) : s.inverseCubicPoint();
		s.e = s.d;
		s.fe = s.fd;
		s.shrinkAt(f, c);
		if ((0 == --count) || (s.fa == 0) || s.narrow(eps)) {
			break;
		}
		c = s.valuesCoincide() ? s.parabolaPoint(3) : s.inverseCubicPoint();
		s.shrinkAt(f, c);
		if ((0 == --count) || (s.fa == 0) || s.narrow(eps)) {
			break;
		}
		// a double-length secant step from the end with the smaller |f|, not farther than half the bracket
		double u, fu;
		if (fabs(s.fa) < fabs(s.fb)) {
			u = s.a;
			fu = s.fa;
		} else {
			u = s.b;
			fu = s.fb;
		}
		c = u - 2 * quo(fu, s.fb - s.fa) * (s.b - s.a);
		if (fabs(c - u) > (s.b - s.a) / 2) {
			c = s.a + (s.b - s.a) / 2;
		}
		s.e = s.d;
		s.fe = s.fd;
		s.shrinkAt(f, c);
		if ((0 == --count) || (s.fa == 0) || s.narrow(eps)) {
			break;
		}
		// the three steps together must halve the bracket; bisect when they did not
		if ((s.b - s.a) < 0.5 * (b0 - a0)) {
			continue;
		}
		s.e = s.d;
		s.fe = s.fd;
		s.shrinkAt(f, s.a + (s.b - s.a) / 2);
		--count;
	}
	if (s.fa == 0) {
		s.b = s.a;
	} else if (s.fb == 0) {
		s.a = s.b;
	}
	lo = s.a;
	hi = s.b;
	iterations = (budget - count) + 2;
}

// ComputeTgasFromEgas (TabulatedCooling.hpp:117-174): T with mu(n_H, T) C = T, C = (gamma - 1) E / (k_B rho / m_H); NaN when the bracket is empty or
// the iteration limit is reached
QK_HD auto tgasInTable(Tables const &t, CellCool const &c, double Egas) -> double // (Emin < Egas < Emax)
{
	const double C = (c.gamma - 1.) * Egas / c.kBn;
	const double reltol = 1.0e-5;
	const int maxIterLimit = 100;
	auto f = [&](double T) {
		const double log_T = clampd(log10(T), 1., 9.);
		const double mu = lookup(t, t.mmw, c.X, log_T);
		return C * mu - T;
	};
	const double T_lo = clampd(C * t.mmw_min, t.T_min, t.T_max);
	const double T_hi = clampd(C * t.mmw_max, t.T_min, t.T_max);
	double T_sol = NAN;
	if (T_lo < T_hi) {
		double lo, hi;
		int used = 0;
		solve748(f, T_lo, T_hi, reltol, maxIterLimit, lo, hi, used);
		T_sol = 0.5 * (lo + hi);
		if ((used >= maxIterLimit) || isNan(T_sol)) {
			T_sol = NAN;
		}
	}
	return T_sol;
}
QK_HD auto tgasFromEgas(Tables const &t0, double rho, double Egas, double gamma) -> double
{
	const Tables t = preparedCopy(t0);
	const CellCool c = cellCool(t, rho, gamma);
	if (Egas <= c.Emin) {
		return t.T_min;
	}
	if (Egas >= c.Emax) {
		return t.T_max;
	}
	return tgasInTable(t, c, Egas);
}

// ComputeMMW (TabulatedCooling.hpp:206-220)
QK_HD auto meanMolecularWeight(Tables const &t0, double rho, double Egas, double gamma) -> double
{
	const Tables t = preparedCopy(t0);
	const double Tgas = tgasFromEgas(t, rho, Egas, gamma);
	const double rhoH = rho * H_mass_fraction;
	const double nH = rhoH / t.m_H;
	return lookup(t, t.mmw, densityAxis(t, log10(nH)), log10(Tgas));
}

// ComputeCoolingLength (TabulatedCooling.hpp:176-204): c_s t_cool with the cooling part of the table only
QK_HD auto coolingLength(Tables const &t0, double rho, double Egas, double gamma) -> double
{
	const Tables t = preparedCopy(t0);
	const double Tgas = tgasFromEgas(t, rho, Egas, gamma);
	const double rhoH = rho * H_mass_fraction;
	const double nH = rhoH / t.m_H;
	const Weights W = weightsOf(t, densityAxis(t, log10(nH)), temperatureAxis(t, log10(Tgas)));
	const double logCool = weighted(t, t.cool, W);
	const double LambdaCool = fastPow10(logCool);
	const double Edot = (rhoH * rhoH) * LambdaCool;
	const double t_cool = Egas / Edot;
	const double mu = weighted(t, t.mmw, W);
	const double c_s = sqrt(gamma * t.k_B * Tgas / (mu * t.m_H));
	return c_s * t_cool;
}

// user_rhs (TabulatedCooling.hpp:222-256): dE_int/dt; false when the temperature iteration failed
QK_HD auto heatingRate(Tables const &t, CellCool const &c, double Eint, double &rate) -> bool
{
	if (Eint <= c.Emin) {
		rate = netHeatingAt(t, c.rhoH, c.X, t.T_min);
	} else if (Eint >= c.Emax) {
		rate = netHeatingAt(t, c.rhoH, c.X, t.T_max);
	} else {
		const double T = tgasInTable(t, c, Eint);
		if (isNan(T)) {
			rate = NAN;
			return false;
		}
		rate = netHeatingAt(t, c.rhoH, c.X, T);
	}
	return true;
}

constexpr int maxSubsteps = 2000; // maxStepsODEIntegrate (ODEIntegrate.hpp:122)

// rk_adaptive_integrate with rk12_single_step for one unknown (ODEIntegrate.hpp:20-53,105-226): Heun's method with the embedded Euler solution,
// step-size factor eps^(-1/2) limited to 20 after a clean step, 1 after a retry, 0.3 on the second failure, [0.1, 0.3] afterwards, 0.5 after a
// failed right-hand side; 7 attempts per step, maxSubsteps steps.  Written as a state and ONE ATTEMPT at a time, so that the kernel can keep every
// lane of a wave busy with an attempt of ITS cell (qk_cooling.hip: a lane that finishes a cell takes the next one from a queue) — the sequence
// of operations on a cell is the reference's loop nest.
struct HeunState {
	double E, time, dt;
	int step, retry;
	int nsteps; // >= 0: finished (the number of steps; maxSubsteps = failure)
};
QK_HD void heunBegin(Tables const &t, CellCool const &c, double E0, double dt_total, HeunState &s)
{
	double rate0 = NAN;
	heatingRate(t, c, E0, rate0);
	const double dt_guess = 0.1 * fabs(E0 / rate0);
	s.E = E0;
	s.time = 0;
	s.dt = isNan(dt_guess) ? dt_total : dt_guess;
	s.step = 0;
	s.retry = 0;
	s.nsteps = -1;
}
QK_HD void heunAttempt(Tables const &t, CellCool const &c, double dt_total, double reltol, double abstol, HeunState &s)
{
	if (s.retry == 0 && (s.time + s.dt) > dt_total) { // (the head of the step loop)
		s.dt = dt_total - s.time;
	}
	const int k = s.retry;
	double eta = NAN;
	double k1 = NAN, k2 = NAN;
	bool ok = heatingRate(t, c, s.E, k1);
	if (ok) {
		k1 *= s.dt;
		ok = heatingRate(t, c, s.E + k1, k2);
	}
	if (!ok) {
		eta = 0.5;
	} else {
		k2 *= s.dt;
		const double Enew = s.E + 0.5 * k1 + 0.5 * k2;
		const double Eerr = -0.5 * k1 + 0.5 * k2;
		const double w = 1. / (reltol * s.E + abstol);
		const double epsilon = sqrt(((Eerr * Eerr) * (w * w)) / 1);
		eta = pow(epsilon, -1.0 / 2.0);
		if (epsilon < 1.0) { // accepted
			s.E = Enew;
			s.time += s.dt;
			eta = fmin(eta, (k == 0) ? 20. : 1.0);
			s.dt *= eta;
			s.retry = 0;
			s.step += 1;
			if (s.time >= dt_total) {
				s.nsteps = s.step;
			} else if (s.step >= maxSubsteps) {
				s.nsteps = maxSubsteps;
			}
			return;
		}
	}
	if (k == 1) {
		eta = fmin(eta, 0.3);
	} else if (k > 1) {
		eta = clampd(eta, 0.1, 0.3);
	}
	s.dt *= eta;
	s.retry = k + 1;
	if (s.retry >= 7) {
		s.nsteps = maxSubsteps; // no attempt of this step was accepted
	}
}
QK_HD auto integrateCooling(Tables const &t0, double rho, double gamma, double &E, double dt_total, double reltol, double abstol) -> int
{
	const Tables t = preparedCopy(t0);
	const CellCool c = cellCool(t, rho, gamma);
	HeunState s;
	heunBegin(t, c, E, dt_total, s);
	while (s.nsteps < 0) {
		heunAttempt(t, c, dt_total, reltol, abstol, s);
	}
	E = s.E;
	return s.nsteps;
}

} // namespace cool
} // namespace qk

#endif // QK_COOLING_DEVICE_HPP_
