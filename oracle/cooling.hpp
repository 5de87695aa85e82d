// ORACLE — TEST INFRASTRUCTURE ONLY (see grid.hpp header).
//
// cooling.hpp: CPU restatement of the reference's tabulated (Cloudy) cooling, function by function:
//   src/cooling/CloudyDataReader.cpp:24-222   table preparation (what follows H5Dread: log10 T, FastMath::log10 of the rates, transposition)
//   src/math/FastMath.hpp:33-72               fastlg / fastpow2 / log10 / pow10
//   src/math/Interpolate2D.hpp:14-80          interpolate2d
//   src/math/root_finding.hpp:62-340          eps_tolerance, bracket, safe_div, secant / quadratic / cubic interpolation, toms748_solve
//   src/math/ODEIntegrate.hpp:20-53,88-226    rk12_single_step, error_norm, rk_adaptive_integrate
//   src/cooling/TabulatedCooling.hpp:82-317   cloudy_cooling_function ... computeCooling (the per-cell body)
// PARITY UNPINNED against the reference binary: the reference's only known-answer test of this module (src/problems/Cooling) needs
// extern/grackle_data_files, an empty submodule here, and the Cloudy-table problem (ShockCloud) has a regression test only.  What pins this file is
// its line-by-line correspondence with the cited sources and the properties tests/test_cooling_oracle.py checks (thermal equilibrium is a fixed
// point, E(T(E)) round trips to the bracket tolerance, the table axes and ranges of the reference's own data file).
#ifndef ORACLE_COOLING_HPP_
#define ORACLE_COOLING_HPP_

#include <algorithm>
#include <array>
#include <cmath>
#include <limits>
#include <utility>
#include <vector>

#include "eos.hpp"

namespace oracle::cooling
{

// ---------------------------------------------------------------- FastMath.hpp
inline auto fastlg(const double x) -> double // :33-40
{
	int n = 0;
	const double y = frexp(x, &n);
	return 2 * (y - 1) + n;
}
inline auto fastpow2(const double x) -> double // :42-49
{
	const int flr = std::floor(x);
	const double remainder = x - flr;
	const double mantissa = 0.5 * (remainder + 1);
	const int exponent = flr + 1;
	return ldexp(mantissa, exponent);
}
inline auto fm_log10(const double x) -> double // :62-66
{
	constexpr double LOG2OLOG10 = 0.301029995663981195;
	return LOG2OLOG10 * fastlg(x);
}
inline auto fm_pow10(const double x) -> double // :68-72
{
	constexpr double LOG10OLOG2 = 3.321928094887362626;
	return fastpow2(LOG10OLOG2 * x);
}

// ---------------------------------------------------------------- amrex::Table1D / Table2D (Fortran order, begin inclusive, end exclusive)
struct Table1D {
	const double *p = nullptr;
	int begin = 0, end = 0;
	auto operator()(int i) const -> double { return p[i - begin]; }
};
struct Table2D {
	const double *p = nullptr;
	int begin[2] = {0, 0}, end[2] = {0, 0};
	auto operator()(int i, int j) const -> double { return p[(i - begin[0]) + static_cast<long>(j - begin[1]) * (end[0] - begin[0])]; }
};

// ---------------------------------------------------------------- Interpolate2D.hpp:14-80
inline auto interpolate2d(double x, double y, Table1D const &xv, Table1D const &yv, Table2D const &table) -> double
{
	double xi = xv(xv.begin);
	double xf = xv(xv.end - 1);
	double yi = yv(yv.begin);
	double yf = yv(yv.end - 1);

	double dx = (xf - xi) / static_cast<double>(xv.end - xv.begin - 1);
	double dy = (yf - yi) / static_cast<double>(yv.end - yv.begin - 1);

	x = std::clamp(x, xi, xf);
	y = std::clamp(y, yi, yf);

	int ix = std::clamp(static_cast<int>(std::floor((x - xi) / dx)), xv.begin, xv.end - 1);
	int iy = std::clamp(static_cast<int>(std::floor((y - yi) / dy)), yv.begin, yv.end - 1);
	int iix = (ix == xv.end - 1) ? ix : ix + 1;
	int iiy = (iy == yv.end - 1) ? iy : iy + 1;

	double x1 = xv(ix);
	double x2 = xv(iix);
	double y1 = yv(iy);
	double y2 = yv(iiy);

	double w11 = 0;
	double w12 = 0;
	double w21 = 0;
	double w22 = 0;

	if (ix != iix && iy != iiy) {
		const double vol = ((x2 - x1) * (y2 - y1));
		w11 = (x2 - x) * (y2 - y) / vol;
		w12 = (x2 - x) * (y - y1) / vol;
		w21 = (x - x1) * (y2 - y) / vol;
		w22 = (x - x1) * (y - y1) / vol;
	} else if (ix == iix && yi != iiy) { // (:54: the ordinate yi against the index iiy, as the reference has it)
		const double vol = (y2 - y1);
		w11 = (y2 - y) / vol;
		w12 = (y - y1) / vol;
	} else if (ix != iix && yi == iiy) { // (:59: likewise)
		const double vol = (x2 - x1);
		w11 = (x2 - x) / vol;
		w21 = (x - x1) / vol;
	} else {
		w11 = 1.0;
	}

	double A = table(ix, iy);
	double B = table(ix, iiy);
	double C = table(iix, iy);
	double D = table(iix, iiy);

	return w11 * A + w12 * B + w21 * C + w22 * D;
}

// ---------------------------------------------------------------- math_impl.hpp
inline auto clamp_mi(double v, double lo, double hi) -> double { return (v < lo) ? lo : (hi < v) ? hi : v; }
template <typename T> inline auto sgn(T val) -> int { return (T(0) < val) - (val < T(0)); }

// ---------------------------------------------------------------- root_finding.hpp
struct eps_tolerance { // :62-82
	double eps;
	auto operator()(const double &a, const double &b) const -> bool { return fabs(a - b) <= (eps * (std::min)(fabs(a), fabs(b))); }
};

template <class F> inline void bracket(F f, double &a, double &b, double c, double &fa, double &fb, double &d, double &fd) // :87-118
{
	double tol = std::numeric_limits<double>::epsilon() * 2;
	if ((b - a) < 2 * tol * a) {
		c = a + (b - a) / 2;
	} else if (c <= a + fabs(a) * tol) {
		c = a + fabs(a) * tol;
	} else if (c >= b - fabs(b) * tol) {
		c = b - fabs(b) * tol;
	}
	double fc = f(c);
	if (fc == 0) {
		a = c;
		fa = 0;
		d = 0;
		fd = 0;
		return;
	}
	if (sgn(fa) * sgn(fc) < 0) {
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

inline auto safe_div(double num, double denom, double r) -> double // :120-132
{
	if (fabs(denom) < 1) {
		if (fabs(denom * std::numeric_limits<double>::max()) <= fabs(num)) {
			return r;
		}
	}
	return num / denom;
}

inline auto secant_interpolate(const double &a, const double &b, const double &fa, const double &fb) -> double // :134-145
{
	double tol = std::numeric_limits<double>::epsilon() * 5;
	double c = a - (fa / (fb - fa)) * (b - a);
	if ((c <= a + fabs(a) * tol) || (c >= b - fabs(b) * tol)) {
		return (a + b) / 2;
	}
	return c;
}

inline auto quadratic_interpolate(const double &a, const double &b, double const &d, const double &fa, const double &fb, double const &fd, unsigned count)
    -> double // :147-176
{
	double B = safe_div(fb - fa, b - a, std::numeric_limits<double>::max());
	double A = safe_div(fd - fb, d - b, std::numeric_limits<double>::max());
	A = safe_div(A - B, d - a, 0.0);

	if (A == 0) {
		return secant_interpolate(a, b, fa, fb);
	}
	double c;
	if (sgn(A) * sgn(fa) > 0) {
		c = a;
	} else {
		c = b;
	}
	for (unsigned i = 1; i <= count; ++i) {
		c -= safe_div(fa + (B + A * (c - b)) * (c - a), B + A * (2 * c - a - b), 1 + c - a);
	}
	if ((c <= a) || (c >= b)) {
		c = secant_interpolate(a, b, fa, fb);
	}
	return c;
}

inline auto cubic_interpolate(const double &a, const double &b, const double &d, const double &e, const double &fa, const double &fb, const double &fd,
			      const double &fe) -> double // :178-208
{
	double q11 = (d - e) * fd / (fe - fd);
	double q21 = (b - d) * fb / (fd - fb);
	double q31 = (a - b) * fa / (fb - fa);
	double d21 = (b - d) * fd / (fd - fb);
	double d31 = (a - b) * fb / (fb - fa);
	double q22 = (d21 - q11) * fb / (fe - fb);
	double q32 = (d31 - q21) * fa / (fd - fa);
	double d32 = (d31 - q21) * fd / (fd - fa);
	double q33 = (d32 - q22) * fa / (fe - fa);
	double c = q31 + q32 + q33 + a;

	if ((c <= a) || (c >= b)) {
		c = quadratic_interpolate(a, b, d, fa, fb, fd, 3);
	}
	return c;
}

template <class F>
inline auto toms748_solve(F f, const double &ax, const double &bx, const double &fax, const double &fbx, eps_tolerance tol, int &max_iter)
    -> std::pair<double, double> // :212-330
{
	int count = max_iter;
	double a, b, fa, fb, c, u, fu, a0, b0, d, fd, e, fe;
	static const double mu = 0.5F;

	a = ax;
	b = bx;
	fa = fax;
	fb = fbx;

	if (tol(a, b) || (fa == 0) || (fb == 0)) {
		max_iter = 0;
		if (fa == 0) {
			b = a;
		} else if (fb == 0) {
			a = b;
		}
		return std::make_pair(a, b);
	}

	fe = e = fd = 1e5F;
	d = 0;

	if (fa != 0) {
		c = secant_interpolate(a, b, fa, fb);
		bracket(f, a, b, c, fa, fb, d, fd);
		--count;

		if (count && (fa != 0) && !tol(a, b)) {
			c = quadratic_interpolate(a, b, d, fa, fb, fd, 2);
			e = d;
			fe = fd;
			bracket(f, a, b, c, fa, fb, d, fd);
			--count;
		}
	}

	while (count && (fa != 0) && !tol(a, b)) {
		a0 = a;
		b0 = b;
		double min_diff = std::numeric_limits<double>::min() * 32;
		bool prof = (fabs(fa - fb) < min_diff) || (fabs(fa - fd) < min_diff) || (fabs(fa - fe) < min_diff) || (fabs(fb - fd) < min_diff) ||
			    (fabs(fb - fe) < min_diff) || (fabs(fd - fe) < min_diff);
		if (prof) {
			c = quadratic_interpolate(a, b, d, fa, fb, fd, 2);
		} else {
			c = cubic_interpolate(a, b, d, e, fa, fb, fd, fe);
		}
		e = d;
		fe = fd;
		bracket(f, a, b, c, fa, fb, d, fd);
		if ((0 == --count) || (fa == 0) || tol(a, b)) {
			break;
		}
		prof = (fabs(fa - fb) < min_diff) || (fabs(fa - fd) < min_diff) || (fabs(fa - fe) < min_diff) || (fabs(fb - fd) < min_diff) ||
		       (fabs(fb - fe) < min_diff) || (fabs(fd - fe) < min_diff);
		if (prof) {
			c = quadratic_interpolate(a, b, d, fa, fb, fd, 3);
		} else {
			c = cubic_interpolate(a, b, d, e, fa, fb, fd, fe);
		}
		bracket(f, a, b, c, fa, fb, d, fd);
		if ((0 == --count) || (fa == 0) || tol(a, b)) {
			break;
		}
		if (fabs(fa) < fabs(fb)) {
			u = a;
			fu = fa;
		} else {
			u = b;
			fu = fb;
		}
		c = u - 2 * (fu / (fb - fa)) * (b - a);
		if (fabs(c - u) > (b - a) / 2) {
			c = a + (b - a) / 2;
		}
		e = d;
		fe = fd;
		bracket(f, a, b, c, fa, fb, d, fd);
		if ((0 == --count) || (fa == 0) || tol(a, b)) {
			break;
		}
		if ((b - a) < mu * (b0 - a0)) {
			continue;
		}
		e = d;
		fe = fd;
		bracket(f, a, b, a + (b - a) / 2, fa, fb, d, fd);
		--count;
	}

	max_iter -= count;
	if (fa == 0) {
		b = a;
	} else if (fb == 0) {
		a = b;
	}
	return std::make_pair(a, b);
}

template <class F> inline auto toms748_solve(F f, const double &ax, const double &bx, eps_tolerance tol, int &max_iter) -> std::pair<double, double> // :332-340
{
	max_iter -= 2;
	std::pair<double, double> r = toms748_solve(f, ax, bx, f(ax), f(bx), tol, max_iter);
	max_iter += 2;
	return r;
}

// ---------------------------------------------------------------- CloudyDataReader.cpp + TabulatedCooling.cpp
constexpr double cloudy_H_mass_fraction = 1. / (1. + 0.098 * 3.971); // TabulatedCooling.hpp:32

struct cloudy_tables {
	std::vector<double> log_nH_v, log_Tgas_v, cool_v, heat_v, mmw_v; // transposed: (nH, T), nH fastest
	Table1D log_nH, log_Tgas;
	Table2D cool, heat, meanMolWeight;
	double T_min = std::numeric_limits<double>::max();
	double T_max = std::numeric_limits<double>::min();
	double mmw_min = std::numeric_limits<double>::max();
	double mmw_max = std::numeric_limits<double>::min();
};

// the arrays as H5Dread delivers them: Parameter1[n0], Temperature[n1], Cooling / Heating / MMW [n0][n1] (C order)
inline auto prepare_tables(int n0, int n1, const double *parameter1, const double *temperature, const double *cooling, const double *heating, const double *mmw)
    -> cloudy_tables
{
	cloudy_tables t;
	// CloudyDataReader.cpp:36-50 with the cgs units of TabulatedCooling.cpp:12-17
	const double mh = 1.67e-24;
	const double CoolUnit = (1.0 * 1.0 * mh * mh) / (1.0 * 1.0 * 1.0 * 1.0);
	const double small_fastlog_value = fm_log10(1.0e-99 / CoolUnit);
	// :91-117
	t.log_nH_v.assign(parameter1, parameter1 + n0);
	t.log_Tgas_v.assign(temperature, temperature + n1);
	for (int w = 0; w < n1; w++) {
		double const T = t.log_Tgas_v[w];
		t.log_Tgas_v[w] = log10(T);
		t.T_min = std::min(T, t.T_min);
		t.T_max = std::max(T, t.T_max);
	}
	// :124-194: the rates, then extract_2d_table (:206-222): table(i, j) = table2D(j, i), i over Parameter1
	auto rates = [&](const double *file, bool take_log) {
		std::vector<double> v(file, file + static_cast<size_t>(n0) * n1);
		if (take_log) {
			for (auto &x : v) {
				double const value = x / CoolUnit;
				x = value > 0 ? fm_log10(value) : small_fastlog_value;
			}
		}
		std::vector<double> out(v.size());
		for (int i = 0; i < n0; ++i) {
			for (int j = 0; j < n1; ++j) {
				out[i + static_cast<size_t>(n0) * j] = v[j + static_cast<size_t>(n1) * i];
			}
		}
		return out;
	};
	t.cool_v = rates(cooling, true);
	t.heat_v = rates(heating, true);
	t.mmw_v = rates(mmw, false);
	for (int q = 0; q < n0 * n1; ++q) {
		t.mmw_min = std::min(mmw[q], t.mmw_min);
		t.mmw_max = std::max(mmw[q], t.mmw_max);
	}
	t.log_nH = Table1D{t.log_nH_v.data(), 0, n0};
	t.log_Tgas = Table1D{t.log_Tgas_v.data(), 0, n1};
	auto t2 = [&](std::vector<double> const &v) {
		Table2D r;
		r.p = v.data();
		r.end[0] = n0;
		r.end[1] = n1;
		return r;
	};
	t.cool = t2(t.cool_v);
	t.heat = t2(t.heat_v);
	t.meanMolWeight = t2(t.mmw_v);
	return t;
}

// ---------------------------------------------------------------- TabulatedCooling.hpp
inline auto cloudy_cooling_function(double const rho, double const T, cloudy_tables const &tables) -> double // :82-99
{
	const double rhoH = rho * cloudy_H_mass_fraction;
	const double nH = rhoH / (C::m_p + C::m_e);
	const double log_nH = std::log10(nH);
	const double log_T = std::log10(T);

	const double logCool = interpolate2d(log_nH, log_T, tables.log_nH, tables.log_Tgas, tables.cool);
	const double logHeat = interpolate2d(log_nH, log_T, tables.log_nH, tables.log_Tgas, tables.heat);
	const double netLambda = fm_pow10(logHeat) - fm_pow10(logCool);

	const double Edot = (rhoH * rhoH) * netLambda;
	return Edot;
}

inline auto ComputeEgasFromTgas(double rho, double Tgas, double gamma, cloudy_tables const &tables) -> double // :101-115
{
	const double rhoH = rho * cloudy_H_mass_fraction;
	const double nH = rhoH / (C::m_p + C::m_e);
	const double mu = interpolate2d(std::log10(nH), std::log10(Tgas), tables.log_nH, tables.log_Tgas, tables.meanMolWeight);
	const double n = rho / ((C::m_p + C::m_e) * mu);
	const double Pgas = n * C::k_B * Tgas;
	const double Egas = Pgas / (gamma - 1.);
	return Egas;
}

inline auto ComputeTgasFromEgas(double rho, double Egas, double gamma, cloudy_tables const &tables) -> double // :117-174
{
	const double Eint_min = ComputeEgasFromTgas(rho, tables.T_min, gamma, tables);
	const double Eint_max = ComputeEgasFromTgas(rho, tables.T_max, gamma, tables);

	if (Egas <= Eint_min) {
		return tables.T_min;
	}
	if (Egas >= Eint_max) {
		return tables.T_max;
	}

	const double rhoH = rho * cloudy_H_mass_fraction;
	const double nH = rhoH / (C::m_p + C::m_e);
	const double log_nH = std::log10(nH);

	const double C_ = (gamma - 1.) * Egas / (C::k_B * (rho / (C::m_p + C::m_e)));

	const double reltol = 1.0e-5;
	const int maxIterLimit = 100;
	int maxIter = maxIterLimit;

	auto f = [log_nH, C_, &tables](const double &T) noexcept {
		double const log_T = clamp_mi(std::log10(T), 1., 9.);
		double const mu = interpolate2d(log_nH, log_T, tables.log_nH, tables.log_Tgas, tables.meanMolWeight);
		double const fun = C_ * mu - T;
		return fun;
	};

	const double T_min = std::clamp(C_ * tables.mmw_min, tables.T_min, tables.T_max);
	const double T_max = std::clamp(C_ * tables.mmw_max, tables.T_min, tables.T_max);

	eps_tolerance const tol{reltol};
	double T_sol = NAN;

	if (T_min < T_max) {
		auto bounds = toms748_solve(f, T_min, T_max, tol, maxIter);
		T_sol = 0.5 * (bounds.first + bounds.second);

		if ((maxIter >= maxIterLimit) || std::isnan(T_sol)) {
			T_sol = NAN;
		}
	}
	return T_sol;
}

inline auto ComputeCoolingLength(double rho, double Egas, double gamma, cloudy_tables const &tables) -> double // :176-204
{
	const double Tgas = ComputeTgasFromEgas(rho, Egas, gamma, tables);
	const double rhoH = rho * cloudy_H_mass_fraction;
	const double nH = rhoH / (C::m_p + C::m_e);
	const double log_nH = std::log10(nH);
	const double log_T = std::log10(Tgas);
	const double logCool = interpolate2d(log_nH, log_T, tables.log_nH, tables.log_Tgas, tables.cool);
	const double LambdaCool = fm_pow10(logCool);
	const double Edot = (rhoH * rhoH) * LambdaCool;
	const double t_cool = Egas / Edot;
	const double mu = interpolate2d(log_nH, log_T, tables.log_nH, tables.log_Tgas, tables.meanMolWeight);
	const double c_s = std::sqrt(gamma * C::k_B * Tgas / (mu * (C::m_p + C::m_e)));
	return c_s * t_cool;
}

inline auto ComputeMMW(double rho, double Egas, double gamma, cloudy_tables const &tables) -> double // :206-220
{
	const double Tgas = ComputeTgasFromEgas(rho, Egas, gamma, tables);
	const double rhoH = rho * cloudy_H_mass_fraction;
	const double nH = rhoH / (C::m_p + C::m_e);
	const double log_nH = std::log10(nH);
	const double log_T = std::log10(Tgas);
	return interpolate2d(log_nH, log_T, tables.log_nH, tables.log_Tgas, tables.meanMolWeight);
}

struct ODEUserData { // :75-79
	double rho{};
	double gamma{};
	cloudy_tables const *tables = nullptr;
};

inline auto user_rhs(double /*t*/, std::array<double, 1> &y_data, std::array<double, 1> &y_rhs, void *user_data) -> int // :222-256
{
	auto *udata = static_cast<ODEUserData *>(user_data);
	const double rho = udata->rho;
	const double gamma = udata->gamma;
	cloudy_tables const &tables = *udata->tables;

	const double Eint_min = ComputeEgasFromTgas(rho, tables.T_min, gamma, tables);
	const double Eint_max = ComputeEgasFromTgas(rho, tables.T_max, gamma, tables);
	const double Eint = y_data[0];

	if (Eint <= Eint_min) {
		y_rhs[0] = cloudy_cooling_function(rho, tables.T_min, tables);
	} else if (Eint >= Eint_max) {
		y_rhs[0] = cloudy_cooling_function(rho, tables.T_max, tables);
	} else {
		const double T = ComputeTgasFromEgas(rho, Eint, gamma, tables);
		if (!std::isnan(T)) {
			y_rhs[0] = cloudy_cooling_function(rho, T, tables);
		} else {
			y_rhs[0] = NAN;
			return 1;
		}
	}
	return 0;
}

// ---------------------------------------------------------------- ODEIntegrate.hpp (N = 1)
using v1 = std::array<double, 1>;

template <typename F> inline auto rk12_single_step(F &&rhs, double t0, v1 const &y, double dt, v1 &ynew, v1 &yerr, void *user_data) -> int // :20-53
{
	v1 k1{};
	v1 y_arg = y;
	int ierr = rhs(t0, y_arg, k1, user_data);
	if (ierr != 0) {
		return ierr;
	}
	k1[0] *= dt;

	v1 k2{};
	y_arg[0] = y[0] + k1[0];
	ierr = rhs(t0 + dt, y_arg, k2, user_data);
	if (ierr != 0) {
		return ierr;
	}
	k2[0] *= dt;

	ynew[0] = y[0] + 0.5 * k1[0] + 0.5 * k2[0];
	yerr[0] = -0.5 * k1[0] + 0.5 * k2[0];
	return 0;
}

inline auto error_norm(v1 const &y0, v1 const &yerr, double reltol, v1 const &abstol) -> double // :105-118
{
	double err_sq = 0;
	for (int i = 0; i < 1; ++i) {
		double w_i = 1. / (reltol * y0[i] + abstol[i]);
		err_sq += (yerr[i] * yerr[i]) * (w_i * w_i);
	}
	const double err = std::sqrt(err_sq / 1);
	return err;
}

constexpr int maxStepsODEIntegrate = 2000; // :120

template <typename F>
inline void rk_adaptive_integrate(F &&rhs, double t0, v1 &y0, double t1, void *user_data, double reltol, v1 const &abstol, int &steps_taken) // :122-224
{
	v1 ydot0{};
	rhs(t0, y0, ydot0, user_data);
	const double dt_guess = 0.1 * std::abs(y0[0] / ydot0[0]);

	const int maxRetries = 7;
	const int p = 2;
	const double eta_max = 20.;
	const double eta_max_errfail_prevstep = 1.0;
	const double eta_max_errfail_again = 0.3;
	const double eta_min_errfail_multiple = 0.1;
	const double eta_retry_failed_rhs = 0.5;

	double time = t0;
	double dt = std::isnan(dt_guess) ? (t1 - t0) : dt_guess;
	v1 &y = y0;
	v1 yerr{};
	v1 ynew{};

	bool success = false;
	for (int i = 0; i < maxStepsODEIntegrate; ++i) {
		if ((time + dt) > t1) {
			dt = t1 - time;
		}

		bool step_success = false;
		for (int k = 0; k < maxRetries; ++k) {
			int ierr = rk12_single_step(rhs, time, y, dt, ynew, yerr, user_data);

			double eta = NAN;
			double epsilon = NAN;

			if (ierr != 0) {
				eta = eta_retry_failed_rhs;
			} else {
				epsilon = error_norm(y, yerr, reltol, abstol);
				eta = std::pow(epsilon, -1.0 / static_cast<double>(p));

				if (epsilon < 1.0) {
					y = ynew;
					time += dt;
					if (k == 0) {
						eta = std::min(eta, eta_max);
					} else {
						eta = std::min(eta, eta_max_errfail_prevstep);
					}
					dt *= eta;
					step_success = true;
					break;
				}
			}

			if (k == 1) {
				eta = std::min(eta, eta_max_errfail_again);
			} else if (k > 1) {
				eta = std::clamp(eta, eta_min_errfail_multiple, eta_max_errfail_again);
			}
			dt *= eta;
		}

		if (!step_success) {
			success = false;
			break;
		}

		if (time >= t1) {
			success = true;
			steps_taken = i + 1;
			break;
		}
	}

	if (!success) {
		steps_taken = maxStepsODEIntegrate;
	}
}

// ---------------------------------------------------------------- computeCooling, the body of its ParallelFor (TabulatedCooling.hpp:277-307)
// U = (rho, x1Mom, x2Mom, x3Mom, Egas, Eint_aux) of one cell; returns nsteps
inline auto computeCoolingCell(double U[6], double dt, cloudy_tables const &tables, double T_floor, double gamma) -> int
{
	const double reltol_floor = 0.01;
	const double rtol = 1.0e-4;
	const double rho = U[0];
	const double x1Mom = U[1], x2Mom = U[2], x3Mom = U[3];
	const double Egas = U[4];
	// RadSystem::ComputeEintFromEgas (radiation_system.hpp:1288-1297)
	const double p_sq = x1Mom * x1Mom + x2Mom * x2Mom + x3Mom * x3Mom;
	const double Ekin = p_sq / (2.0 * rho);
	const double Eint = Egas - Ekin;

	ODEUserData user_data{rho, gamma, &tables};
	v1 y = {Eint};
	v1 const abstol = {reltol_floor * ComputeEgasFromTgas(rho, T_floor, gamma, tables)};

	int nsteps = 0;
	rk_adaptive_integrate(user_rhs, 0, y, dt, &user_data, rtol, abstol, nsteps);

	const double Eint_new = y[0];
	const double dEint = Eint_new - Eint;
	U[4] += dEint;
	U[5] += dEint;
	return nsteps;
}

} // namespace oracle::cooling

#endif // ORACLE_COOLING_HPP_
