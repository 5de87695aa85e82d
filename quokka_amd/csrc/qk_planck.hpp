// qk_planck.hpp — the Planck-spectrum arithmetic shared by the HIP library's multigroup kernels (qk_rad_mg_device.hpp) and the host mirror's
// RadSystem<problem_t> members that problem files call for initial and boundary states (host/quokka_rad_system.hpp, host/compat/planck_integral.hpp):
// ONE body each, callable on the host and on the device.  Counterparts in the reference, whose operation order these keep (bit parity):
//   src/radiation/planck_integral.hpp:225-262          Y(x) = (15 / pi^4) int_0^x t^3 / (e^t - 1) dt, tabulated in log10 x
//   src/radiation/radiation_system.hpp:430-461         group fractions of a Planck spectrum
//   src/radiation/radiation_system.hpp:483-513         thermal radiation of the groups and its temperature derivative
//   src/radiation/radiation_system.hpp:1311-1326       4 pi B(nu) / c
//   src/radiation/radiation_system.hpp:1354-1385       bin-centre opacity, group fluxes in the diffusion limit
//   src/radiation/radiation_system.hpp:1289-1308       gas energy <-> internal energy
// The table itself (data/planck_integral_table.inc) is computed from the definition by tools/make_planck_table.py.
#ifndef QK_PLANCK_HPP_
#define QK_PLANCK_HPP_

#include <cmath>

#if defined(__HIPCC__)
#define QK_PL_HD __host__ __device__ inline
#else
#define QK_PL_HD inline
#endif

namespace qk::planck
{

constexpr int TABLE_POINTS = 1000;
constexpr double LOG10_X_FIRST = -3., LOG10_X_LAST = 2.;

// the table as a function-local constant: constant memory on the device, .rodata on the host
struct LocalTable {
	QK_PL_HD auto operator[](int j) const -> double
	{
		constexpr double Y[TABLE_POINTS] = {
#include "../data/planck_integral_table.inc"
		};
		return Y[j];
	}
};

// Y(x).  Below the table the second-order series (-4 + x) x + 8 log((2 + x) / 2), held between 0 and the table's first entry so that Y stays
// monotonic; inside, linear interpolation in log10 x between the two bracketing points; 1 beyond the table's end.
template <class Table> QK_PL_HD auto fractionBelow(double x, Table const &Y) -> double
{
	if (!(x > 0.)) {
		return 0.;
	}
	const double lx = log10(x);
	if (lx >= LOG10_X_LAST) {
		return 1.0;
	}
	if (lx < LOG10_X_FIRST) {
		const double series = (-4 + x) * x + 8 * log((2 + x) / 2);
		const double cap = Y[0];
		return (series > cap) ? cap : ((series < 0.) ? 0. : series);
	}
	constexpr double span = LOG10_X_LAST - LOG10_X_FIRST;
	constexpr double step = span / (TABLE_POINTS - 1);
	const int cell = static_cast<int>((lx - LOG10_X_FIRST) / span * (TABLE_POINTS - 1));
	if (cell < 0) {
		return 0.0;
	}
	if (cell >= TABLE_POINTS - 1) {
		return 1.0;
	}
	const double left = Y[cell];
	const double rise = (Y[cell + 1] - left) / step;
	return rise * (lx - (LOG10_X_FIRST + cell * step)) + left;
}

// f[g] = Y(x_{g+1}) - Y(x_g) with x = edge * unit_over_kT, Y = 0 at the first edge and 1 at the last whatever the edges are
template <int NG, class Table> QK_PL_HD void groupFractions(const double *edges, double unit_over_kT, Table const &Y, double *f)
{
	double below = 0.0;
	for (int g = 0; g + 1 < NG; ++g) {
		const double x = edges[g + 1] * unit_over_kT;
		const double upto = (x >= 100.) ? 1.0 : fractionBelow(x, Y);
		f[g] = upto - below;
		below = upto;
	}
	f[NG - 1] = 1.0 - below;
}

// v[g] <- max(scale * v[g], floor): a T^4 times the group fractions with the radiation-energy floor (floor < 0: none)
template <int NG> QK_PL_HD void scaleFloored(double scale, double floor, double *v)
{
	for (int g = 0; g < NG; ++g) {
		const double e = scale * v[g];
		v[g] = (e < floor) ? floor : e;
	}
}
template <int NG> QK_PL_HD void scale(double s, double *v)
{
	for (int g = 0; g < NG; ++g) {
		v[g] = s * v[g];
	}
}

// 4 pi B(nu) / c with nu in units of the group edges: coeff = energy_unit / (k_B T), norm = pi^4 / 15, aT4 = a_rad T^4; `cube` is how the caller
// forms x^3 (std::pow on the host as the problem files do, the library's faithfully rounded product on the device)
template <class Cube> QK_PL_HD auto spectralDensity(double coeff, double nu, double norm, double aT4, Cube const &cube) -> double
{
	const double x = coeff * nu;
	if (x > 100.) {
		return 0.0;
	}
	const double shape = (x <= 1.0e-10) ? x * x - x * x * x / 2. : cube(x) / (exp(x) - 1.0);
	return coeff / norm * aT4 * shape;
}

// flux of each group for gas moving at `vel` through a Planck field in the diffusion limit: (vel a) T^4 [4/3 Y - 1/3 x (x^3 / (e^x - 1)) / norm] between
// the group's edges (vel_arad = vel * a_rad and T4 enter as the reference multiplies them: left to right)
template <int NG, class Table, class Cube>
QK_PL_HD void diffusionLimitFluxes(const double *edges, double coeff, double norm, double vel_arad, double T4, Table const &Y, Cube const &cube, double *flux)
{
	double lower = 0.;
	for (int g = 0; g <= NG; ++g) {
		const double x = coeff * edges[g];
		const double at_edge = 4. / 3. * fractionBelow(x, Y) - 1. / 3. * x * (cube(x) / (exp(x) - 1.0)) / norm;
		if (g > 0) {
			flux[g - 1] = vel_arad * T4 * (at_edge - lower);
		}
		lower = at_edge;
	}
}

// opacity at the logarithmic centre of each bin of a piecewise power law: lower value times (edge ratio)^(exponent / 2)
template <int NG> QK_PL_HD void binCentreOpacity(const double *edges, const double *exponent, const double *lower_value, double *kappa)
{
	for (int g = 0; g < NG; ++g) {
		kappa[g] = lower_value[g] * pow(edges[g + 1] / edges[g], 0.5 * exponent[g]);
	}
}

// Levermore's closure (radiation_system.hpp:773-790): the Eddington factor chi(f) = (3 + 4 f^2) / (5 + 2 sqrt(4 - 3 f^2)) of a reduced flux held in [0, 1]
QK_PL_HD auto levermoreFactor(double reduced_flux) -> double
{
	const double f = (reduced_flux < 0.) ? 0. : ((reduced_flux > 1.) ? 1. : reduced_flux);
	const double ff = f * f;
	return (3.0 + 4.0 * ff) / (5.0 + 2.0 * sqrt(4.0 - 3.0 * ff));
}

// kinetic energy of (rho, p): |p|^2 / (2 rho)
QK_PL_HD auto kineticEnergy(double rho, double px, double py, double pz) -> double { return (px * px + py * py + pz * pz) / (2.0 * rho); }

} // namespace qk::planck

#endif // QK_PLANCK_HPP_
