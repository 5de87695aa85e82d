// quadrature.hpp — the reference's src/math/quadrature.hpp (with the part of gauss.hpp it uses): 7-point Gauss-Legendre quadrature of a device
// function over an interval / rectangle / box, and the Wendland C2 kernel the supernova-injection problems smooth their sources with.
// Nodes and weights: the roots of P_7 and 2 / ((1 - x^2) P_7'(x)^2), to 17 digits.
#ifndef QK_HOST_COMPAT_QUADRATURE_HPP_
#define QK_HOST_COMPAT_QUADRATURE_HPP_

#include <cmath>

#include "../amrex_mini.hpp"

template <typename F> AMREX_FORCE_INLINE AMREX_GPU_DEVICE auto quad_1d(F &&f, amrex::Real x0, amrex::Real x1) -> amrex::Real
{
	constexpr double node[4] = {0.0, 0.40584515137739717, 0.74153118559939444, 0.94910791234275852};
	constexpr double weight[4] = {0.41795918367346939, 0.38183005050511894, 0.27970539148927667, 0.12948496616886969};
	const double mid = (x0 + x1) / 2, half = (x1 - x0) / 2;
	double sum = weight[0] * f(mid);
	for (int n = 1; n < 4; ++n) {
		const double d = half * node[n];
		sum += weight[n] * (f(mid + d) + f(mid - d));
	}
	return sum * half;
}
template <typename F> AMREX_FORCE_INLINE AMREX_GPU_DEVICE auto quad_2d(F &&f, amrex::Real x0, amrex::Real x1, amrex::Real y0, amrex::Real y1) -> amrex::Real
{
	return quad_1d([=] AMREX_GPU_DEVICE(amrex::Real y) { return quad_1d([=] AMREX_GPU_DEVICE(amrex::Real x) { return f(x, y); }, x0, x1); }, y0, y1);
}
template <typename F>
AMREX_FORCE_INLINE AMREX_GPU_DEVICE auto quad_3d(F &&f, amrex::Real x0, amrex::Real x1, amrex::Real y0, amrex::Real y1, amrex::Real z0, amrex::Real z1) -> amrex::Real
{
	return quad_1d([=] AMREX_GPU_DEVICE(amrex::Real z) { return quad_2d([=] AMREX_GPU_DEVICE(amrex::Real x, amrex::Real y) { return f(x, y, z); }, x0, x1, y0, y1); },
		       z0, z1);
}
// W(r) = 21 / (2 pi) (1 - r)^4 (4 r + 1) inside the unit ball (Wendland 1995), normalised to unit volume integral
AMREX_FORCE_INLINE AMREX_GPU_DEVICE auto kernel_wendland_c2(const amrex::Real r) -> amrex::Real
{
	if (r > 1.0) {
		return 0;
	}
	return (21. / (2. * M_PI)) * std::pow((1.0 - r), 4) * (4.0 * r + 1.0);
}

#endif // QK_HOST_COMPAT_QUADRATURE_HPP_
