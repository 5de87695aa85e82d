// planck_integral.hpp of the host mirror: what the reference's `radiation/planck_integral.hpp` gives a problem file — the normalised incomplete
// Planck integral Y(x) = (15 / pi^4) int_0^x t^3 / (e^t - 1) dt by linear interpolation in log10 x on 1000 points.  The table is
// quokka_amd/data/planck_integral_table.inc (computed from the definition by tools/make_planck_table.py), the same numbers the HIP library
// and the CPU oracle use.
#ifndef QK_HOST_PLANCK_INTEGRAL_HPP_
#define QK_HOST_PLANCK_INTEGRAL_HPP_

#include <cmath>

#include "../amrex_mini.hpp"

using Real = amrex::Real;

static constexpr bool USE_SECOND_ORDER = false;
static constexpr double PI = M_PI;
static constexpr Real gInf = PI * PI * PI * PI / 15.0;
static constexpr int INTERP_SIZE = 1000;
static constexpr Real LOG_X_MIN = -3.;
static constexpr Real LOG_X_MAX = 2.;

AMREX_GPU_HOST_DEVICE AMREX_FORCE_INLINE auto qk_planck_table_entry(int j) -> Real
{
	constexpr Real Y[INTERP_SIZE] = {
#include "../../data/planck_integral_table.inc"
	};
	return Y[j];
}
static constexpr Real Y_INTERP_MIN = 5.1310665123189676e-11; // = Y_interp[0]

AMREX_GPU_HOST_DEVICE AMREX_FORCE_INLINE auto interpolate_planck_integral(Real logx) -> Real
{
	const int j = static_cast<int>((logx - LOG_X_MIN) / (LOG_X_MAX - LOG_X_MIN) * (INTERP_SIZE - 1));
	const Real gap = (LOG_X_MAX - LOG_X_MIN) / (INTERP_SIZE - 1);
	if (j < 0) {
		return 0.0;
	}
	if (j >= INTERP_SIZE - 1) {
		return 1.0;
	}
	const Real y0 = qk_planck_table_entry(j);
	const Real slope = (qk_planck_table_entry(j + 1) - y0) / gap;
	return slope * (logx - (LOG_X_MIN + j * gap)) + y0;
}

AMREX_GPU_HOST_DEVICE AMREX_FORCE_INLINE auto integrate_planck_from_0_to_x(const Real x) -> Real
{
	if (x <= 0.) {
		return 0.;
	}
	const Real logx = std::log10(x);
	Real y = NAN;
	if (logx < LOG_X_MIN) {
		y = (-4 + x) * x + 8 * std::log((2 + x) / 2);
		if (y > Y_INTERP_MIN) {
			y = Y_INTERP_MIN;
		} else if (y < 0.) {
			y = 0.;
		}
	} else if (logx >= LOG_X_MAX) {
		return 1.0;
	} else {
		y = interpolate_planck_integral(logx);
	}
	return y;
}

#endif // QK_HOST_PLANCK_INTEGRAL_HPP_
