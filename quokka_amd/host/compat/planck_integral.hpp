// planck_integral.hpp of the host mirror: what the reference's `radiation/planck_integral.hpp` gives a problem file — the names only.  The normalised
// incomplete Planck integral Y(x) = (15 / pi^4) int_0^x t^3 / (e^t - 1) dt is evaluated by the HIP library's own body (csrc/qk_planck.hpp, shared
// with its multigroup kernels) on quokka_amd/data/planck_integral_table.inc (computed from the definition by tools/make_planck_table.py).
#ifndef QK_HOST_PLANCK_INTEGRAL_HPP_
#define QK_HOST_PLANCK_INTEGRAL_HPP_

#include "../../csrc/qk_planck.hpp"
#include "../amrex_mini.hpp"

using Real = amrex::Real;

// (names problem files and RadSystem use: src/radiation/planck_integral.hpp:19-27)
static constexpr double PI = M_PI;
static constexpr Real gInf = PI * PI * PI * PI / 15.0;
static constexpr int INTERP_SIZE = qk::planck::TABLE_POINTS;
static constexpr Real LOG_X_MIN = qk::planck::LOG10_X_FIRST, LOG_X_MAX = qk::planck::LOG10_X_LAST;

AMREX_GPU_HOST_DEVICE AMREX_FORCE_INLINE auto integrate_planck_from_0_to_x(const Real x) -> Real { return qk::planck::fractionBelow(x, qk::planck::LocalTable{}); }
// (the reference also exposes the table interpolation by itself, interpolate_planck_integral; nothing outside its own header calls it: not mirrored)

#endif // QK_HOST_PLANCK_INTEGRAL_HPP_
