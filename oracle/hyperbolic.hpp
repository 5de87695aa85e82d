// ORACLE — TEST INFRASTRUCTURE ONLY (see grid.hpp header).
//
// hyperbolic.hpp: restatement of
//   reference src/util/ArrayView_3d.hpp:18-113   (Array4View<DIR>, reorderMultiIndex<DIR>)
//   reference src/math/math_impl.hpp:15,18       (clamp, sgn)
//   reference src/hyperbolic_system.hpp:47-66    (MC, minmod)
//   reference src/hyperbolic_system.hpp:129-181  (ReconstructStatesConstant)
//   reference src/hyperbolic_system.hpp:183-247  (ReconstructStatesPLM)
//   reference src/hyperbolic_system.hpp:295-433  (ReconstructStatesPPM)
// Same operation order as the reference; build with -ffp-contract=off (reference
// CMakeLists.txt:31 DISABLE_FMAD=ON).
#ifndef ORACLE_HYPERBOLIC_HPP_
#define ORACLE_HYPERBOLIC_HPP_

#include <cmath>
#include <utility>

#include "grid.hpp"

namespace oracle
{

enum FluxDir : int { X1 = 0, X2 = 1, X3 = 2 };
enum SlopeLimiter : int { lim_minmod = 0, lim_MC = 1 };

// math_impl.hpp:15
inline auto clamp(double v, double lo, double hi) -> double { return (v < lo) ? lo : (hi < v) ? hi : v; }
// math_impl.hpp:18
inline auto sgn(double val) -> int { return static_cast<int>(0.0 < val) - static_cast<int>(val < 0.0); }

// AMREX_SPACEDIM of the reference build being restated: a compile-time global there, a process-wide variable here (set by the simulation that
// is about to run; 1-D and 3-D builds share ArrayView_3d.hpp, a 2-D build uses ArrayView_2d.hpp, where X2 is an index SWAP)
inline int g_spacedim = 3;

// ArrayView_3d.hpp:22-28 / ArrayView_2d.hpp:13-17. Loop index (i_in,j_in,k_in) -> view index (i,j,k)
struct Idx3 {
	int i, j, k;
};
inline auto reorderMultiIndex(int dir, int i, int j, int k) -> Idx3
{
	if (dir == X1) {
		return {i, j, k};
	}
	if (dir == X2) {
		if (g_spacedim == 2) {
			return {j, i, k};
		}
		return {j, k, i};
	}
	return {k, i, j};
}

// ArrayView_3d.hpp:30-113. view(i,j,k,n): X1 arr(i,j,k,n); X2 arr(k,i,j,n); X3 arr(j,k,i,n)
// ArrayView_2d.hpp:27-77.  view(i,j,k,n): X1 arr(i,j,k,n); X2 arr(j,i,k,n)
template <typename T> struct View {
	Array4<T> a;
	int dir;
	bool swap2d;
	View(Array4<T> arr, int d) : a(arr), dir(d), swap2d(g_spacedim == 2) {}
	auto operator()(int i, int j, int k, int n = 0) const -> T &
	{
		if (dir == X1) {
			return a(i, j, k, n);
		}
		if (dir == X2) {
			if (swap2d) {
				return a(j, i, k, n);
			}
			return a(k, i, j, n);
		}
		return a(j, k, i, n);
	}
};

// hyperbolic_system.hpp:58-61
inline auto MC(double a, double b) -> double
{
	return 0.5 * (sgn(a) + sgn(b)) * std::min(0.5 * std::abs(a + b), std::min(2.0 * std::abs(a), 2.0 * std::abs(b)));
}
// hyperbolic_system.hpp:63-66
inline auto minmod(double a, double b) -> double { return 0.5 * (sgn(a) + sgn(b)) * std::min(std::abs(a), std::abs(b)); }

inline auto SlopeFunc(int limiter, double x, double y) -> double { return (limiter == lim_minmod) ? minmod(x, y) : MC(x, y); }

// hyperbolic_system.hpp:164-181 (per-cell body), launched over `cellRange` x nvars (:139 / :159)
inline void ReconstructStatesConstant(int dir, Array4<const double> q_in, Array4<double> left_in, Array4<double> right_in, Box const &cellRange,
				      int nvars)
{
	View<const double> q(q_in, dir);
	View<double> leftState(left_in, dir);
	View<double> rightState(right_in, dir);
	for (int n = 0; n < nvars; ++n) {
		for (int k_in = cellRange.lo[2]; k_in <= cellRange.hi[2]; ++k_in) {
			for (int j_in = cellRange.lo[1]; j_in <= cellRange.hi[1]; ++j_in) {
				for (int i_in = cellRange.lo[0]; i_in <= cellRange.hi[0]; ++i_in) {
					auto [i, j, k] = reorderMultiIndex(dir, i_in, j_in, k_in);
					leftState(i, j, k, n) = q(i - 1, j, k, n);
					rightState(i, j, k, n) = q(i, j, k, n);
				}
			}
		}
	}
}

// hyperbolic_system.hpp:218-247
inline void ReconstructStatesPLM(int dir, int limiter, Array4<const double> q_in, Array4<double> left_in, Array4<double> right_in,
				 Box const &cellRange, int nvars)
{
	View<const double> q(q_in, dir);
	View<double> leftState(left_in, dir);
	View<double> rightState(right_in, dir);
	for (int n = 0; n < nvars; ++n) {
		for (int k_in = cellRange.lo[2]; k_in <= cellRange.hi[2]; ++k_in) {
			for (int j_in = cellRange.lo[1]; j_in <= cellRange.hi[1]; ++j_in) {
				for (int i_in = cellRange.lo[0]; i_in <= cellRange.hi[0]; ++i_in) {
					auto [i, j, k] = reorderMultiIndex(dir, i_in, j_in, k_in);
					const auto lslope = SlopeFunc(limiter, q(i, j, k, n) - q(i - 1, j, k, n), q(i - 1, j, k, n) - q(i - 2, j, k, n));
					const auto rslope = SlopeFunc(limiter, q(i + 1, j, k, n) - q(i, j, k, n), q(i, j, k, n) - q(i - 1, j, k, n));
					leftState(i, j, k, n) = q(i - 1, j, k, n) + 0.25 * lslope;
					rightState(i, j, k, n) = q(i, j, k, n) - 0.25 * rslope;
				}
			}
		}
	}
}

// hyperbolic_system.hpp:337-433 (MULTIDIM_EXTREMA_CHECK is hard-disabled, :30)
inline void ReconstructStatesPPM(int dir, Array4<const double> q_in, Array4<double> left_in, Array4<double> right_in, Box const &cellRange, int nvars,
				 int iReadFrom = 0, int iWriteFrom = 0)
{
	View<const double> q(q_in, dir);
	View<double> leftState(left_in, dir);
	View<double> rightState(right_in, dir);
	for (int n = 0; n < nvars; ++n) {
		for (int k_in = cellRange.lo[2]; k_in <= cellRange.hi[2]; ++k_in) {
			for (int j_in = cellRange.lo[1]; j_in <= cellRange.hi[1]; ++j_in) {
				for (int i_in = cellRange.lo[0]; i_in <= cellRange.hi[0]; ++i_in) {
					auto [i, j, k] = reorderMultiIndex(dir, i_in, j_in, k_in);

					// :365 bounds from neighbouring cell averages along the axis
					const std::pair<double, double> bounds =
					    std::minmax({q(i, j, k, iReadFrom + n), q(i - 1, j, k, iReadFrom + n), q(i + 1, j, k, iReadFrom + n)});

					// :380-385 interface estimate, grouped symmetrically
					const double coef_1 = (7. / 12.);
					const double coef_2 = (-1. / 12.);
					const double a_minus = (coef_1 * q(i, j, k, iReadFrom + n) + coef_2 * q(i + 1, j, k, iReadFrom + n)) +
							       (coef_1 * q(i - 1, j, k, iReadFrom + n) + coef_2 * q(i - 2, j, k, iReadFrom + n));
					const double a_plus = (coef_1 * q(i + 1, j, k, iReadFrom + n) + coef_2 * q(i + 2, j, k, iReadFrom + n)) +
							      (coef_1 * q(i, j, k, iReadFrom + n) + coef_2 * q(i - 1, j, k, iReadFrom + n));

					// :388-391
					double new_a_minus = clamp(a_minus, bounds.first, bounds.second);
					double new_a_plus = clamp(a_plus, bounds.first, bounds.second);

					// :396-400
					const double a = q(i, j, k, iReadFrom + n);
					const double dq_minus = (a - new_a_minus);
					const double dq_plus = (new_a_plus - a);
					const double qa = dq_plus * dq_minus;

					if (qa <= 0.0) { // :402 local extremum
						const double dq0 =
						    MC(q(i + 1, j, k, iReadFrom + n) - q(i, j, k, iReadFrom + n), q(i, j, k, iReadFrom + n) - q(i - 1, j, k, iReadFrom + n));
						new_a_minus = a - 0.5 * dq0;
						new_a_plus = a + 0.5 * dq0;
					} else { // :418
						if (std::abs(dq_minus) >= 2.0 * std::abs(dq_plus)) {
							new_a_minus = a - 2.0 * dq_plus;
						}
						if (std::abs(dq_plus) >= 2.0 * std::abs(dq_minus)) {
							new_a_plus = a + 2.0 * dq_minus;
						}
					}

					rightState(i, j, k, iWriteFrom + n) = new_a_minus;
					leftState(i + 1, j, k, iWriteFrom + n) = new_a_plus;
				}
			}
		}
	}
}

} // namespace oracle

#endif // ORACLE_HYPERBOLIC_HPP_
