// math/interpolate.hpp — interpolate_value of the reference's src/math/interpolate.{hpp,cpp} (itself the numpy.interp kernel):
// bracket x in the monotonically increasing table, then  y = slope * (x - x_j) + y_j  with slope = (y_{j+1} - y_j) / (x_{j+1} - x_j);
// NaN outside the table, y_j exactly on a node and on the last node.  Written for the host mirror (ICs are host-evaluated).
#ifndef QK_HOST_MATH_INTERPOLATE_HPP_
#define QK_HOST_MATH_INTERPOLATE_HPP_

#include <cmath>
#include <cstdint>

#include "../amrex_mini.hpp" // QK_HD: the tables are read inside device lambdas in device mode

// index j with arr[j] <= key < arr[j + 1];  -1 below the table, len above it
QK_HD inline auto qk_bracket(double key, double const *arr, int64_t len) -> int64_t
{
	if (key > arr[len - 1]) {
		return len;
	}
	if (key < arr[0]) {
		return -1;
	}
	int64_t lo = 0, hi = len; // invariant: arr[lo] <= key, (hi == len or key < arr[hi])
	while (hi - lo > 1) {
		int64_t const mid = lo + (hi - lo) / 2;
		if (key >= arr[mid]) {
			lo = mid;
		} else {
			hi = mid;
		}
	}
	return lo;
}

QK_HD inline auto interpolate_value(double x, double const *arr_x, double const *arr_y, int arr_len) -> double
{
	int64_t const j = qk_bracket(x, arr_x, arr_len);
	if (j == -1 || j == arr_len) {
		return NAN;
	}
	if (j == arr_len - 1 || x == arr_x[j]) {
		return arr_y[j];
	}
	double const slope = (arr_y[j + 1] - arr_y[j]) / (arr_x[j + 1] - arr_x[j]);
	return slope * (x - arr_x[j]) + arr_y[j];
}

// interpolate_arrays of the reference (src/math/interpolate.cpp:107-133): y[i] = table(x[i]) for a sorted table; NaN outside it
QK_HD inline void interpolate_arrays(double const *x, double *y, int len, double const *arr_x, double const *arr_y, int arr_len)
{
	for (int i = 0; i < len; ++i) {
		y[i] = interpolate_value(x[i], arr_x, arr_y, arr_len);
	}
}

#endif
