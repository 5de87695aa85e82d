// ode_integrate.hpp — adaptive explicit Runge-Kutta integration of small ODE systems inside a cell (the reference's src/math/ODEIntegrate.hpp: used by its
// cooling modules and tested by src/problems/ODEIntegration).  Same names, arguments and return conventions; written from the published methods:
//   rk12_single_step    Heun's method with the embedded forward-Euler solution (orders 2 / 1): error estimate = y_Heun - y_Euler
//   rk23_single_step    Bogacki & Shampine (1989), orders 3 / 2, first-same-as-last
//   error_norm          the weighted rms norm of SUNDIALS / ARKODE: sqrt(mean((err_i / (reltol * y_i + abstol_i))^2))
//   rk_adaptive_integrate   steps of rk12 under an "I" step-size controller eta = eps^(-1/p), with the growth / shrink limits ARKODE documents
//                       (eta_max 20 after a clean step, 1 after a retry; 0.3 on the second failure, [0.1, 0.3] afterwards; 0.5 after a failed rhs),
//                       at most 7 attempts per step and maxStepsODEIntegrate steps.  steps_taken = maxStepsODEIntegrate signals failure.
#ifndef QK_HOST_COMPAT_ODE_INTEGRATE_HPP_
#define QK_HOST_COMPAT_ODE_INTEGRATE_HPP_

#include <algorithm>
#include <cmath>

#include "../amrex_mini.hpp"
#include "util_compat.hpp"

using Real = amrex::Real;

// rhs(t, y, ydot, user_data) -> 0 on success
template <typename F, int N>
AMREX_GPU_HOST_DEVICE AMREX_FORCE_INLINE auto rk12_single_step(F &&rhs, Real t0, quokka::valarray<Real, N> const &y, Real dt, quokka::valarray<Real, N> &ynew,
							       quokka::valarray<Real, N> &yerr, void *user_data) -> int
{
	quokka::valarray<Real, N> slope0{}, slope1{};
	quokka::valarray<Real, N> stage = y;
	if (int const rc = rhs(t0, stage, slope0, user_data); rc != 0) {
		return rc;
	}
	slope0 *= dt; // the Euler increment
	stage = y + slope0;
	if (int const rc = rhs(t0 + dt, stage, slope1, user_data); rc != 0) {
		return rc;
	}
	slope1 *= dt;
	ynew = y + 0.5 * slope0 + 0.5 * slope1; // Heun
	yerr = -0.5 * slope0 + 0.5 * slope1;	  // Heun - Euler
	return 0;
}

template <typename F, int N>
AMREX_GPU_HOST_DEVICE AMREX_FORCE_INLINE auto rk23_single_step(F &&rhs, Real t0, quokka::valarray<Real, N> const &y, Real dt, quokka::valarray<Real, N> &ynew,
							       quokka::valarray<Real, N> &yerr, void *user_data) -> int
{
	// Bogacki-Shampine tableau: c = (0, 1/2, 3/4, 1); b3 = (2/9, 1/3, 4/9, 0); b2 = (7/24, 1/4, 1/3, 1/8)
	quokka::valarray<Real, N> k1{}, k2{}, k3{}, k4{};
	quokka::valarray<Real, N> stage = y;
	if (int const rc = rhs(t0, stage, k1, user_data); rc != 0) {
		return rc;
	}
	k1 *= dt;
	stage = y + 0.5 * k1;
	if (int const rc = rhs(t0 + 0.5 * dt, stage, k2, user_data); rc != 0) {
		return rc;
	}
	k2 *= dt;
	stage = y + 0.75 * k2;
	if (int const rc = rhs(t0 + 0.75 * dt, stage, k3, user_data); rc != 0) {
		return rc;
	}
	k3 *= dt;
	stage = y + (2. / 9.) * k1 + (1. / 3.) * k2 + (4. / 9.) * k3; // the third-order solution, also the fourth stage (FSAL)
	if (int const rc = rhs(t0 + dt, stage, k4, user_data); rc != 0) {
		return rc;
	}
	k4 *= dt;
	ynew = stage;
	yerr = (2. / 9. - 7. / 24.) * k1 + (1. / 3. - 1. / 4.) * k2 + (4. / 9. - 1. / 3.) * k3 - (1. / 8.) * k4;
	return 0;
}

template <int N>
AMREX_GPU_HOST_DEVICE AMREX_FORCE_INLINE auto error_norm(quokka::valarray<Real, N> const &y0, quokka::valarray<Real, N> const &yerr, Real reltol,
							 quokka::valarray<Real, N> const &abstol) -> Real
{
	Real sum = 0;
	for (int i = 0; i < N; ++i) {
		Real const weight = 1. / (reltol * y0[i] + abstol[i]);
		sum += (yerr[i] * yerr[i]) * (weight * weight);
	}
	return std::sqrt(sum / N);
}

constexpr int maxStepsODEIntegrate = 2000;

template <typename F, int N>
AMREX_GPU_HOST_DEVICE AMREX_FORCE_INLINE void rk_adaptive_integrate(F &&rhs, Real t0, quokka::valarray<Real, N> &y0, Real t1, void *user_data, Real reltol,
								    quokka::valarray<Real, N> const &abstol, int &steps_taken)
{
	constexpr int order = 2, attempts = 7;
	constexpr Real growAfterCleanStep = 20., growAfterRetry = 1.0, shrinkSecondFailure = 0.3, shrinkFloor = 0.1, shrinkFailedRhs = 0.5;
	// first step: a tenth of the shortest time scale |y / ydot|
	quokka::valarray<Real, N> ydot{};
	rhs(t0, y0, ydot, user_data);
	Real first = 0.1 * min(abs(y0 / ydot));
	Real t = t0;
	Real dt = std::isnan(first) ? (t1 - t0) : first;
	quokka::valarray<Real, N> err{}, trial{};
	steps_taken = maxStepsODEIntegrate; // until proven otherwise
	for (int n = 0; n < maxStepsODEIntegrate; ++n) {
		if (t + dt > t1) {
			dt = t1 - t;
		}
		bool accepted = false;
		for (int a = 0; a < attempts && !accepted; ++a) {
			Real eta;
			if (rk12_single_step(rhs, t, y0, dt, trial, err, user_data) != 0) {
				eta = shrinkFailedRhs;
			} else {
				Real const eps = error_norm(y0, err, reltol, abstol);
				eta = std::pow(eps, -1.0 / static_cast<Real>(order));
				if (eps < 1.0) {
					y0 = trial;
					t += dt;
					dt *= std::min(eta, (a == 0) ? growAfterCleanStep : growAfterRetry);
					accepted = true;
					continue;
				}
			}
			if (a == 1) {
				eta = std::min(eta, shrinkSecondFailure);
			} else if (a > 1) {
				eta = std::clamp(eta, shrinkFloor, shrinkSecondFailure);
			}
			dt *= eta;
		}
		if (!accepted) {
			return; // steps_taken = maxStepsODEIntegrate: failure
		}
		if (t >= t1) {
			steps_taken = n + 1;
			return;
		}
	}
}

#endif // QK_HOST_COMPAT_ODE_INTEGRATE_HPP_
