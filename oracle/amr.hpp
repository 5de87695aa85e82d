// amr.hpp — CPU restatement of the data-parallel AMR pieces (test infrastructure, see oracle/__init__.py):
//   tagRelativeGradient   QuokkaSimulation<problem_t>::ErrorEst of src/problems/HydroBlast3D/test_hydro3d_blast.cpp:118-151
//                         (pressure, P > P_min) and src/problems/RadhydroShell/test_radhydro_shell.cpp:337-371 (density, rho >= rho_min)
//   tagCenteredGradient   ErrorEst of src/problems/HydroShocktube/test_hydro_shocktube.cpp:146-170 (centred density difference / (2 dx))
//   averageDown           amrex::average_down / amrex_avgdown as called from AMRSimulation::AverageDownTo (src/simulation.hpp:1949-1964).
//                         AMReX is not vendored in /root/reference: the kernel is restated from its published form
//                         (crse = volfrac * sum over kref, jref, iref of the fine cells) — parity unpinned beyond that.
#ifndef ORACLE_AMR_HPP_
#define ORACLE_AMR_HPP_

#include <algorithm>
#include <cmath>

#include "grid.hpp"
#include "hydro.hpp"

namespace oracle
{

constexpr char TagBox_SET = 2; // amrex::TagBox::SET

// field < 0: HydroSystem::ComputePressure; otherwise the conserved component `field`
inline void tagRelativeGradient(HydroSystem const &hydro, Array4<const double> const &state, Array4<char> const &tag, Box const &box, int ndim, int field,
				double eta_threshold, double q_min, bool min_inclusive)
{
	auto q = [&](int i, int j, int k) -> double { return (field < 0) ? hydro.ComputePressure(state, i, j, k) : state(i, j, k, field); };
	for (int k = box.lo[2]; k <= box.hi[2]; ++k) {
		for (int j = box.lo[1]; j <= box.hi[1]; ++j) {
			for (int i = box.lo[0]; i <= box.hi[0]; ++i) {
				double const P = q(i, j, k);
				double const del_x = std::max(std::abs(q(i + 1, j, k) - P), std::abs(P - q(i - 1, j, k)));
				double del = del_x;
				if (ndim >= 2) {
					double const del_y = std::max(std::abs(q(i, j + 1, k) - P), std::abs(P - q(i, j - 1, k)));
					del = std::max(del, del_y);
				}
				if (ndim == 3) {
					double const del_z = std::max(std::abs(q(i, j, k + 1) - P), std::abs(P - q(i, j, k - 1)));
					del = std::max(del, del_z);
				}
				double const gradient_indicator = del / P;
				bool const above = min_inclusive ? (P >= q_min) : (P > q_min);
				if ((gradient_indicator > eta_threshold) && above) {
					tag(i, j, k) = TagBox_SET;
				}
			}
		}
	}
}

// del = (q(+1) - q(-1)) / (2 dx) along `dir`; SET where sqrt(del^2) / q > eta and q >= q_min
inline void tagCenteredGradient(Array4<const double> const &state, Array4<char> const &tag, Box const &box, int comp, int dir, double dx, double eta_threshold,
				double q_min, bool min_inclusive)
{
	int const e[3] = {dir == 0 ? 1 : 0, dir == 1 ? 1 : 0, dir == 2 ? 1 : 0};
	for (int k = box.lo[2]; k <= box.hi[2]; ++k) {
		for (int j = box.lo[1]; j <= box.hi[1]; ++j) {
			for (int i = box.lo[0]; i <= box.hi[0]; ++i) {
				double const rho = state(i, j, k, comp);
				double const del_x = (state(i + e[0], j + e[1], k + e[2], comp) - state(i - e[0], j - e[1], k - e[2], comp)) / (2.0 * dx);
				double const gradient_indicator = std::sqrt(del_x * del_x) / rho;
				bool const above = min_inclusive ? (rho >= q_min) : (rho > q_min);
				if (gradient_indicator > eta_threshold && above) {
					tag(i, j, k) = TagBox_SET;
				}
			}
		}
	}
}

// region: coarse cells to fill (must be covered by `fine`)
inline void averageDown(Array4<const double> const &fine, Array4<double> const &crse, Box const &region, int scomp, int ncomp, const int ratio[3])
{
	double const volfrac = 1.0 / static_cast<double>(ratio[0] * ratio[1] * ratio[2]);
	for (int n = 0; n < ncomp; ++n) {
		for (int k = region.lo[2]; k <= region.hi[2]; ++k) {
			for (int j = region.lo[1]; j <= region.hi[1]; ++j) {
				for (int i = region.lo[0]; i <= region.hi[0]; ++i) {
					double c = 0.0;
					for (int kref = 0; kref < ratio[2]; ++kref) {
						for (int jref = 0; jref < ratio[1]; ++jref) {
							for (int iref = 0; iref < ratio[0]; ++iref) {
								c += fine(i * ratio[0] + iref, j * ratio[1] + jref, k * ratio[2] + kref, n + scomp);
							}
						}
					}
					crse(i, j, k, n + scomp) = volfrac * c;
				}
			}
		}
	}
}

// amrex::mf_linear_slope_minmax_interp (method 1) / mf_pc_interp (method 0) with PreInterpState / PostInterpState, restated
// from AMReX's documentation exactly as quokka_amd/csrc/qk_amr_fill.hip does (parity with the reference UNPINNED):
// fine cells of `region` (fine index space) from  w_old * crse_old + w_new * crse_new.
inline void interpFromCoarse(Array4<double> const &fine, Array4<const double> const &crse_old, Array4<const double> const &crse_new, Box const &region,
			     double w_old, double w_new, int ncomp, int method, bool hooks, int ndim, const int ratio[3])
{
	constexpr int RHO = 0, MX = 1, MY = 2, MZ = 3, ENE = 4;
	auto tv = [&](int i, int j, int k, int c) -> double {
		double const a = crse_old(i, j, k, c);
		if (w_new == 0.0) {
			return a;
		}
		return w_old * a + w_new * crse_new(i, j, k, c);
	};
	auto cv = [&](int i, int j, int k, int n) -> double {
		if (hooks && n == ENE) {
			double const rho = tv(i, j, k, RHO), px = tv(i, j, k, MX), py = tv(i, j, k, MY), pz = tv(i, j, k, MZ), Etot = tv(i, j, k, ENE);
			double const kinetic_energy = (px * px + py * py + pz * pz) / (2.0 * rho);
			return (Etot - kinetic_energy) / rho;
		}
		return tv(i, j, k, n);
	};
	auto fdiv = [](int a, int r) { return (a >= 0) ? a / r : -((-a + r - 1) / r); };
	for (int k = region.lo[2]; k <= region.hi[2]; ++k) {
		for (int j = region.lo[1]; j <= region.hi[1]; ++j) {
			for (int i = region.lo[0]; i <= region.hi[0]; ++i) {
				int const idx[3] = {i, j, k};
				int ic[3];
				double off[3];
				for (int d = 0; d < 3; ++d) {
					ic[d] = fdiv(idx[d], ratio[d]);
					off[d] = (idx[d] - ic[d] * ratio[d] + 0.5) / ratio[d] - 0.5;
				}
				for (int n = 0; n < ncomp; ++n) {
					double const u = cv(ic[0], ic[1], ic[2], n);
					double val = u;
					if (method == 1) {
						double umax = u, umin = u;
						int const k0 = (ndim == 3) ? -1 : 0, k1 = (ndim == 3) ? 1 : 0, j0 = (ndim >= 2) ? -1 : 0, j1 = (ndim >= 2) ? 1 : 0;
						for (int c = k0; c <= k1; ++c) {
							for (int b = j0; b <= j1; ++b) {
								for (int a = -1; a <= 1; ++a) {
									double const v = cv(ic[0] + a, ic[1] + b, ic[2] + c, n);
									umax = std::max(umax, v);
									umin = std::min(umin, v);
								}
							}
						}
						double s[3] = {0.5 * (cv(ic[0] + 1, ic[1], ic[2], n) - cv(ic[0] - 1, ic[1], ic[2], n)), 0., 0.};
						if (ndim >= 2) {
							s[1] = 0.5 * (cv(ic[0], ic[1] + 1, ic[2], n) - cv(ic[0], ic[1] - 1, ic[2], n));
						}
						if (ndim == 3) {
							s[2] = 0.5 * (cv(ic[0], ic[1], ic[2] + 1, n) - cv(ic[0], ic[1], ic[2] - 1, n));
						}
						double alpha = 1.0;
						if (s[0] != 0.0 || s[1] != 0.0 || s[2] != 0.0) {
							double const dumax = std::abs(s[0]) * static_cast<double>(ratio[0] - 1) / (2.0 * ratio[0]) +
									     std::abs(s[1]) * static_cast<double>(ratio[1] - 1) / (2.0 * ratio[1]) +
									     std::abs(s[2]) * static_cast<double>(ratio[2] - 1) / (2.0 * ratio[2]);
							if (dumax * alpha > (umax - u)) {
								alpha = (umax - u) / dumax;
							}
							if (dumax * alpha > (u - umin)) {
								alpha = (u - umin) / dumax;
							}
						}
						val = u + off[0] * (s[0] * alpha) + off[1] * (s[1] * alpha) + off[2] * (s[2] * alpha);
					}
					fine(i, j, k, n) = val;
				}
				if (hooks && ncomp > ENE) {
					double const rho = fine(i, j, k, RHO), px = fine(i, j, k, MX), py = fine(i, j, k, MY), pz = fine(i, j, k, MZ);
					double const e = fine(i, j, k, ENE);
					double const Eint = rho * e;
					double const kinetic_energy = (px * px + py * py + pz * pz) / (2.0 * rho);
					fine(i, j, k, ENE) = Eint + kinetic_energy;
				}
			}
		}
	}
}

} // namespace oracle

#endif
